"""Per-robot presets and the articulation-template build.

Mirrors the constants and the mass / inertia re-assignment that
``phys_model.__init__`` applies before any env is created
(/root/reference/diffphys/dp_model.py:76-205).  The result of
:func:`build_articulation` is what the reference calls ``articulation_builder``.

Compiled templates (``templates/<robot>.npz``) are produced from the URDFs by
``scripts/compile_templates.py`` so that tests, bench and smoke need neither
the URDF files nor any reference path at run time.
"""
import os

import numpy as np

from . import sim
from .import_urdf import parse_urdf

# (urdf relative path, joint_attach_ke, joint_attach_kd, kp, kd, shape_ke, shape_kd)  dp_model.py:83-119
PRESETS = {
    "laikago": ("laikago/laikago.urdf", 16000.0, 200.0, 220.0, 2.0, 1.0e4, 0.0),
    "quad": ("quad.urdf", 8000.0, 200.0, 660.0, 5.0, 1.0e4, 0.0),
    "human": ("human.urdf", 8000.0, 200.0, 660.0, 5.0, 1.0e4, 0.0),
}

# links whose box is doubled and made heavier (/root/reference/diffphys/robot.py:61-66,79-82)
KP_LINKS = {
    "quad": ["link_155_Vorderpfote_R_Y", "link_150_Vorderpfote_L_Y", "link_170_Pfote2_R_Y", "link_165_Pfote2_L_Y"],
    "human": ["link_24_mixamorig:RightFoot_Y", "link_19_mixamorig:LeftFoot_Y"],
}

# How `link_weight = clip(1e3 * np.prod(shape_geo_scale[idx]), 1, 5)` (/root/reference/diffphys/dp_model.py:185-191) reads,
# which depends on what warp_lang 0.7.2's ModelBuilder stores in shape_geo_scale (un-vendored):
#   "prod3"      a 3-tuple (hx, hy, hz): the product is the box's half-extent volume            <- shipped default
#   "prod4_zero" a 4-tuple with a trailing 0.0: the product is 0 for every link whose tuple the reference did not rebuild,
#                i.e. all non-foot links get exactly 1.0; the feet (rebuilt as a 3-tuple at :173-177) keep the volume rule
# The default follows the reference's own code, which rebuilds the feet's entry as a 3-tuple and multiplies all of a tuple's
# entries -- consistent only with 3-tuples.  The choice is recorded in the compiled template ("mass_rule").
MASS_RULES = ("prod3", "prod4_zero")
DEFAULT_MASS_RULE = "prod3"


def link_weight(scale3, rule, is_kp_link):
    """clip(1e3 * prod(shape_geo_scale entry), 1, 5) under the given reading; scale3 = (hx, hy, hz) after the foot doubling."""
    if rule not in MASS_RULES:
        raise ValueError("mass_rule must be one of %s" % (MASS_RULES,))
    prod = float(np.prod(scale3)) if (rule == "prod3" or is_kp_link) else 0.0
    return float(min(5.0, max(1.0, 1e3 * prod)))


TEMPLATE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "templates")


def build_articulation(name, urdf_root, mass_rule=DEFAULT_MASS_RULE):
    """Returns (builder, info).  ``urdf_root`` is the directory holding
    ``laikago/laikago.urdf``, ``human.urdf``, ``quad.urdf`` (the reference keeps
    them under ``data/urdf_templates``).  ``mass_rule``: see MASS_RULES (human / quad only)."""
    rel, attach_ke, attach_kd, kp, kd, shape_ke, shape_kd = PRESETS[name]
    urdf_path = os.path.join(urdf_root, rel)
    b = sim.ModelBuilder()
    parse_urdf(
        urdf_path,
        b,
        xform=sim.transform((0.0, 0.417, 0.0), sim.quat_from_axis_angle((1.0, 0.0, 0.0), 0.0)),
        floating=True,
        density=1000,
        armature=0.01,
        stiffness=220.0,
        damping=2.0,
        shape_ke=shape_ke,
        shape_kd=shape_kd,
        shape_kf=1.0e2,
        shape_mu=1,
        limit_ke=0,
        limit_kd=0,
    )
    body_names = _body_link_names(urdf_path)
    if name in KP_LINKS:
        # dp_model.py:151-191: feet x2 in size, x8 mass, x32 inertia; inertia normalised by mass;
        # mass replaced by clip(1e3 * prod(half extents), 1, 5)
        for idx, link_name in enumerate(body_names):
            tup = b.shape_geo_scale[idx]
            if link_name in KP_LINKS[name]:
                b.shape_geo_scale[idx] = (tup[0] * 2, tup[1] * 2, tup[2] * 2)
                b.body_mass[idx] *= 2 ** 3
                b.body_inertia[idx] = b.body_inertia[idx] * 2 ** 5
            b.body_inertia[idx] = b.body_inertia[idx] / b.body_mass[idx]
            b.body_mass[idx] = link_weight(b.shape_geo_scale[idx], mass_rule, link_name in KP_LINKS[name])
    else:
        for idx in range(len(b.body_mass)):  # dp_model.py:194-196
            b.body_inertia[idx] = b.body_inertia[idx] / b.body_mass[idx]

    n = len(b.joint_target_ke)
    b.joint_target_ke = [0.0] * 6 + [kp] * (n - 6)  # dp_model.py:200-205
    b.joint_target_kd = [0.0] * 6 + [kd] * (n - 6)
    info = dict(joint_attach_ke=attach_ke, joint_attach_kd=attach_kd, kp=kp, kd=kd, body_names=body_names,
                mass_rule=mass_rule if name in KP_LINKS else "mesh_density")
    return b, info


def _body_link_names(urdf_path):
    """Link name of each body, in body order (root, then one per kept joint)."""
    from .urdf_io import URDF

    robot = URDF.load(urdf_path)
    names = [robot.links[0].name]
    for j in robot.joints:
        suf = j.name[-2:]
        if suf == "_R":
            names.append(j.child[:-2] + "_Y")
        elif suf in ("_P", "_Y"):
            continue
        else:
            names.append(j.child)
    return names


def make_env(name, urdf_root, num_envs, device="cuda", mass_rule=DEFAULT_MASS_RULE):
    """What ``reinit_envs`` does (/root/reference/diffphys/dp_model.py:384-401)."""
    art, info = build_articulation(name, urdf_root, mass_rule)
    builder = sim.ModelBuilder()
    for _ in range(num_envs):
        builder.add_rigid_articulation(art)
    env = builder.finalize(device)
    env.ground = True
    env.joint_attach_ke = info["joint_attach_ke"]
    env.joint_attach_kd = info["joint_attach_kd"]
    env.collide(None)
    return env, art, info


def load_template(name):
    """Compiled template dict (numpy arrays) from ``templates/<name>.npz``."""
    path = os.path.join(TEMPLATE_DIR, name + ".npz")
    with np.load(path) as z:
        tpl = {k: z[k] for k in z.files}
    tpl.setdefault("mass_rule", np.asarray("mesh_density" if name not in KP_LINKS else DEFAULT_MASS_RULE))
    return tpl


def env_from_template(name, num_envs, device="cuda"):
    return sim.Model.from_template(load_template(name), num_envs, device)
