"""URDF -> ModelBuilder, same call signature and model semantics as the
reference's ``parse_urdf`` (/root/reference/diffphys/import_urdf.py:106-291),
built on this package's own URDF/mesh readers and ModelBuilder.

Behaviours kept on purpose (each is visible in the reference at the cited line):
  * floating base = FREE joint whose coordinates are seeded from ``xform`` (:144-165)
  * ``revolute``/``continuous`` -> REVOLUTE, ``fixed`` -> FIXED, ``floating`` -> FREE (:182-191)
  * a joint whose name ends in ``_R`` becomes a COMPOUND (x, y', z'') joint whose child is
    the matching ``*_Y`` link; the ``_P`` / ``_Y`` joints are skipped (:192-196)
  * a parent link that was never indexed falls back to the root body (:198-200)
  * limits default to +-1e3 unless the <limit> carries lower/upper (:206-214)
  * a <dynamics damping> value, once seen, sticks for all later joints (:217-219)
  * density > 0 ignores the URDF <inertial> block entirely (:225-228)
  * one collision mesh per body; every collision adds a shape (:23-103)
"""
import math

import numpy as np

from . import sim
from .urdf_io import URDF


def _add_collisions(builder, body, collisions, density, ke, kd, kf, mu):
    for c in collisions:
        pos = c.origin.xyz
        rot = sim.quat_rpy(*c.origin.rpy)
        common = dict(body=body, pos=pos, density=density, ke=ke, kd=kd, kf=kf, mu=mu)
        if c.kind == "box":
            builder.add_shape_box(rot=rot, hx=c.size[0] * 0.5, hy=c.size[1] * 0.5, hz=c.size[2] * 0.5, **common)
        elif c.kind == "sphere":
            builder.add_shape_sphere(rot=rot, radius=c.radius, **common)
        elif c.kind == "cylinder":
            # URDF cylinders run along z, capsules here along x
            r = sim.quat_from_axis_angle((0.0, 1.0, 0.0), math.pi * 0.5)
            builder.add_shape_capsule(rot=sim.quat_mul(rot, r), radius=c.radius, half_width=c.length * 0.5, **common)
        elif c.kind == "mesh":
            v, f = c.mesh()
            builder.add_shape_mesh(rot=rot, mesh=sim.Mesh(v, f.reshape(-1)), **common)


def parse_urdf(
    filename,
    builder,
    xform,
    floating=False,
    density=0.0,
    stiffness=100.0,
    damping=10.0,
    armature=0.0,
    shape_ke=1.0e4,
    shape_kd=1.0e3,
    shape_kf=1.0e2,
    shape_mu=0.25,
    limit_ke=100.0,
    limit_kd=10.0,
):
    robot = URDF.load(filename)
    link_index = {}
    builder.add_articulation()

    def inertial_of(link):
        if density == 0.0:
            return link.inertial_origin.xyz, link.inertia, link.mass
        return np.zeros(3), np.zeros((3, 3)), 0.0

    mat = (shape_ke, shape_kd, shape_kf, shape_mu)
    com, I_m, m = inertial_of(robot.base_link)
    if floating:
        root = builder.add_body(
            origin=sim.transform_identity(), parent=-1, joint_type=sim.JOINT_FREE, joint_armature=armature, com=com, I_m=I_m, m=m
        )
        s = builder.joint_q_start[root]
        builder.joint_q[s : s + 3] = [float(x) for x in xform.p]
        builder.joint_q[s + 3 : s + 7] = [float(x) for x in xform.q]
        _add_collisions(builder, root, robot.links[0].collisions, density, *mat)
    else:
        root = builder.add_body(origin=sim.transform_identity(), parent=-1, joint_xform=xform, joint_type=sim.JOINT_FIXED)
        _add_collisions(builder, root, robot.links[0].collisions, 0.0, *mat)
    link_index[robot.links[0].name] = root

    type_map = {
        "revolute": sim.JOINT_REVOLUTE,
        "continuous": sim.JOINT_REVOLUTE,
        "prismatic": sim.JOINT_PRISMATIC,
        "fixed": sim.JOINT_FIXED,
        "floating": sim.JOINT_FREE,
    }
    for joint in robot.joints:
        jtype = type_map.get(joint.joint_type)
        axis = joint.axis if jtype in (sim.JOINT_REVOLUTE, sim.JOINT_PRISMATIC) else (0.0, 0.0, 0.0)
        child_name = joint.child
        suffix = joint.name[-2:]
        if suffix == "_R":
            jtype = sim.JOINT_COMPOUND
            child_name = joint.child[:-2] + "_Y"
        elif suffix in ("_P", "_Y"):
            continue

        parent = link_index.get(joint.parent, root)
        X_pj = sim.transform(joint.origin.xyz, sim.quat_rpy(*joint.origin.rpy))
        lower = joint.limit_lower if joint.limit_lower is not None else -1.0e3
        upper = joint.limit_upper if joint.limit_upper is not None else 1.0e3
        if joint.damping:
            damping = joint.damping
        child_link = robot.link_map[child_name]
        com, I_m, m = inertial_of(child_link)

        if jtype == sim.JOINT_COMPOUND:
            link = builder.add_body(
                origin=sim.transform_identity(),
                parent=parent,
                joint_xform=X_pj,
                joint_xform_child=sim.transform((0.0, 0.0, 0.0), sim.quat_identity()),
                joint_type=jtype,
                joint_limit_lower=[lower] * 3,
                joint_limit_upper=[upper] * 3,
                joint_limit_ke=limit_ke,
                joint_limit_kd=limit_kd,
                joint_target_ke=[stiffness] * 3,
                joint_target_kd=[damping] * 3,
                joint_armature=armature,
            )
        else:
            link = builder.add_body(
                origin=sim.transform_identity(),
                parent=parent,
                joint_xform=X_pj,
                joint_axis=axis,
                joint_type=jtype,
                joint_limit_lower=lower,
                joint_limit_upper=upper,
                joint_limit_ke=limit_ke,
                joint_limit_kd=limit_kd,
                joint_target_ke=stiffness,
                joint_target_kd=damping,
                joint_armature=armature,
                com=com,
                I_m=I_m,
                m=m,
            )
        _add_collisions(builder, link, child_link.collisions, density, *mat)
        link_index[child_name] = link
