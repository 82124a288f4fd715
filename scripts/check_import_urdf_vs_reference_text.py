#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (reads /root/reference; nothing of it is copied).  Mechanical check of this package's ``import_urdf.parse_urdf``
 against the TEXT of the reference's ``diffphys/import_urdf.py``: the reference's module is imported unchanged, with
stand-ins for the two libraries it needs and the image lacks --

  * ``urdfpy``: a thin adapter that presents this package's own XML reader (diffphys_amd/urdf_io.py) through the attributes the reference
    reads (``robot.links / joints / link_map / base_link``, 4x4 ``origin`` matrices, ``geometry.box / sphere / cylinder / mesh.meshes``,
    ``limit.lower / upper``, ``dynamics.damping``, ``matrix_to_xyz_rpy``);
  * ``warp``: this package's ``sim`` module (``ModelBuilder``, ``quat_rpy``, ``transform`` ...) under the names the reference calls.

Both sides therefore share the XML reader and the builder: what is compared is the ~270 lines of CONTROL FLOW between them (which joints
become bodies, `_R` / `_P` / `_Y` handling, limits / damping defaults and stickiness, which inertial block is used, collision shapes and
their arguments) -- every array of the two resulting builders, for each URDF the reference trains on and both settings of `floating` /
`density` (the reference's call site: dp_model.py:128-147, floating, density 1000).  It pins nothing about Warp's ModelBuilder itself (row f1 stays "Warp side unpinned").

    python scripts/check_import_urdf_vs_reference_text.py            # prints a table, exit code 1 on any difference
"""
import importlib
import math
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))

from diffphys_amd import import_urdf as mine  # noqa: E402
from diffphys_amd import sim, urdf_io  # noqa: E402


def rpy_matrix(xyz, rpy):
    """URDF origin -> 4x4 (fixed-axis roll, pitch, yaw: R = Rz(yaw) Ry(pitch) Rx(roll)), what urdfpy stores"""
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    R = np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                  [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                  [-sp, cp * sr, cp * cr]])
    M = np.eye(4)
    M[:3, :3], M[:3, 3] = R, xyz
    return M


def matrix_to_xyz_rpy(M):
    """inverse of rpy_matrix (urdfpy.utils.matrix_to_xyz_rpy's published convention; |pitch| < pi/2 branch first)"""
    M = np.asarray(M, dtype=np.float64)
    xyz = M[:3, 3]
    if abs(M[2, 0]) < 1.0 - 1e-12:
        p = -math.asin(M[2, 0])
        cp = math.cos(p)
        r = math.atan2(M[2, 1] / cp, M[2, 2] / cp)
        y = math.atan2(M[1, 0] / cp, M[0, 0] / cp)
    else:  # gimbal lock: yaw := 0
        y = 0.0
        if M[2, 0] < 0:
            p, r = math.pi / 2, math.atan2(M[0, 1], M[0, 2])
        else:
            p, r = -math.pi / 2, math.atan2(-M[0, 1], -M[0, 2])
    return np.array([xyz[0], xyz[1], xyz[2], r, p, y])


class NS(types.SimpleNamespace):
    pass


def as_urdfpy(robot):
    """this package's urdf_io.URDF seen through urdfpy's attribute names"""

    def geometry(c):
        g = NS(box=None, sphere=None, cylinder=None, mesh=None)
        if c.kind == "box":
            g.box = NS(size=c.size)
        elif c.kind == "sphere":
            g.sphere = NS(radius=c.radius)
        elif c.kind == "cylinder":
            g.cylinder = NS(radius=c.radius, length=c.length)
        elif c.kind == "mesh":
            v, f = c.mesh()
            g.mesh = NS(meshes=[NS(vertices=v, faces=f.reshape(-1, 3))], scale=c.mesh_scale)
        return g

    def link(l):
        return NS(name=l.name, collisions=[NS(origin=rpy_matrix(c.origin.xyz, c.origin.rpy), geometry=geometry(c)) for c in l.collisions],
                  inertial=NS(origin=rpy_matrix(l.inertial_origin.xyz, l.inertial_origin.rpy), inertia=l.inertia, mass=l.mass))

    links = [link(l) for l in robot.links]
    lmap = {l.name: l for l in links}

    def joint(j):
        has_lim = j.limit_lower is not None or j.limit_upper is not None
        return NS(name=j.name, joint_type=j.joint_type, parent=j.parent, child=j.child, axis=j.axis, origin=rpy_matrix(j.origin.xyz, j.origin.rpy),
                  limit=NS(lower=j.limit_lower, upper=j.limit_upper) if has_lim else None,
                  dynamics=NS(damping=j.damping) if j.damping is not None else None)

    return NS(links=links, joints=[joint(j) for j in robot.joints], link_map=lmap, base_link=lmap[robot.base_link.name])


def install_standins():
    up = types.ModuleType("urdfpy")
    up.matrix_to_xyz_rpy = matrix_to_xyz_rpy
    up.URDF = NS(load=lambda fn: as_urdfpy(urdf_io.URDF.load(fn)))
    wp = types.ModuleType("warp")
    wsim = types.ModuleType("warp.sim")
    wmodel = types.ModuleType("warp.sim.model")
    for k in dir(sim):
        if k.startswith("JOINT_"):
            setattr(wsim, k, getattr(sim, k))
    wsim.ModelBuilder = sim.ModelBuilder
    wmodel.Mesh = lambda vertices, faces: sim.Mesh(np.asarray(vertices, dtype=np.float64), np.asarray(faces, dtype=np.int64))
    wp.sim = wsim
    wp.transform_identity, wp.transform, wp.quat_rpy, wp.quat_from_axis_angle = sim.transform_identity, sim.transform, sim.quat_rpy, sim.quat_from_axis_angle
    wp.mul = sim.quat_mul
    wp.normalize = lambda v: np.asarray(v, dtype=np.float64) / np.linalg.norm(v)

    def quat_from_matrix(M):  # (only ever called on the identity: import_urdf.py:252-253)
        assert np.allclose(M, np.eye(3))
        return sim.quat_identity()

    wp.quat_from_matrix = quat_from_matrix
    sys.modules.update({"urdfpy": up, "warp": wp, "warp.sim": wsim, "warp.sim.model": wmodel})


def builder_arrays(b):
    out = {}
    for k, v in sorted(vars(b).items()):
        if k.startswith("_") or callable(v):
            continue
        if isinstance(v, (list, tuple)):
            flat = []
            for e in v:
                if isinstance(e, sim.transform):
                    flat.append(np.concatenate([np.asarray(e.p, dtype=np.float64), np.asarray(e.q, dtype=np.float64)]))
                elif isinstance(e, sim.Mesh):
                    flat.append(np.concatenate([np.asarray(e.vertices, dtype=np.float64).reshape(-1), np.asarray(e.indices, dtype=np.float64).reshape(-1)]))
                elif e is None:
                    flat.append(np.array([np.nan]))
                else:
                    flat.append(np.asarray(e, dtype=np.float64).reshape(-1))
            out[k] = np.concatenate(flat) if flat else np.zeros(0)
        elif isinstance(v, (int, float, np.ndarray)):
            out[k] = np.asarray(v, dtype=np.float64).reshape(-1)
    return out


def main():
    install_standins()
    sys.path.insert(0, REF)
    ref = importlib.import_module("diffphys.import_urdf")
    d = os.path.join(REF, "data/urdf_templates")
    lk = "laikago/"
    cases = [  # (urdf, floating, density): the reference's call site (dp_model.py:128-147: floating, density 1000) and the other branches
        (lk + "laikago.urdf", True, 1000.0), ("quad.urdf", True, 1000.0), ("human.urdf", True, 1000.0),
        (lk + "laikago_toes.urdf", True, 1000.0), (lk + "laikago_toes_limits.urdf", True, 1000.0), (lk + "laikago_toes_zup.urdf", True, 1000.0),
        (lk + "laikago_toes_zup_lores.urdf", True, 1000.0),
        (lk + "laikago.urdf", True, 0.0), ("quad.urdf", True, 0.0), ("human.urdf", True, 0.0),
        (lk + "laikago.urdf", False, 0.0), ("human.urdf", False, 500.0), ("quad.urdf", False, 1000.0),
    ]
    xform = sim.transform((0.1, 0.7, -0.2), sim.quat_rpy(-0.5 * math.pi, 0.1, 0.2))
    bad = 0
    print("%-42s %-9s %-8s %7s %7s %9s  %s" % ("urdf", "floating", "density", "bodies", "shapes", "arrays", "max |reference text - this package|"))
    for fn, floating, density in cases:
        path = os.path.join(d, fn)
        if not os.path.exists(path):
            continue
        kw = dict(floating=floating, density=density, stiffness=220.0, damping=2.0, armature=0.01, shape_ke=1e4, shape_kd=1e3, shape_kf=1e2, shape_mu=1.0,
                  limit_ke=0.0, limit_kd=0.0)
        ba, bb = sim.ModelBuilder(), sim.ModelBuilder()
        ref.parse_urdf(path, ba, xform, **kw)
        mine.parse_urdf(path, bb, xform, **kw)
        A, B = builder_arrays(ba), builder_arrays(bb)
        worst, where = 0.0, ""
        if A.keys() != B.keys():
            worst, where = np.inf, "different attributes %s" % (set(A) ^ set(B))
        for k in A:
            if k not in B:
                continue
            if A[k].shape != B[k].shape:
                worst, where = np.inf, "%s: shape %s vs %s" % (k, A[k].shape, B[k].shape)
                break
            if A[k].size:
                e = np.nanmax(np.abs(A[k] - B[k])) if not np.all(np.isnan(A[k]) == np.isnan(B[k])) or np.any(~np.isnan(A[k])) else 0.0
                if np.any(np.isnan(A[k]) != np.isnan(B[k])):
                    e = np.inf
                if e > worst:
                    worst, where = e, k
        print("%-42s %-9s %-8g %7d %7d %9d  %.3g %s" % (fn, floating, density, len(ba.body_mass), len(ba.shape_body), len(A), worst, where))
        bad += worst > 1e-10
    print("import_urdf vs the reference's text: %s" % ("SAME builders (<= 1e-10: the rpy -> 4x4 -> rpy round trip of the urdfpy stand-in; every count, index and flag equal)" if not bad else "%d case(s) DIFFER" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
