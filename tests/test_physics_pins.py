"""The three items of the oracle that rest on RECALL of un-vendored warp_lang 0.7.2, pinned by
physics / measured instead of by memory ("Next" item 1d).  CPU only.

  A.3  eval_fk's twist: checked against float64 finite differences of the FK pose itself.  What holds exactly: the angular
       velocity of every body is the time derivative of its orientation; the linear twist of a body whose ancestors are at
       rest is the velocity of its origin plus omega x com_BODY-FRAME.  Warp's formula (as recalled) keeps the body-frame
       com in that cross product, so the result is the physical COM velocity only up to  omega x (com - R com)  -- the test
       asserts that identity, i.e. it pins the implementation to physics plus ONE named term.
  A.2  mesh mass properties: a closed mesh's mass / inertia from the tetrahedral quadrature equal the analytic solid's (box:
       exact; icosphere: converging with the tessellation); duplicated vertices / degenerate triangles (what an STL soup
       delivers) do not change the mass, and the inertia obeys the parallel-axis identity about the vertex mean.
       The quad / human mass re-assignment depends on whether ModelBuilder stores shape_geo_scale as a 3- or a 4-tuple: both
       readings are implemented (robots.MASS_RULES), named in the template, and tested.
  A.1  acos / asin: the build's policy (clamped argument, guarded adjoint) is a NAMED DEVIATION from the survey's recall
       (unguarded).  The C oracle runs both; the difference is measured here.
"""
import math

import numpy as np
import pytest
import torch

from helpers import build_template, chain2
from diffphys_amd import mesh_io, robots, sim
from oracle import ref_torch as rt


# --------------------------------------------------------------------------------------------------------- A.3
def _fk(T, q, qd):
    bq, bqd = rt.eval_fk(T, torch.as_tensor(q)[None], torch.as_tensor(qd)[None])
    return bq[0].numpy(), bqd[0].numpy()


def _advance(tpl, q, qd, h):
    """joint coordinates after time h at constant joint rates (free joint: world-frame v, w; quaternion by exponential)."""
    q = q.copy()
    for i in range(int(tpl["nb"])):
        ty, qs, ds = int(tpl["joint_type"][i]), int(tpl["joint_q_start"][i]), int(tpl["joint_qd_start"][i])
        if ty == sim.JOINT_FREE:
            w, v = qd[ds:ds + 3], qd[ds + 3:ds + 6]
            q[qs:qs + 3] += h * v
            ang = np.linalg.norm(w) * h
            ax = w / (np.linalg.norm(w) + 1e-300)
            dq = np.r_[ax * math.sin(ang / 2), math.cos(ang / 2)]
            x, y, z, ww = dq
            a, b, c, d = q[qs + 3:qs + 7]
            q[qs + 3:qs + 7] = [ww * a + d * x + y * c - z * b, ww * b + d * y + z * a - x * c, ww * c + d * z + x * b - y * a,
                                ww * d - x * a - y * b - z * c]  # dq * q  (world-frame angular velocity)
        elif ty == sim.JOINT_REVOLUTE:
            q[qs] += h * qd[ds]
        elif ty == sim.JOINT_COMPOUND:
            q[qs:qs + 3] += h * qd[ds:ds + 3]
    return q


def _qrot(q, v):
    u, w = q[:3], q[3]
    return v * (2 * w * w - 1) + 2 * w * np.cross(u, v) + 2 * u * np.dot(u, v)


def _omega_from_quats(q0, q1, h):
    """angular velocity (world) from two orientations h apart: dq = q1 * conj(q0)."""
    x, y, z, w = q1
    a, b, c, d = -q0[0], -q0[1], -q0[2], q0[3]
    dq = np.array([w * a + d * x + y * c - z * b, w * b + d * y + z * a - x * c, w * c + d * z + x * b - y * a, w * d - x * a - y * b - z * c])
    if dq[3] < 0:
        dq = -dq
    return 2 * dq[:3] / h


@pytest.mark.parametrize("jt", [sim.JOINT_REVOLUTE, sim.JOINT_COMPOUND])
def test_fk_twist_is_the_time_derivative_of_the_fk_pose(jt):
    tpl = build_template(chain2(jt, axis=(0.0, 0.6, 0.8)), attach_ke=8000.0, attach_kd=200.0)
    T = rt.Template(tpl, torch.float64)
    nq, nqd = T.nq, T.nqd
    com = tpl["body_com"].astype(np.float64)
    rng = np.random.RandomState(3)
    h = 1e-6
    for case in ("root only", "child only", "both"):
        q = np.zeros(nq)
        q[:7] = [0.3, 1.0, -0.2, 0.2, -0.3, 0.1, 0.9]
        q[3:7] /= np.linalg.norm(q[3:7])
        q[7:] = rng.uniform(-0.5, 0.5, nq - 7)
        qd = np.zeros(nqd)
        if case != "child only":
            qd[:6] = rng.uniform(-1, 1, 6)
        if case != "root only":
            qd[6:] = rng.uniform(-1, 1, nqd - 6)
        b0, v0 = _fk(T, q, qd)
        bm, _ = _fk(T, _advance(tpl, q, qd, -h), qd)
        bp, _ = _fk(T, _advance(tpl, q, qd, +h), qd)
        for i in range(2):
            # (1) angular part: exactly the derivative of the orientation, for every body and every motion.  For the free
            #     root this also fixes the frame of joint_qd's angular part (the joint frame = world for a root at X_p = id)
            w_fd = _omega_from_quats(bm[i, 3:], bp[i, 3:], 2 * h)
            assert np.allclose(v0[i, :3], w_fd, atol=1e-6), (case, i, v0[i, :3], w_fd)
        # (2) linear part of the ROOT: origin velocity + omega x com (body-frame com, Warp's formula as recalled) ...
        v_origin = (bp[0, :3] - bm[0, :3]) / (2 * h)
        assert np.allclose(v0[0, 3:], v_origin + np.cross(v0[0, :3], com[0]), atol=1e-6), case
        # ... which is the physical COM velocity up to the one named term omega x (com - R com)
        x_com = lambda b: b[0, :3] + _qrot(b[0, 3:], com[0])
        v_com = (x_com(bp) - x_com(bm)) / (2 * h)
        named = np.cross(v0[0, :3], com[0] - _qrot(b0[0, 3:], com[0]))
        assert np.allclose(v0[0, 3:], v_com + named, atol=1e-6), case
        if case == "child only":
            # (3) a child moving on a resting parent: its twist is the joint's own: v = omega_rel x com_child (same form)
            w_rel = v0[1, :3]
            assert np.allclose(v0[1, 3:], np.cross(w_rel, com[1]), atol=1e-9)
            assert np.abs(v0[0]).max() == 0


# --------------------------------------------------------------------------------------------------------- A.2
def _box_mesh(hx, hy, hz):
    v = np.array([[sx * hx, sy * hy, sz * hz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]  # outward
    f = [(a, b, c) for a, b, c, d in quads] + [(a, c, d) for a, b, c, d in quads]
    return v, np.array(f)


def _icosphere(level):
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    v = [np.array(p, float) / np.linalg.norm(p) for p in v]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(level):
        cache, nf = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[k] = len(v) - 1
            return cache[k]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v), np.array(f)


def test_mesh_mass_properties_equal_the_analytic_solid():
    # closed box mesh: the order-2 tetrahedral quadrature integrates x_i x_j exactly -> mass and inertia are the box's
    hx, hy, hz = 0.3, 0.1, 0.2
    v, f = _box_mesh(hx, hy, hz)
    m = sim.Mesh(v, f.reshape(-1))
    vol = 8 * hx * hy * hz
    assert abs(m.mass - vol) < 1e-12 and np.allclose(m.com, 0)
    assert np.allclose(m.I, np.diag([vol / 3 * (hy * hy + hz * hz), vol / 3 * (hx * hx + hz * hz), vol / 3 * (hx * hx + hy * hy)]), atol=1e-12)
    # through the builder, scaled: m = rho s^3 V, I = rho s^5 I_1  (what add_shape_mesh applies)
    b = sim.ModelBuilder()
    b.add_articulation()
    body = b.add_body(origin=sim.transform_identity(), parent=-1, joint_type=sim.JOINT_FREE, joint_armature=0.0)
    b.add_shape_mesh(body, mesh=m, scale=(2.0, 2.0, 2.0), density=500.0)
    assert abs(b.body_mass[body] - 500.0 * 8 * vol) < 1e-9
    assert np.allclose(b.body_inertia[body], 500.0 * 32 * m.I, atol=1e-9)
    # icospheres converge to the sphere (mass 4/3 pi r^3, inertia 2/5 m r^2) from below, ~4x closer per subdivision
    errs = []
    for level in (1, 2, 3):
        vs, fs = _icosphere(level)
        ms = sim.Mesh(vs, fs.reshape(-1))
        m_ref = 4.0 / 3.0 * math.pi
        errs.append((abs(ms.mass - m_ref) / m_ref, np.abs(ms.I - np.eye(3) * 0.4 * m_ref).max() / (0.4 * m_ref)))
        assert ms.mass < m_ref and np.allclose(ms.I, np.eye(3) * ms.I[0, 0], atol=1e-9)
    assert errs[2][0] < 0.01 and errs[2][1] < 0.02
    assert errs[1][0] / errs[2][0] > 3.5 and errs[0][0] / errs[1][0] > 3.5


def test_duplicated_vertices_and_degenerate_triangles():
    """What an STL delivers: a triangle soup (every vertex repeated per face), plus a zero-area sliver.  After the merge the
    vertex count is the solid's, the sliver is gone; the mass is the solid's volume either way, and the inertia -- taken about
    the VERTEX MEAN, which the duplicates move -- obeys the parallel-axis identity (so merged or not is the same physics)."""
    hx, hy, hz = 0.3, 0.1, 0.2
    v, f = _box_mesh(hx, hy, hz)
    v = v + np.array([0.05, -0.02, 0.01])                      # off-centre: vertex mean matters
    soup_v = v[f].reshape(-1, 3)                               # 36 vertices, 8 distinct
    soup_v = np.concatenate([soup_v, soup_v[:1], soup_v[:1], soup_v[1:2]])  # + a degenerate triangle (two equal corners)
    soup_f = np.arange(len(soup_v)).reshape(-1, 3)
    mv, mf = mesh_io._merge_vertices(soup_v, soup_f)
    assert len(mv) == 8 and len(mf) == 12
    merged, soup = sim.Mesh(mv, mf.reshape(-1)), sim.Mesh(soup_v, soup_f.reshape(-1))
    vol = 8 * hx * hy * hz
    assert abs(merged.mass - vol) < 1e-12 and abs(soup.mass - vol) < 1e-12
    centroid = np.array([0.05, -0.02, 0.01])
    I_c = np.diag([vol / 3 * (hy * hy + hz * hz), vol / 3 * (hx * hx + hz * hz), vol / 3 * (hx * hx + hy * hy)])
    for mesh in (merged, soup):
        d = mesh.com - centroid                                # com = vertex mean (Warp's choice), not the centroid
        assert np.allclose(mesh.I, I_c + vol * (np.dot(d, d) * np.eye(3) - np.outer(d, d)), atol=1e-12)
    assert np.allclose(merged.com, centroid) and not np.allclose(soup.com, centroid)


def test_mass_rules_both_readings_of_shape_geo_scale():
    """dp_model.py:185-191 of the reference: link_weight = clip(1e3 * prod(shape_geo_scale[idx]), 1, 5).  With a 3-tuple
    (hx, hy, hz) that is the box volume / 8 in litres; with Warp builds that store a trailing 0 (a 4-tuple) the product is 0
    and every NON-foot link gets exactly 1.0, while the feet -- whose tuple the reference rebuilds as a 3-tuple -- keep the
    volume rule.  Both readings are selectable and named in the template; the shipped default is the 3-tuple one."""
    assert set(robots.MASS_RULES) == {"prod3", "prod4_zero"} and robots.DEFAULT_MASS_RULE == "prod3"
    scales = [(0.1, 0.2, 0.1), (0.02, 0.02, 0.02), (0.2, 0.2, 0.1)]
    for rule in robots.MASS_RULES:
        out = [robots.link_weight(s, rule, is_kp_link=False) for s in scales]
        kp = robots.link_weight(scales[2], rule, is_kp_link=True)
        if rule == "prod3":
            assert np.allclose(out, [2.0, 1.0, 4.0]) and abs(kp - 4.0) < 1e-12
        else:
            assert out == [1.0, 1.0, 1.0] and abs(kp - 4.0) < 1e-12
    for name in ("human", "quad", "laikago"):
        tpl = robots.load_template(name)
        assert str(tpl["mass_rule"]) == ("prod3" if name != "laikago" else "mesh_density")


# --------------------------------------------------------------------------------------------------------- A.1
def test_acos_policies_measured(oracle_libs):
    """Guarded (default) vs unguarded (SURVEY App. A.1's recall) acos / asin in the C oracle: identical away from |x| = 1;
    at a joint angle of exactly 0 (twist.w == 1 in fp32) the unguarded policy turns that env's gradients into NaN / inf --
    which ForwardWarp.backward scrubs to 0 (remove_nan) -- while the guarded one drops one term and stays finite."""
    from diffphys_amd import synth
    from oracle.ref_c import RefC

    tpl = robots.load_template("laikago")
    rc = RefC(tpl, np.float32)
    bs, T = 6, 8
    inp = synth.make_inputs(tpl, "laikago", bs=bs, nsteps=T, seed=0, steps_per_frame=3)
    inp["refs"] = inp["refs"] + 0.05                       # regular: every joint away from its reference
    q = inp["q_init"].reshape(bs, -1)
    q[:, 7:] += 0.03
    res = {}
    try:
        for unguarded in (False, True):
            rc.set_acos_policy(unguarded)
            st = rc.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
            res[unguarded] = (st["wp_pos"].copy(), rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"]))
        assert np.array_equal(res[False][0], res[True][0])
        for k in res[False][1]:
            assert np.array_equal(res[False][1][k], res[True][1][k]), k   # no |x| = 1 anywhere: the policies coincide
        # singular: env 0 with all joint angles exactly 0 and zero references -> twist.w == 1
        q[0, 7:] = 0.0
        inp["refs"].reshape(T, bs, -1)[:, 0, :] = 0.0
        for unguarded in (False, True):
            rc.set_acos_policy(unguarded)
            st = rc.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
            res[unguarded] = (st["wp_pos"].copy(), rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"]))
    finally:
        rc.set_acos_policy(False)
    g_guard, g_raw = res[False][1]["q_init"].reshape(bs, -1), res[True][1]["q_init"].reshape(bs, -1)
    assert np.isfinite(g_guard).all()
    assert not np.isfinite(g_raw[0]).all(), "the unguarded adjoint must blow up in the env that sits on the singularity"
    assert np.array_equal(g_guard[1:], g_raw[1:])          # the other envs are unaffected
    assert np.array_equal(res[False][0], res[True][0])     # forward values: acos(1) = 0 either way
