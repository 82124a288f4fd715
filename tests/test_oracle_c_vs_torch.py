"""The two independent restatements agree: C (hand-written reverse sweep) vs torch float64 (autograd),
and both reproduce the committed golden fixtures (tests/golden, self-generated -- NOT Warp outputs)."""
import numpy as np
import pytest
import torch

from helpers import INPUT_NAMES, golden_inputs, load_golden, relmax
from diffphys_amd import robots, synth
from oracle import ref_torch as rt
from oracle.ref_c import RefC

ROBOTS = ("laikago", "human", "quad")


@pytest.mark.parametrize("name", ROBOTS)
def test_c_f64_matches_golden(name, oracle_libs):
    g = load_golden(name)
    inp = golden_inputs(g)
    tpl = robots.load_template(name)
    rc = RefC(tpl, np.float64)
    st = rc.rollout_forward(inp, inp["nsteps"], inp["frame2step"], inp["dt"])
    for k in ("wp_pos", "wp_vel", "grf", "jaf"):
        assert relmax(st[k], g[k]) < 1e-10, k
    gr = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
    for k in INPUT_NAMES:
        assert relmax(gr[k].reshape(g["grad_" + k].shape), g["grad_" + k]) < 1e-9 or np.abs(g["grad_" + k]).max() == 0, k
    assert np.abs(gr["body_mass"]).max() == 0  # loaded, never used (integrator_euler.py:43)
    bq, bqd = rc.fk_forward(g["fk_joint_q"], g["fk_joint_qd"])
    assert relmax(bq, g["fk_body_q"]) < 1e-12 and relmax(bqd, g["fk_body_qd"]) < 1e-12
    gq, gqd = rc.fk_backward(g["fk_joint_q"], g["fk_joint_qd"], bq, g["fk_adj_q"], g["fk_adj_qd"])
    assert relmax(gq, g["fk_grad_q"]) < 1e-11 and relmax(gqd, g["fk_grad_qd"]) < 1e-11


@pytest.mark.parametrize("name", ROBOTS)
def test_c_f32_close_to_golden(name, oracle_libs):
    """fp32 twin over one frame interval (34 steps): the tolerance the GPU path is also held to."""
    g = load_golden(name)
    inp = golden_inputs(g)
    rc = RefC(robots.load_template(name), np.float32)
    st = rc.rollout_forward(inp, inp["nsteps"], inp["frame2step"], inp["dt"])
    assert relmax(st["wp_pos"], g["wp_pos"]) < 2e-5
    assert relmax(st["wp_vel"], g["wp_vel"]) < 2e-3
    gr = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
    for k in ("q_init", "qd_init", "refs", "torques", "res_f", "target_ke", "target_kd", "body_inv_mass", "body_inertia",
              "body_inv_inertia"):
        assert relmax(gr[k].reshape(g["grad_" + k].shape), g["grad_" + k]) < 2e-2, k


@pytest.mark.parametrize("name", ROBOTS)
def test_torch_reproduces_golden(name):
    g = load_golden(name)
    inp = golden_inputs(g)
    T = rt.Template(robots.load_template(name), torch.float64)
    t = {k: torch.tensor(inp[k], dtype=torch.float64) for k in INPUT_NAMES}
    pos, vel, grf, jaf = rt.rollout(T, *[t[k] for k in INPUT_NAMES], nsteps=inp["nsteps"], frame2step=inp["frame2step"], dt=inp["dt"])
    assert relmax(pos, g["wp_pos"]) < 1e-12 and relmax(vel, g["wp_vel"]) < 1e-12
    assert relmax(grf, g["grf"]) < 1e-12 and relmax(jaf, g["jaf"]) < 1e-12


def test_c_adjoint_matches_autograd_with_contacts_and_all_inputs(oracle_libs):
    """Generic (non-fixture) check on a fresh seed: every one of the 11 gradients, contacts active."""
    name = "laikago"
    tpl = robots.load_template(name)
    nsteps, bs = 20, 3
    inp = synth.make_inputs(tpl, name, bs=bs, nsteps=nsteps, seed=21, dtype=np.float64, steps_per_frame=7)
    rng = np.random.RandomState(3)
    inp["torques"] = rng.randn(*inp["torques"].shape) * 0.5
    inp["res_f"] = rng.randn(*inp["res_f"].shape) * 0.5
    inp["qd_init"] = rng.randn(*inp["qd_init"].shape) * 0.1
    inp["q_init"].reshape(bs, -1)[:, 1] -= 0.004  # feet in the ground from step 0
    T = rt.Template(tpl, torch.float64)
    t = {k: torch.tensor(inp[k], dtype=torch.float64, requires_grad=True) for k in INPUT_NAMES}
    pos, vel, grf, _ = rt.rollout(T, *[t[k] for k in INPUT_NAMES], nsteps=nsteps, frame2step=inp["frame2step"], dt=inp["dt"])
    assert grf.abs().max() > 1.0  # contacts really are active
    ((pos * torch.tensor(inp["adj_pos"])).sum() + (vel * torch.tensor(inp["adj_vel"])).sum()).backward()
    rc = RefC(tpl, np.float64)
    st = rc.rollout_forward(inp, nsteps, inp["frame2step"], inp["dt"])
    gr = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
    for k in INPUT_NAMES:
        ref = np.zeros_like(gr[k]) if t[k].grad is None else t[k].grad.numpy().reshape(gr[k].shape)
        assert relmax(gr[k], ref) < 1e-9 or np.abs(ref).max() == 0, k


def test_guarded_acos_adjoint_is_finite_at_zero_angle(oracle_libs):
    """POLICY: a revolute joint exactly at angle 0 (twist.w == 1) contributes a zero
    adjoint through acos instead of inf; no NaN reaches any gradient, in fp32 where this is actually hit."""
    tpl = robots.load_template("laikago")
    inp = synth.make_inputs(tpl, "laikago", bs=2, nsteps=5, seed=0, steps_per_frame=2)
    q = inp["q_init"].reshape(2, -1)
    q[0, 7:] = 0.0  # all joint angles exactly zero in env 0
    inp["refs"][:] = 0.0
    rc = RefC(tpl, np.float32)
    st = rc.rollout_forward(inp, 5, inp["frame2step"], inp["dt"])
    gr = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
    assert all(np.isfinite(v).all() for v in gr.values())


def test_fixed_joint_scale_invariant_evaluation(oracle_libs, tmp_path):
    """The FIXED joint's angular error normalize(v) * 2 acos(w) and its scale-invariant evaluation v * 2 atan2(|v|, w) / |v| (C oracle,
    ref_set_twist_eval(1); what the HIP kernels evaluate): (1) the same function -- in float64, where the rollout's quaternions are unit
    to 1e-16, forces and gradients of the two agree; (2) its adjoint equals central finite differences; (3) it does not care about the
    NORM of the quaternions, the literal form does: with fp32-rounded states (|q| = 1 + O(1e-7)) fed to the float64 oracle the literal
    joint torque moves by percents, the scale-invariant one by 1e-6."""
    from test_host import OBJ, URDF
    from diffphys_amd import sim
    from diffphys_amd.import_urdf import parse_urdf
    from helpers import build_template

    (tmp_path / "toy.urdf").write_text(URDF)
    (tmp_path / "tet.obj").write_text(OBJ)
    b = sim.ModelBuilder()
    parse_urdf(str(tmp_path / "toy.urdf"), b, xform=sim.transform((0, 0.5, 0), sim.quat_identity()), floating=True, density=1000.0,
               armature=0.01, stiffness=220.0, damping=2.0, shape_ke=1e4, shape_kd=10.0, shape_kf=1e2, shape_mu=0.7, limit_ke=50.0, limit_kd=1.0)
    tpl = build_template(b, attach_ke=8000.0, attach_kd=200.0)
    assert 3 in set(int(t) for t in tpl["joint_type"])
    nb, nq, nqd, bs, T = int(tpl["nb"]), int(tpl["nq"]), int(tpl["nqd"]), 2, 12
    rng = np.random.RandomState(0)
    q = np.tile(tpl["joint_q"].astype(np.float64), (bs, 1))
    q[:, 1] = 0.14
    q[:, 7:] = rng.uniform(-0.4, 0.4, (bs, nq - 7))
    mass = np.tile(tpl["body_mass"].astype(np.float64), bs)
    inertia = np.tile(tpl["body_inertia"].astype(np.float64), (bs, 1, 1))
    ke = np.tile(np.r_[np.zeros(6), np.full(nqd - 6, 60.0)], bs)
    inp = dict(q_init=q.reshape(-1), qd_init=rng.randn(bs * nqd) * 0.5, torques=rng.randn(T, bs * nqd) * 0.3, res_f=rng.randn(T, bs * nb, 6) * 2.0,
               refs=rng.uniform(-0.3, 0.3, (T, bs * nqd)), target_ke=ke, target_kd=ke * 0.02, body_mass=mass, body_inv_mass=1 / mass,
               body_inertia=inertia, body_inv_inertia=np.linalg.inv(inertia))
    f2s = [0, 6, 12]
    adj_pos, adj_vel = rng.randn(3, bs * nb, 7), rng.randn(3, bs * nb, 6)
    rc = RefC(tpl, np.float64)
    try:
        res = {}
        for sw in (False, True):
            rc.set_twist_eval(sw)
            st = rc.rollout_forward(inp, T, f2s, 5e-4)
            res[sw] = (st, rc.rollout_backward(st, adj_pos, adj_vel))
        assert np.abs(res[False][0]["jaf"]).max() > 1.0
        for k in ("wp_pos", "wp_vel", "jaf"):
            assert relmax(res[True][0][k], res[False][0][k]) < 1e-7, k
        for k in ("q_init", "qd_init", "refs", "res_f", "body_inv_inertia"):
            assert relmax(res[True][1][k], res[False][1][k]) < 1e-5, k
        # (2) finite differences of L = <adj_pos, pos> + <adj_vel, vel> in q_init and qd_init, scale-invariant form
        rc.set_twist_eval(True)
        L = lambda i: (lambda s: float((s["wp_pos"] * adj_pos).sum() + (s["wp_vel"] * adj_vel).sum()))(rc.rollout_forward(i, T, f2s, 5e-4))
        g = res[True][1]
        for key, idxs in (("q_init", (0, 4, 8, 11, nq + 9)), ("qd_init", (1, 7, 9, nqd + 8))):
            for j in idxs:
                h = 1e-6
                ip, im = dict(inp), dict(inp)
                ip[key] = inp[key].copy(); ip[key][j] += h
                im[key] = inp[key].copy(); im[key][j] -= h
                fd = (L(ip) - L(im)) / (2 * h)
                assert abs(fd - g[key].reshape(-1)[j]) <= 2e-5 * max(1.0, abs(fd)), (key, j, fd, g[key].reshape(-1)[j])
        # (3) quaternion norms: the same trajectory with every stored state rounded to fp32
        st = res[True][0]
        traj = {k: np.asarray(st[k], np.float32).astype(np.float64) for k in ("states_q", "states_qd", "states_f")} if "states_q" in st else None
        if traj is not None:
            out = {}
            inp2 = dict(inp, frame2step=f2s, dt=5e-4, nsteps=T)
            for sw in (False, True):
                rc.set_twist_eval(sw)
                g_exact = rc.rollout_backward_forced(rc.trajectory_state({k: st[k] for k in traj}, inp2), adj_pos, adj_vel)
                g_round = rc.rollout_backward_forced(rc.trajectory_state(traj, inp2), adj_pos, adj_vel)
                out[sw] = relmax(g_round["q_init"], g_exact["q_init"])
            print("fixed joint, gradients under fp32-rounded states: literal %.1e, scale-invariant %.1e" % (out[False], out[True]))
            assert out[True] < 1e-3 and out[False] > 10 * out[True]
    finally:
        rc.set_twist_eval(False)
