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
