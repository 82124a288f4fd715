"""Parity tests proper (run on a real MI355X with -m gpu).  Everything goes through the C ABI
(libpprdiffphys_hip.so via ctypes); the oracles are only the checker.

Stated tolerances (fp32, relative to the max magnitude of the compared tensor, "relmax"):
  * one frame interval (34 steps) against the float64 golden fixtures:
      body poses 5e-5, body twists 2e-3, ground / joint wrenches 5e-3, every gradient 2e-2
    (the fp32 C oracle meets the same bars on CPU: tests/test_oracle_c_vs_torch.py)
  * against the fp32 C oracle on fresh seeds, 64 envs x 34 steps: same bars, per tensor
  * 100-step rollouts are chaotic at stiff contacts (hard part 3): there the bar is
    statistical -- median over envs of the per-env pose error < 1e-5, 90th percentile < 1e-3 -- plus exact
    size-independent properties (unit quaternions, velocity clamps, finiteness, batch-composition invariance).
"""
import numpy as np
import pytest
import torch

from helpers import GRAD_LEAD, INPUT_NAMES, golden_inputs, load_golden, relmax

pytestmark = pytest.mark.gpu

FWD = ("q_init", "qd_init", "torques", "res_f", "refs", "target_ke", "target_kd", "body_inv_mass", "body_inertia", "body_inv_inertia")
BWD = ("q_init", "qd_init", "torques", "refs", "target_ke", "target_kd", "body_inv_mass", "body_inertia", "body_inv_inertia")
GRADS = FWD


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "run with -m gpu on a GPU box"
    return torch.device("cuda:0")


def gpu_rollout(dm, inp, dev, backward=True, keep_traj=False):
    from diffphys_amd import dp_model

    bs = inp["q_init"].size // dm.nq
    T, f2s = inp["nsteps"], inp["frame2step"]
    t = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=np.float32)).to(dev) for k in INPUT_NAMES + ("adj_pos", "adj_vel")}
    fos = list(f2s)
    pos, vel, grf, jaf, ws = dm.rollout_forward(bs, T, inp["dt"], *[t[k] for k in FWD], frame2step=fos)
    out = dict(wp_pos=pos.cpu().numpy(), wp_vel=vel.cpu().numpy(), grf=grf.cpu().numpy(), jaf=jaf.cpu().numpy())
    if backward:
        g = dm.rollout_backward(bs, T, inp["dt"], *[t[k] for k in BWD], fos, ws, t["adj_pos"], t["adj_vel"])
        out["grads"] = {k: v.cpu().numpy() for k, v in g.items()}
    if keep_traj:  # the states, total wrenches and clamp masks the kernel saved for its adjoint
        bq, bqd, bf, mask = dm.saved_trajectory(ws, bs, T)
        out["traj"] = dict(states_q=bq.cpu().numpy(), states_qd=bqd.cpu().numpy(), states_f=bf.cpu().numpy(), clamp=mask.cpu().numpy())
    return out


@pytest.mark.parametrize("name,segw", [("laikago", 0), ("laikago", 32), ("laikago", 64), ("human", 0), ("human", 64), ("quad", 0), ("quad", 64)])
def test_golden_one_frame_interval(name, segw, dev):
    from diffphys_amd import hip_backend, robots

    g = load_golden(name)
    inp = golden_inputs(g)
    dm = hip_backend.DeviceModel(robots.load_template(name))
    if segw:
        dm.set_segment_width(segw)
    out = gpu_rollout(dm, inp, dev)
    # bars = ~5x what round 3's kernels measure on these fixtures (poses 6e-7, twists 2e-5, wrenches 1e-4, gradients 4e-5);
    # rounds 1-2 had 5e-5 / 2e-3 / 5e-3 / 2e-2
    assert relmax(out["wp_pos"], g["wp_pos"]) < 5e-6
    assert relmax(out["wp_vel"], g["wp_vel"]) < 1e-4
    assert relmax(out["grf"], g["grf"]) < 5e-4 and relmax(out["jaf"], g["jaf"]) < 5e-4
    for k in GRADS:
        assert np.isfinite(out["grads"][k]).all(), k
        assert relmax(out["grads"][k].reshape(g["grad_" + k].shape), g["grad_" + k]) < 2e-4, k
    # FK fixture
    bq, bqd = dm.fk_forward(torch.from_numpy(g["fk_joint_q"]).to(dev), torch.from_numpy(g["fk_joint_qd"]).to(dev))
    assert relmax(bq.cpu().numpy(), g["fk_body_q"]) < 2e-6 and relmax(bqd.cpu().numpy(), g["fk_body_qd"]) < 2e-6
    gq, gqd = dm.fk_backward(torch.from_numpy(g["fk_joint_q"]).to(dev), torch.from_numpy(g["fk_joint_qd"]).to(dev),
                             torch.from_numpy(g["fk_adj_q"]).to(dev), torch.from_numpy(g["fk_adj_qd"]).to(dev))
    # pd_fk_backward returns the gradients with ForwardKinematics.backward's post-processing (values above 1 -> 1, dp_model.py:1110,1123)
    assert relmax(gq.cpu().numpy(), np.minimum(g["fk_grad_q"], 1.0)) < 5e-6 and relmax(gqd.cpu().numpy(), np.minimum(g["fk_grad_qd"], 1.0)) < 5e-6
    assert (g["fk_grad_q"] > 1.0).any(), "the fixture must exercise the clamp"


@pytest.mark.parametrize("name,bs", [("laikago", 64), ("human", 33), ("quad", 50)])
def test_vs_c_oracle_fresh_seed(name, bs, dev, oracle_libs):
    from diffphys_amd import hip_backend, robots, synth
    from oracle.ref_c import RefC

    tpl = robots.load_template(name)
    T = 34
    inp = synth.make_inputs(tpl, name, bs=bs, nsteps=T, seed=77, penetration=0.003)
    rng = np.random.RandomState(1)
    inp["torques"] = (rng.randn(*inp["torques"].shape) * 0.5).astype(np.float32)
    inp["res_f"] = (rng.randn(*inp["res_f"].shape) * 0.5).astype(np.float32)
    inp["qd_init"] = (rng.randn(*inp["qd_init"].shape) * 0.1).astype(np.float32)
    out = gpu_rollout(hip_backend.DeviceModel(tpl), inp, dev)
    # the float64 oracle is the authority; the fp32 oracle evaluates the revolute twist angle literally (2 acos(twist.w): one ulp
    # of twist.w is 7e-4 rad at a small angle), the kernel through atan2 (pd_math.h twist_angle) -- for Laikago the kernel is
    # the closer of the two to float64, so it is no longer compared with the fp32 oracle's gradients
    rc = RefC(tpl, np.float64)
    st = rc.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    gr = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
    rc32 = RefC(tpl, np.float32)
    st32 = rc32.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    g32 = rc32.rollout_backward(st32, inp["adj_pos"], inp["adj_vel"])
    assert np.abs(st["grf"]).max() > 1.0, "contacts must be active in this test"
    # (measured over three seeds: poses <= 2e-6, twists <= 5.3e-4 (quad), wrenches <= 1.7e-4)
    assert relmax(out["wp_pos"], st["wp_pos"]) < 1e-5 and relmax(out["wp_vel"], st["wp_vel"]) < 2e-3
    assert relmax(out["grf"], st["grf"]) < 1e-3 and relmax(out["jaf"], st["jaf"]) < 1e-3
    for k in GRADS:
        e, e32 = relmax(out["grads"][k].reshape(gr[k].shape), gr[k]), relmax(g32[k], gr[k])
        print("%s %-18s kernel vs float64 %.1e   fp32 oracle vs float64 %.1e" % (name, k, e, e32))
        assert e < max(2e-2, 2 * e32), k


def test_long_rollout_statistics_and_invariants(dev, oracle_libs):
    """BASELINE config C2 shape (Laikago mi-pace, 256 envs x 100 steps) against the fp32 C oracle."""
    from diffphys_amd import hip_backend, robots, synth
    from oracle.ref_c import RefC

    tpl = robots.load_template("laikago")
    bs, T = 256, 100
    inp = synth.make_inputs(tpl, "laikago", bs=bs, nsteps=T, seed=4, penetration=0.002)
    out = gpu_rollout(hip_backend.DeviceModel(tpl), inp, dev)
    st = RefC(tpl, np.float32).rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    F = len(inp["frame2step"])
    e = np.abs(out["wp_pos"].astype(np.float64) - st["wp_pos"]).reshape(F, bs, -1).max((0, 2))
    assert np.median(e) < 1e-5 and np.percentile(e, 90) < 1e-3, (np.median(e), np.percentile(e, 90))
    q = out["wp_pos"][..., 3:]
    assert np.abs(np.linalg.norm(q, axis=-1) - 1).max() < 1e-5          # r1 = normalize(...)  (integrator_euler.py:72)
    assert np.abs(out["wp_vel"]).max() <= 10.0                           # clamps (:78-88)
    assert all(np.isfinite(v).all() for v in out["grads"].values())
    assert np.abs(out["grads"]["refs"].reshape(T, bs, 18)[:, :, :6]).max() == 0  # FREE joint reads no dof (:382)


@pytest.mark.parametrize("family", [1, 2], ids=["lane-per-body", "quad-lane"])
def test_full_size_properties_and_batch_invariance(family, dev):
    """Headline size (4096 envs x 100 steps): finiteness, clamps, and an env's result does not depend on
    which other envs share its wavefront / launch (run a 37-env prefix alone, compare bit for bit) -- WITHIN a kernel family: the
    lane-per-body kernels and the quad-lane kernels (pd_model_set_kernel_family; by default the batch size picks between them, so a
    batch of <= 512 / 1 024 envs and the same envs inside a larger batch differ by fp32 round-off) each reproduce their own bits."""
    from diffphys_amd import hip_backend, robots, synth

    tpl = robots.load_template("laikago")
    bs, T, sub = 4096, 100, 37
    inp = synth.make_inputs(tpl, "laikago", bs=bs, nsteps=T, seed=9, seqs=("mi-trot", "mi-spin"))
    dm = hip_backend.DeviceModel(tpl)
    dm.set_kernel_family(family)
    full = gpu_rollout(dm, inp, dev)
    assert all(np.isfinite(full[k]).all() for k in ("wp_pos", "wp_vel", "grf", "jaf"))
    assert all(np.isfinite(v).all() for v in full["grads"].values())
    assert np.abs(full["wp_vel"]).max() <= 10.0
    nb, nq, nqd = 13, 19, 18
    part = dict(inp)
    cut = lambda a, per, lead=(): np.ascontiguousarray(a.reshape(lead + (bs, per))[..., :sub, :].reshape(lead + (sub * per,)))
    part.update(q_init=cut(inp["q_init"], nq), qd_init=cut(inp["qd_init"], nqd), torques=cut(inp["torques"], nqd, (T,)),
                refs=cut(inp["refs"], nqd, (T,)), target_ke=cut(inp["target_ke"], nqd), target_kd=cut(inp["target_kd"], nqd),
                body_mass=cut(inp["body_mass"], nb), body_inv_mass=cut(inp["body_inv_mass"], nb),
                res_f=np.ascontiguousarray(inp["res_f"].reshape(T, bs, nb, 6)[:, :sub].reshape(T, -1, 6)),
                body_inertia=np.ascontiguousarray(inp["body_inertia"].reshape(bs, nb, 3, 3)[:sub].reshape(-1, 3, 3)),
                body_inv_inertia=np.ascontiguousarray(inp["body_inv_inertia"].reshape(bs, nb, 3, 3)[:sub].reshape(-1, 3, 3)),
                adj_pos=np.ascontiguousarray(inp["adj_pos"].reshape(-1, bs, nb, 7)[:, :sub].reshape(-1, sub * nb, 7)),
                adj_vel=np.ascontiguousarray(inp["adj_vel"].reshape(-1, bs, nb, 6)[:, :sub].reshape(-1, sub * nb, 6)))
    small = gpu_rollout(dm, part, dev)
    F = len(inp["frame2step"])
    assert np.array_equal(small["wp_pos"], full["wp_pos"].reshape(F, bs, nb, 7)[:, :sub].reshape(F, -1, 7))
    assert np.array_equal(small["grads"]["q_init"], full["grads"]["q_init"].reshape(bs, nq)[:sub].reshape(-1))
    for _ in range(3):  # run-to-run: the wave pairs hand over through polled LDS words, a race would show up as changing bits
        again = gpu_rollout(dm, inp, dev)
        assert all(np.array_equal(again[k], full[k]) for k in ("wp_pos", "wp_vel", "grf", "jaf"))
        assert all(np.array_equal(again["grads"][k], full["grads"][k]) for k in full["grads"])


# (1, 50, [0, 33]) is BASELINE config C1: Laikago mi-pace, ONE env, 50-step rollout, frames 0 and 33 -- run on the HIP path
# (the oracle is only the checker)
@pytest.mark.parametrize("bs,T,f2s", [(1, 1, [0]), (3, 2, [0, 1]), (17, 5, [4]), (5, 7, [0, 3, 6]), (1, 50, [0, 33])])
def test_edge_shapes(bs, T, f2s, dev, oracle_libs):
    from diffphys_amd import hip_backend, robots, synth
    from oracle.ref_c import RefC

    tpl = robots.load_template("laikago")
    inp = synth.make_inputs(tpl, "laikago", bs=bs, nsteps=T, seed=3, penetration=0.002)
    F = len(f2s)
    rng = np.random.RandomState(0)
    inp["frame2step"] = f2s
    inp["adj_pos"] = (rng.randn(F, bs * 13, 7) * 1e-3).astype(np.float32)
    inp["adj_vel"] = (rng.randn(F, bs * 13, 6) * 1e-3).astype(np.float32)
    out = gpu_rollout(hip_backend.DeviceModel(tpl), inp, dev)
    rc = RefC(tpl, np.float32)
    st = rc.rollout_forward(inp, T, f2s, inp["dt"])
    gr = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
    assert relmax(out["wp_pos"], st["wp_pos"]) < 1e-5
    # (target_ke is left out here: with refs == current angles its gradient is fp32 round-off of (q - target) ~ 1e-10)
    for k in ("q_init", "qd_init", "refs", "res_f", "body_inertia"):
        ref = gr[k]
        if np.abs(ref).max() > 0:
            assert relmax(out["grads"][k].reshape(ref.shape), ref) < 2e-2, k
        else:
            assert np.abs(out["grads"][k]).max() == 0


def test_zero_angle_singularity_is_finite(dev):
    """All joint angles exactly 0 (twist.w == 1 in fp32): guarded acos adjoint, no NaN/inf anywhere."""
    from diffphys_amd import hip_backend, robots, synth

    tpl = robots.load_template("laikago")
    inp = synth.make_inputs(tpl, "laikago", bs=4, nsteps=6, seed=0, steps_per_frame=2)
    inp["q_init"].reshape(4, -1)[0, 7:] = 0.0
    inp["refs"][:] = 0.0
    out = gpu_rollout(hip_backend.DeviceModel(tpl), inp, dev)
    assert all(np.isfinite(v).all() for v in out["grads"].values())


@pytest.mark.parametrize("name", ["laikago", "human"])
def test_gradient_post_processing_inside_the_kernels(name, dev):
    """remove_nan is applied where the adjoint STORES (diffphys/dp_model.py:1294-1384 with dp_utils.py:43-57,
    clip=False): the raw gradients of the C ABI -- no torch pass behind them -- carry exact zeros where a NaN arose and keep inf.
      * NaN injected through res_f (one body of env 1, step 2): that env's forward state is NaN from there on, its gradients are
        NaN before the scrub; every gradient of the call must be finite-or-inf and env 1's must be exactly 0 where they were NaN,
        while the other envs' gradients equal those of the clean rollout bit for bit;
      * an infinite upstream gradient (adj_vel of one component at the last frame): the gradient of the residual force on that
        component is +inf and must survive; the NaNs that inf * 0 makes elsewhere are zeros.
    The FK boundary (NaN -> 0, values above 1 -> 1; dp_model.py:1109-1123) is checked on pd_fk_backward's raw output."""
    from diffphys_amd import hip_backend, robots, synth

    tpl = robots.load_template(name)
    nb = int(tpl["nb"])
    bs, T = 5, 6
    inp = synth.make_inputs(tpl, name, bs=bs, nsteps=T, seed=3, steps_per_frame=5, penetration=0.002)
    dm = hip_backend.DeviceModel(tpl)
    clean = gpu_rollout(dm, inp, dev)
    assert all(np.isfinite(v).all() for v in clean["grads"].values())
    bad = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in inp.items()}
    bad["res_f"].reshape(T, bs, nb, 6)[2, 1, 3, 4] = np.nan
    out = gpu_rollout(dm, bad, dev)
    for k, lead in GRAD_LEAD.items():
        g, c = out["grads"][k], clean["grads"][k]
        assert not np.isnan(g).any(), k
        ge = g.reshape(T, bs, -1) if lead else g.reshape(bs, -1)
        ce = c.reshape(T, bs, -1) if lead else c.reshape(bs, -1)
        others = [e for e in range(bs) if e != 1]
        assert np.array_equal(ge[:, others] if lead else ge[others], ce[:, others] if lead else ce[others]), k
    # steps 0 .. 2 of env 1 see only NaN adjoints (the state after step 2 is NaN): exact zeros
    assert np.all(out["grads"]["res_f"].reshape(T, bs, -1)[:3, 1] == 0.0) and np.all(out["grads"]["q_init"].reshape(bs, -1)[1] == 0.0)
    # an infinite seed survives as inf
    seeded = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in inp.items()}
    F = len(inp["frame2step"])
    assert inp["frame2step"][-1] == T - 1 or inp["frame2step"][-1] == T
    seeded["adj_vel"].reshape(F, bs, nb, 6)[F - 1, 2, 0, 4] = np.inf   # linear velocity y of the root of env 2
    out = gpu_rollout(dm, seeded, dev)
    assert all(not np.isnan(v).any() for v in out["grads"].values())
    assert np.isinf(out["grads"]["res_f"]).any(), "the infinite gradient was scrubbed"
    # FK boundary: raw pd_fk_backward output is NaN-free and clamped from above at 1
    jq = torch.from_numpy(inp["q_init"].reshape(bs, -1)).to(dev)
    jqd = torch.zeros(bs, int(tpl["nqd"]), device=dev)
    aq = torch.full((bs, nb, 7), 5.0, device=dev)
    aq[0, 0, 0] = float("nan")
    gq, gqd = dm.fk_backward(jq, jqd, aq, torch.full((bs, nb, 6), 5.0, device=dev))
    gq, gqd = gq.cpu().numpy(), gqd.cpu().numpy()
    assert not np.isnan(gq).any() and not np.isnan(gqd).any() and gq.max() <= 1.0 and gqd.max() <= 1.0 and gq.max() == 1.0
    assert gq.min() < -1.0, "only an UPPER clamp: large negative gradients pass"


def test_autograd_boundaries(dev, oracle_libs):
    """ForwardWarp / ForwardKinematics keep the reference's contract: shapes, side outputs, grad post-processing."""
    from diffphys_amd import dp_model, robots, synth

    tpl = robots.load_template("quad")
    bs, T = 6, 34
    inp = synth.make_inputs(tpl, "quad", bs=bs, nsteps=T, seed=5, penetration=0.002)

    class Host:
        pass

    h = Host()
    h.env = robots.env_from_template("quad", bs, device=dev)
    h.num_envs, h.steps_idx, h.frame2step, h.dt = bs, range(T), inp["frame2step"], inp["dt"]
    t = {k: torch.from_numpy(inp[k]).to(dev).requires_grad_(True) for k in INPUT_NAMES}
    pos, vel = dp_model.ForwardWarp.apply(*[t[k] for k in INPUT_NAMES], h)
    assert pos.shape == (2, bs * 26, 7) and vel.shape == (2, bs * 26, 6)
    assert len(h.grfs) == 2 and h.grfs[0].shape == (bs * 26, 6) and len(h.jafs) == 2
    assert len(h.sim_trajs) == 2 and h.sim_trajs[0].shape == (26, 7)
    (pos.sum() + vel.sum()).backward()
    for k in INPUT_NAMES:
        assert t[k].grad is not None and t[k].grad.shape == t[k].shape and torch.isfinite(t[k].grad).all(), k
    assert t["body_mass"].grad.abs().max() == 0
    # ForwardKinematics: [F,bs,nq] -> [bs,F,nb,7]; grads clamped from above at 1 (dp_model.py:1110,1123)
    F = 3
    rq = torch.from_numpy(inp["q_init"]).to(dev).view(1, bs, -1).repeat(F, 1, 1).clone().requires_grad_(True)
    rqd = torch.zeros(F, bs, 81, device=dev, requires_grad=True)
    bq, bqd, bq_np = dp_model.ForwardKinematics.apply(rq, rqd, h.env)
    assert bq.shape == (bs, F, 26, 7) and bqd.shape == (bs, F, 26, 6) and len(bq_np) == F and bq_np[0].shape == (26, 7)
    (bq.sum() * 100 + bqd.sum()).backward()
    assert rq.grad.shape == rq.shape and rq.grad.max() <= 1.0 and rq.grad.min() < -1.0
    # the rollout's state 0 is eval_fk of q_init
    assert torch.allclose(pos[0].view(bs, 26, 7), bq[:, 0], atol=1e-6)


def test_errors_are_loud(dev):
    from diffphys_amd import hip_backend, robots

    dm = hip_backend.DeviceModel(robots.load_template("laikago"))
    z = torch.zeros(19, device=dev)
    with pytest.raises(ValueError):
        dm.fk_forward(z, torch.zeros(17, device=dev))          # wrong size
    with pytest.raises(TypeError):
        dm.fk_forward(z.double(), torch.zeros(18, device=dev))  # wrong dtype
    with pytest.raises(ValueError):
        dm.fk_forward(z.cpu(), torch.zeros(18))                 # host memory
    with pytest.raises(RuntimeError):
        dm.set_segment_width(8)
    tpl = dict(robots.load_template("laikago"))
    tpl["joint_type"] = tpl["joint_type"].copy()
    tpl["joint_type"][3] = 0  # prismatic: the reference's joint kernel does not handle it either
    with pytest.raises(RuntimeError):
        hip_backend.DeviceModel(tpl)


def test_config_c3_human_1024(dev, oracle_libs):
    """BASELINE config C3: human URDF (19 bodies, 18 compound joints), 1024 envs x 100 steps, fwd + adjoint,
    against the fp32 C oracle: statistical pose bar over 100 steps, finite gradients, gradient parity per env (median)."""
    from diffphys_amd import hip_backend, robots, synth
    from oracle.ref_c import RefC

    tpl = robots.load_template("human")
    bs, T = 1024, 100
    inp = synth.make_inputs(tpl, "human", bs=bs, nsteps=T, seed=12, penetration=0.002)
    out = gpu_rollout(hip_backend.DeviceModel(tpl), inp, dev)
    rc = RefC(tpl, np.float32)
    st = rc.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    gr = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
    F = len(inp["frame2step"])
    e = np.abs(out["wp_pos"].astype(np.float64) - st["wp_pos"]).reshape(F, bs, -1).max((0, 2))
    assert np.median(e) < 1e-5 and np.percentile(e, 90) < 1e-3, (np.median(e), np.percentile(e, 90))
    assert all(np.isfinite(v).all() for v in out["grads"].values())
    g, r = out["grads"]["q_init"].reshape(bs, -1), gr["q_init"].reshape(bs, -1)
    per_env = np.abs(g - r).max(1) / (np.abs(r).max(1) + 1e-12)
    assert np.median(per_env) < 1e-3, np.median(per_env)


def test_config_c5_quad_8192_gradcheck(dev, oracle_libs):
    """BASELINE config C5: AI4Animation quadruped (26 bodies, 25 compound joints), contact-rich, 8192 envs.
    Gradient check of every input against the float64 C oracle, per env and per tensor ON THAT ENV'S OWN SCALE (round 4; round 3
    divided by the batch-wide maximum).  The config's rtol 1e-4
    is the median bar at BOTH horizons -- 4 steps and the config's own T = 34 (one frame interval) -- with the per-env 99th
    percentile at 1e-2 for T = 34 (round 2 had 2e-2 / 1.0 there).  The envs above the median bar are the
    ones that sit on a contact edge (see test_config_size_gradients_every_tensor_per_env for the explained per-env version)."""
    from diffphys_amd import hip_backend, robots, synth
    from oracle.ref_c import RefC

    tpl = robots.load_template("quad")
    bs = 8192
    dm = hip_backend.DeviceModel(tpl)
    for T, tol, p99 in ((4, 1e-4, 5e-3), (34, 1e-4, 1e-2)):
        inp = synth.make_inputs(tpl, "quad", bs=bs, nsteps=T, seed=31, steps_per_frame=3 if T == 4 else 33, penetration=0.004)
        rng = np.random.RandomState(2)
        inp["qd_init"] = (rng.randn(*inp["qd_init"].shape) * 0.1).astype(np.float32)
        inp["torques"] = (rng.randn(*inp["torques"].shape) * 0.5).astype(np.float32)
        out = gpu_rollout(dm, inp, dev)
        rc = RefC(tpl, np.float64)
        st = rc.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
        gr = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
        assert np.abs(st["grf"]).max() > 10.0  # contact-rich
        # per env AND per tensor on that env's own scale (helpers.grad_env_errors; round 3 divided by the batch-wide maximum, so a
        # handful of envs with huge gradients set the scale -- VERDICT r3 weak #3)
        from helpers import grad_env_errors
        errs = grad_env_errors(out["grads"], gr, bs)
        for k in GRADS:
            if np.abs(gr[k]).max() == 0:
                continue
            per_env = errs[k]
            print("C5 T=%d %-18s median %.1e p99 %.1e max %.1e" % (T, k, np.median(per_env), np.percentile(per_env, 99), per_env.max()))
            assert np.median(per_env) < tol, (T, k, float(np.median(per_env)))
            assert np.percentile(per_env, 99) < p99, (T, k, float(np.percentile(per_env, 99)))


def test_phys_model_training_iterations(dev):
    """Row f3: the reference-shaped optimisation loop runs on the HIP rollout and the imitation loss goes down."""
    from diffphys_amd.dataloader import DataLoader
    from diffphys_amd.phys_model import phys_model

    sys_argv = ["--seqname", "mi-pace", "--urdf_template", "laikago", "--num_rounds", "1", "--iters_per_round", "16",
                "--logroot", "/tmp/pprdp_test_log/", "--logname", "t", "--num_envs", "8", "--frames_per_wdw", "4"]
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("pd_main", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                          "ppr-diffphys_amd", "main.py"))
    pd_main = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pd_main)
    opts = pd_main.get_opts(sys_argv)
    torch.manual_seed(0)
    np.random.seed(0)
    model = phys_model(opts, DataLoader(opts)).cuda()
    model.train()
    assert abs(float(model.global_q.detach()[1])) < 0.2  # ground offset from the FK of frame 0
    model.reinit_envs(opts["num_envs"], frames_per_wdw=opts["frames_per_wdw"])
    fs = torch.arange(8, device=model.device) * 3
    losses = []
    for it in range(12):
        model.set_progress(it)
        out = model.forward(frame_start=fs)
        model.backward(out["total_loss"])
        gd = model.update()
        losses.append(float(out["total_loss"].detach()))
        assert np.isfinite(losses[-1])
    assert len(gd) > 0 and all(torch.isfinite(v) for v in gd.values())
    assert np.mean(losses[-3:]) < np.mean(losses[:3]), losses
    q = model.query()
    assert q["sim_poses"].shape == (4, 13, 7) and q["sim_traj"].shape == (4, 3838, 3) and q["grf"].shape == (4, 8 * 13, 6)
    model.save_checkpoint(0)
    assert os.path.exists("/tmp/pprdp_test_log/mi-pace-t/ckpt_phys_latest.pth")
    # nothing between forward() and backward() may wait for the device (lazy env-0 copies, cached step->frame table,
    # sync-free reduce_loss / SE(3) helpers, NaN check deferred to update()): a forward + backward pair enqueues without a
    # single host synchronisation -- checked by running it with synchronisation turned into an error
    torch.cuda.synchronize()
    prev = torch.cuda.get_sync_debug_mode()
    torch.cuda.set_sync_debug_mode("error")
    try:
        noise = None
        out = model.forward(frame_start=fs, q_init_noise=torch.zeros(8 * 19, device=model.device))
        model.backward(out["total_loss"])
    finally:
        torch.cuda.set_sync_debug_mode(prev)
    model.update()


@pytest.mark.gpu
def test_generic_joint_kernel_on_toy_robot(dev, oracle_libs, tmp_path):
    """A URDF mixing free + revolute + compound + fixed joints and box / sphere / mesh / capsule contacts goes through the
    model compiler and the GENERIC kernel instantiation (all joint types), which none of the three shipped robots uses."""
    from test_host import OBJ, URDF
    from diffphys_amd import hip_backend, sim
    from diffphys_amd.import_urdf import parse_urdf
    from helpers import build_template
    from oracle.ref_c import RefC

    (tmp_path / "toy.urdf").write_text(URDF)
    (tmp_path / "tet.obj").write_text(OBJ)
    b = sim.ModelBuilder()
    parse_urdf(str(tmp_path / "toy.urdf"), b, xform=sim.transform((0, 0.5, 0), sim.quat_identity()), floating=True, density=1000.0,
               armature=0.01, stiffness=220.0, damping=2.0, shape_ke=1e4, shape_kd=10.0, shape_kf=1e2, shape_mu=0.7, limit_ke=50.0, limit_kd=1.0)
    tpl = build_template(b, attach_ke=8000.0, attach_kd=200.0)
    nb, nq, nqd = int(tpl["nb"]), int(tpl["nq"]), int(tpl["nqd"])
    assert sorted(set(int(t) for t in tpl["joint_type"])) == [1, 3, 4, 5]
    bs, T = 9, 30
    rng = np.random.RandomState(0)
    q = np.tile(tpl["joint_q"].astype(np.float64), (bs, 1))
    q[:, 1] = 0.13 + rng.rand(bs) * 0.02                 # arm sphere and leg mesh are in the ground
    yaw = rng.uniform(-0.3, 0.3, bs)
    q[:, 3:7] = np.stack([np.sin(yaw / 2) * 0.1, np.sin(yaw / 2), 0 * yaw, np.cos(yaw / 2)], -1)
    q[:, 3:7] /= np.linalg.norm(q[:, 3:7], axis=1, keepdims=True)
    q[:, 7:] = rng.uniform(-0.4, 0.4, (bs, nq - 7))
    q[:, 8] = 1.6 + rng.rand(bs) * 0.2                   # compound x-angle beyond its +1.5 limit: limit force path
    mass = np.tile(tpl["body_mass"].astype(np.float64), bs)
    inertia = np.tile(tpl["body_inertia"].astype(np.float64), (bs, 1, 1))
    ke = np.tile(np.r_[np.zeros(6), np.full(nqd - 6, 60.0)], bs)
    inp = dict(q_init=q.reshape(-1), qd_init=rng.randn(bs * nqd) * 0.2, torques=rng.randn(T, bs * nqd) * 0.3,
               res_f=rng.randn(T, bs * nb, 6) * 0.3, refs=rng.uniform(-0.3, 0.3, (T, bs * nqd)), target_ke=ke, target_kd=ke * 0.02,
               body_mass=mass, body_inv_mass=1 / mass, body_inertia=inertia, body_inv_inertia=np.linalg.inv(inertia),
               adj_pos=rng.randn(3, bs * nb, 7) * 1e-3, adj_vel=rng.randn(3, bs * nb, 6) * 1e-3)
    inp = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in inp.items()}
    inp.update(frame2step=[0, 14, 29], nsteps=T, dt=5e-4)
    dm = hip_backend.DeviceModel(tpl)
    out = gpu_rollout(dm, inp, dev)
    # The FIXED joint's angular error is evaluated scale-invariantly by the kernels (pd_math.h fixed_ang_h: NAMED DEVIATION in evaluation,
    # like the revolute twist angle -- the literal normalize(v) * 2 acos(w) turns the 1e-7 norm error of fp32 quaternions into +-9e-4 rad
    # at the joint's operating point, for ANY evaluator): both C oracles take the same form with set_twist_eval(True).  Rounds 1-3 compared
    # against the literal fp32 oracle here and had to allow 15 % in the joint forces and medians only in the gradients.
    from helpers import own_trajectory_check

    rc, rc64 = RefC(tpl, np.float32), RefC(tpl, np.float64)
    try:
        rc.set_twist_eval(True); rc64.set_twist_eval(True)
        st = rc.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
        st64 = rc64.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    finally:
        rc.set_twist_eval(False); rc64.set_twist_eval(False)
    assert np.abs(st["grf"]).max() > 5.0 and np.abs(st["jaf"]).max() > 1.0
    for k, floor in (("wp_pos", 2e-5), ("wp_vel", 2e-3), ("grf", 5e-3), ("jaf", 5e-3)):
        e, y = relmax(out[k], st64[k]), relmax(st[k], st64[k])
        print("toy robot %s: kernels vs float64 %.1e, fp32 oracle vs float64 %.1e" % (k, e, y))
        assert e < max(floor, 3.0 * y), (k, e, y)
    assert all(np.isfinite(v).all() for v in out["grads"].values())
    own = own_trajectory_check(dm, tpl, inp, dev, abs_floor=1e-8)   # (float64 reference in the same scale-invariant form: helpers.py)
    print("toy robot: own trajectory worst env %.1e, median %.1e (plain fp32: %.1e)" % (own["worst"].max(), np.median(own["worst"]), np.median(own["fp32_atan2"])))
    assert (own["worst"] <= np.maximum(1e-3, 3.0 * own["fp32_atan2"])).all(), (own["worst"], own["fp32_atan2"])


@pytest.mark.gpu
def test_revolute_chain_jointed_to_the_world(dev, oracle_libs):
    """The specialised kernels (one joint type) assume that every joint which is not FREE hangs on a body; a model with a
    joint to the WORLD must take the generic instantiation instead (pd_host.hip) and still match the oracle.  The model: one
    Laikago leg (hip motor, upper, lower leg) hung from a point above the ground by its hip joint -- revolute joints only,
    no floating base -- with the lower leg dipping into the ground."""
    from diffphys_amd import hip_backend, robots
    from oracle.ref_c import RefC

    full = robots.load_template("laikago")
    keep = [1, 2, 3]
    tpl = {k: v for k, v in full.items()}
    for k in ("joint_type", "joint_X_p", "joint_X_c", "joint_axis", "body_com", "body_mass", "body_inertia", "body_names", "shape_materials"):
        tpl[k] = np.ascontiguousarray(full[k][keep])
    tpl["joint_parent"] = np.array([-1, 0, 1], np.int32)
    tpl["joint_q_start"] = np.array([0, 1, 2], np.int32)
    tpl["joint_qd_start"] = np.array([0, 1, 2], np.int32)
    tpl["nb"], tpl["nq"], tpl["nqd"] = np.int32(3), np.int32(3), np.int32(3)
    dof = [6, 7, 8]  # the three joints' dofs in the full robot
    for k in ("joint_target_ke", "joint_target_kd", "joint_limit_lower", "joint_limit_upper", "joint_limit_ke", "joint_limit_kd"):
        tpl[k] = np.ascontiguousarray(full[k][dof])
    tpl["joint_q"] = np.zeros(3, np.float32)
    X_p = tpl["joint_X_p"].copy()
    X_p[0, :3] = [0.0, 0.36, 0.0]  # hip joint anchored in the world, 36 cm up: the lower leg reaches the ground
    tpl["joint_X_p"] = X_p
    sel = np.isin(full["contact_body"], keep)
    remap = {1: 0, 2: 1, 3: 2}
    tpl["contact_body"] = np.array([remap[int(b)] for b in full["contact_body"][sel]], np.int32)
    for k in ("contact_point", "contact_dist", "contact_material"):
        tpl[k] = np.ascontiguousarray(full[k][sel])
    tpl["contact_material"] = np.zeros_like(tpl["contact_material"])
    nb, nq, nqd = 3, 3, 3
    bs, T = 11, 40
    rng = np.random.RandomState(4)
    mass = np.tile(tpl["body_mass"].astype(np.float64), bs)
    inertia = np.tile(tpl["body_inertia"].astype(np.float64), (bs, 1, 1))
    inp = dict(q_init=rng.uniform(-0.5, 0.5, bs * nq), qd_init=rng.randn(bs * nqd) * 0.3, torques=rng.randn(T, bs * nqd) * 0.3,
               res_f=rng.randn(T, bs * nb, 6) * 0.3, refs=rng.uniform(-0.4, 0.4, (T, bs * nqd)), target_ke=np.full(bs * nqd, 220.0),
               target_kd=np.full(bs * nqd, 2.0), body_mass=mass, body_inv_mass=1 / mass, body_inertia=inertia,
               body_inv_inertia=np.linalg.inv(inertia), adj_pos=rng.randn(3, bs * nb, 7) * 1e-3, adj_vel=rng.randn(3, bs * nb, 6) * 1e-3)
    inp = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in inp.items()}
    inp.update(frame2step=[0, 19, 39], nsteps=T, dt=5e-4)
    dm = hip_backend.DeviceModel(tpl)
    out = gpu_rollout(dm, inp, dev)
    rc = RefC(tpl, np.float32)
    st = rc.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    gr = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
    assert np.abs(st["grf"]).max() > 1.0, "the leg must touch the ground"
    assert relmax(out["wp_pos"], st["wp_pos"]) < 2e-5 and relmax(out["wp_vel"], st["wp_vel"]) < 1e-3
    assert relmax(out["grf"], st["grf"]) < 2e-3 and relmax(out["jaf"], st["jaf"]) < 2e-3
    for k in GRADS:
        assert np.isfinite(out["grads"][k]).all(), k
        assert relmax(out["grads"][k].reshape(gr[k].shape), gr[k]) < 2e-2, k


@pytest.mark.gpu
def test_rotated_child_joint_frame_takes_the_generic_kernel(dev, oracle_libs):
    """The single-joint-type kernels also assume identity child joint frames (joint_X_c rotations; every robot of the reference
    has them) and drop the compound joint's products with that frame from the adjoint.  A model with a ROTATED child frame must
    take the generic instantiation (pd_host.hip) and match the oracle: the human with two of its child frames turned."""
    from diffphys_amd import hip_backend, robots, synth
    from oracle.ref_c import RefC

    name = "human"
    tpl = dict(robots.load_template(name))
    X_c = np.array(tpl["joint_X_c"], np.float32).copy()
    for body, (ax, ang) in ((4, ((1.0, 0.0, 0.0), 0.3)), (11, ((0.0, 0.6, 0.8), -0.5))):
        X_c[body, 3:6] = np.asarray(ax, np.float32) * np.sin(ang / 2)
        X_c[body, 6] = np.cos(ang / 2)
    tpl["joint_X_c"] = X_c
    inp = synth.make_inputs(tpl, name, bs=6, nsteps=20, seed=3, steps_per_frame=7, penetration=0.003)
    out = gpu_rollout(hip_backend.DeviceModel(tpl), inp, dev)
    rc = RefC(tpl, np.float32)
    st = rc.rollout_forward(inp, inp["nsteps"], inp["frame2step"], inp["dt"])
    gr = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
    # the same model with identity frames gives different numbers (the turned frames matter) ...
    plain = gpu_rollout(hip_backend.DeviceModel(robots.load_template(name)), inp, dev)
    assert relmax(plain["jaf"], out["jaf"]) > 1e-3
    # ... and the generic kernel agrees with the oracle
    assert relmax(out["wp_pos"], st["wp_pos"]) < 5e-5 and relmax(out["wp_vel"], st["wp_vel"]) < 2e-3
    assert relmax(out["grf"], st["grf"]) < 5e-3 and relmax(out["jaf"], st["jaf"]) < 5e-3
    for k in GRADS:
        assert np.isfinite(out["grads"][k]).all(), k
        assert relmax(out["grads"][k].reshape(gr[k].shape), gr[k]) < 2e-2, k


def test_empty_batch_and_graph_capture(dev):
    """Empty inputs return empty outputs; and the launch path neither allocates through HIP nor synchronises, so the whole
    device side of an optimisation iteration -- FK, rollout, fused frame losses, adjoint rollout, FK adjoint -- can be
    captured into ONE HIP graph (torch.cuda.graphs) and replayed, repeatedly, with bit-identical results."""
    from diffphys_amd import dp_model, hip_backend, robots, synth

    tpl = robots.load_template("laikago")
    dm = hip_backend.DeviceModel(tpl)
    z = lambda *s: torch.zeros(*s, device=dev)
    fos = [0, 4]
    pos, vel, grf, jaf, ws = dm.rollout_forward(0, 5, 5e-4, z(0), z(0), z(5, 0), z(5, 0, 6), z(5, 0), z(0), z(0), z(0), z(0, 3, 3), z(0, 3, 3),
                                                frame2step=fos)
    assert pos.shape == (2, 0, 7) and vel.shape == (2, 0, 6) and ws.numel() == 0
    bq, bqd = dm.fk_forward(z(0, 19), z(0, 18))
    assert bq.shape == (0, 13, 7)

    bs, T = 64, 34
    inp = synth.make_inputs(tpl, "laikago", bs=bs, nsteps=T, seed=1, penetration=0.002)
    t = {k: torch.from_numpy(inp[k]).to(dev) for k in INPUT_NAMES + ("adj_pos", "adj_vel")}
    fos = list(inp["frame2step"])
    F = len(inp["frame2step"])

    tq = torch.from_numpy(np.tile(inp["q_init"].reshape(bs, -1), (F, 1))).to(dev)   # target joint coordinates of the F frames
    tqd = torch.zeros(F * bs, 18, device=dev)

    def run():
        """what one optimisation iteration asks of the library: FK of the targets, rollout, fused pose loss of
        every frame (loss + both gradients in one launch), adjoint rollout seeded by the loss gradient, FK adjoint"""
        tgt_q, tgt_qd = dm.fk_forward(tq, tqd)
        o = dm.rollout_forward(bs, T, inp["dt"], *[t[k] for k in FWD], frame2step=fos)
        loss, g_pred, _ = hip_backend.se3_loss(o[0].view(F * bs, 13, 7), tgt_q, 0.1)
        lv, g_vel, _ = hip_backend.se3_loss(o[1].view(F * bs, 13, 6), tgt_qd, 0.1)
        g = dm.rollout_backward(bs, T, inp["dt"], *[t[k] for k in BWD], fos, o[4], g_pred.view(F, bs * 13, 7), g_vel.view(F, bs * 13, 6))
        gq, _ = dm.fk_backward(tq, tqd, g_pred, g_vel)
        return o[0], g["q_init"], loss, gq

    ref = [x.clone() for x in run()]
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()  # warm-up on the capture stream
        with torch.cuda.graph(graph, stream=side):
            cap = run()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):  # several replays: every one must reproduce the eager results bit for bit
        for x in cap:
            x.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(cap, ref))
    assert float(ref[2].abs().max()) > 0 and float(ref[1].abs().max()) > 0


@pytest.mark.parametrize("segw", [0, 64])
def test_many_contacts_overflow_paths(segw, dev, oracle_libs):
    """Robot dropped INTO the ground (chassis and legs hundreds of points deep): exercises the cooperative big-body
    tile cull, hit-list overflow with mid-sweep flushes, the hit-log 'did not fit' marker and the adjoint's re-cull
    fallback -- none of which a standing robot reaches.  Short horizon: the 500 N clamps make this regime violent."""
    from diffphys_amd import hip_backend, robots, synth
    from oracle.ref_c import RefC

    tpl = robots.load_template("laikago")
    bs, T = 7, 6
    inp = synth.make_inputs(tpl, "laikago", bs=bs, nsteps=T, seed=8, steps_per_frame=2)
    q = inp["q_init"].reshape(bs, -1)
    q[:, 1] -= np.linspace(0.25, 0.42, bs).astype(np.float32)   # chassis bottom well below y = 0 for the later envs
    rng = np.random.RandomState(0)
    inp["qd_init"] = (rng.randn(*inp["qd_init"].shape) * 0.3).astype(np.float32)  # tangential velocity => friction term
    dm = hip_backend.DeviceModel(tpl)
    if segw:
        dm.set_segment_width(segw)
    out = gpu_rollout(dm, inp, dev)
    rc = RefC(tpl, np.float32)
    st = rc.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    gr = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
    # how many candidates are active at step 0 in the deepest env (oracle-independent count from the fixture geometry)
    low = synth.fk_pose_np(tpl, q.astype(np.float64))
    cb = tpl["contact_body"].astype(np.int64)
    X = low[:, cb]
    y = X[..., 1] + synth._qrot(X[..., 3:], np.broadcast_to(tpl["contact_point"].astype(np.float64), X[..., :3].shape))[..., 1]
    nact = (y <= 0).sum(-1)
    assert nact.max() > 500 and nact.min() > 31, nact   # beyond the 31-entry hit log and the 128-entry hit list
    assert relmax(out["grf"], st["grf"]) < 1e-3 and relmax(out["wp_pos"], st["wp_pos"]) < 1e-4
    assert relmax(out["wp_vel"], st["wp_vel"]) < 5e-3 and relmax(out["jaf"], st["jaf"]) < 1e-2
    assert all(np.isfinite(v).all() for v in out["grads"].values())
    for k in ("q_init", "qd_init", "res_f", "refs", "body_inv_mass", "body_inertia"):
        assert relmax(out["grads"][k].reshape(gr[k].shape), gr[k]) < 2e-2, k


@pytest.mark.gpu
@pytest.mark.parametrize("family", [1, 2], ids=["lane-per-body", "quad-lane"])
def test_speculative_sweep_is_redone_when_bodies_outrun_their_margin(family, dev, oracle_libs):
    """The wave-specialised forward culls ground contacts one epoch ahead with a per-body sink margin and must fall back
    to the exact sweep when a body moves further than that.  Downward kicks of 3000 m/s^2 on every body from step 3 on
    (|dv_y| = 1.5 m/s per step: the margin of the epoch's first state is outrun within two steps) force that path;
    results must still match the oracle, and must not depend on how envs share a wave."""
    from diffphys_amd import hip_backend, robots, synth
    from oracle.ref_c import RefC

    tpl = robots.load_template("laikago")
    bs, T = 24, 26
    inp = synth.make_inputs(tpl, "laikago", bs=bs, nsteps=T, seed=21, steps_per_frame=5, penetration=0.003)
    nb = int(tpl["nb"])
    rf = inp["res_f"].reshape(T, bs, nb, 6)
    mass = inp["body_mass"].reshape(bs, nb)
    kick = np.zeros((T, bs), np.float32)
    kick[3:16, ::2] = 3000.0      # every other env: its wave-mates stay calm, the wave must still redo the sweep
    kick[8:20, 1::4] = 1500.0
    rf[..., 4] -= kick[:, :, None] * mass[None]
    dm = hip_backend.DeviceModel(tpl)
    dm.set_kernel_family(family)   # (both families speculate the cull the same way: the contact wave is shared code)
    out = gpu_rollout(dm, inp, dev)
    rc = RefC(tpl, np.float32)
    st = rc.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    gr = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
    assert np.abs(st["grf"]).max() > 100.0, "the kicked robots must hit the ground hard"
    assert relmax(out["wp_pos"], st["wp_pos"]) < 1e-4 and relmax(out["wp_vel"], st["wp_vel"]) < 5e-3
    assert relmax(out["grf"], st["grf"]) < 5e-3 and relmax(out["jaf"], st["jaf"]) < 1e-2
    for k in ("q_init", "qd_init", "res_f", "refs", "body_inv_mass"):
        # (another fp32 trajectory than the fp32 oracle's through 26 violent steps: 2e-2 measured for the lane-per-body kernels, 2.7e-2 for
        # the quad-lane ones; the tight per-env gradient check is tests/test_gpu_tight.py's own-trajectory test, run for both families)
        e_k = relmax(out["grads"][k].reshape(gr[k].shape), gr[k])
        assert family == 2 or e_k < 2e-2, (k, e_k)
    # ... and, for both families, against the float64 adjoint of the kernel's OWN trajectory (helpers.own_trajectory_check): no chaos in
    # that comparison, every env measured (24 envs x 26 violent steps; measured worst env 1.3e-4 / 1.7e-4)
    from helpers import own_trajectory_check
    chk = own_trajectory_check(dm, tpl, inp, dev)
    print("family %d: own-trajectory worst env %.1e (median %.1e), hit-log entries missing %d" % (family, chk["worst"].max(), np.median(chk["worst"]), chk["hitlog_missing"]))
    assert chk["hitlog_missing"] == 0 and chk["worst"].max() < 1e-3, (float(chk["worst"].max()), chk["hitlog_missing"])
    # batch-composition invariance: the same envs alone (different wave-mates, different redo pattern) give the same bits
    sub = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in inp.items()}
    pick = np.arange(1, bs, 3)
    for k in ("q_init", "qd_init", "target_ke", "target_kd", "body_mass", "body_inv_mass", "body_inertia", "body_inv_inertia"):
        sub[k] = inp[k].reshape(bs, -1)[pick].reshape(-1)
    for k in ("torques", "refs", "res_f"):
        sub[k] = inp[k].reshape(T, bs, -1)[:, pick].reshape(T, -1)
    F = len(inp["frame2step"])
    for k in ("adj_pos", "adj_vel"):
        sub[k] = inp[k].reshape(F, bs, -1)[:, pick].reshape(F, -1)
    dm2 = hip_backend.DeviceModel(tpl)
    dm2.set_kernel_family(family)
    out2 = gpu_rollout(dm2, sub, dev)
    assert np.array_equal(out2["wp_pos"].reshape(F, len(pick), -1), out["wp_pos"].reshape(F, bs, -1)[:, pick])
    assert np.array_equal(out2["grads"]["q_init"].reshape(len(pick), -1), out["grads"]["q_init"].reshape(bs, -1)[pick])


def test_unsplit_forward_speculates_and_redoes_its_cull(dev, oracle_libs):
    """Compound robots at more than four env groups per compute unit run the UNSPLIT forward kernel (one wave does bodies, joints
    and contacts), which since round 3 keeps speculated contact candidates for PD_SPEC_K steps like the contact wave does.  A quad
    batch just above that threshold with downward kicks on part of the envs (margins outrun inside an epoch => the cull is redone
    at once) must match the fp32 C oracle, and the same envs in a small batch -- which takes the wave-specialised kernel and the
    exact hand-over path -- must give the same forward bits: both kernels sum an env's contact wrenches in list order."""
    from diffphys_amd import hip_backend, robots, synth
    from oracle.ref_c import RefC

    tpl = robots.load_template("quad")
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    bs, T = 8 * cus + 64, 22   # env groups (2 envs each) > 4 per compute unit
    inp = synth.make_inputs(tpl, "quad", bs=bs, nsteps=T, seed=23, steps_per_frame=7, penetration=0.004)
    nb = int(tpl["nb"])
    rf = inp["res_f"].reshape(T, bs, nb, 6)
    mass = inp["body_mass"].reshape(bs, nb)
    kick = np.zeros((T, bs), np.float32)
    kick[2:12, ::3] = 3000.0     # every third env: its wave-mate stays calm
    kick[9:18, 1::5] = 1500.0
    rf[..., 4] -= kick[:, :, None] * mass[None]
    dm = hip_backend.DeviceModel(tpl)
    out = gpu_rollout(dm, inp, dev)
    info = dm.last_launch_info(0)
    assert info["threads_per_wg"] * 2 == 64 * info["envs_per_wg"], info   # one wave per env group: the unsplit kernel ran
    rc = RefC(tpl, np.float32)
    st = rc.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    gr = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
    assert np.abs(st["grf"]).max() > 100.0, "the kicked robots must hit the ground hard"
    assert relmax(out["wp_pos"], st["wp_pos"]) < 1e-4 and relmax(out["wp_vel"], st["wp_vel"]) < 5e-3
    assert relmax(out["grf"], st["grf"]) < 5e-3 and relmax(out["jaf"], st["jaf"]) < 1e-2
    for k in ("q_init", "qd_init", "res_f", "refs", "body_inv_mass"):
        assert relmax(out["grads"][k].reshape(gr[k].shape), gr[k]) < 2e-2, k
    sub = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in inp.items()}
    pick = np.arange(0, 96, 1)   # kicked and calm envs alike
    for k in ("q_init", "qd_init", "target_ke", "target_kd", "body_mass", "body_inv_mass", "body_inertia", "body_inv_inertia"):
        sub[k] = inp[k].reshape(bs, -1)[pick].reshape(-1)
    for k in ("torques", "refs", "res_f"):
        sub[k] = inp[k].reshape(T, bs, -1)[:, pick].reshape(T, -1)
    F = len(inp["frame2step"])
    for k in ("adj_pos", "adj_vel"):
        sub[k] = inp[k].reshape(F, bs, -1)[:, pick].reshape(F, -1)
    dm2 = hip_backend.DeviceModel(tpl)
    out2 = gpu_rollout(dm2, sub, dev)
    info2 = dm2.last_launch_info(0)
    assert info2["threads_per_wg"] == 64 * info2["envs_per_wg"], info2    # two waves per env group: the wave-specialised kernel
    for k in ("wp_pos", "wp_vel", "grf", "jaf"):
        assert np.array_equal(out2[k].reshape(F, len(pick), -1), out[k].reshape(F, bs, -1)[:, pick]), k
    assert np.array_equal(out2["grads"]["q_init"].reshape(len(pick), -1), out["grads"]["q_init"].reshape(bs, -1)[pick])


@pytest.mark.parametrize("dim", [7, 6])
def test_fused_se3_loss_matches_the_torch_composition(dim, dev):
    """pd_se3_loss (one launch: loss + both gradients) against the torch restatement of the reference's
    se3_loss (diffphys/dp_utils.py:113-138) evaluated in float64 -- random poses, tiny / zero axis-angles, identical
    rotations (clamped acos: zero rotation gradient), NaN entries (loss and gradients 0).  Tolerances: loss 2e-5
    relative to max, gradients 5e-4 relative to max (fp32 acos near the clamp)."""
    from diffphys_amd import dp_utils
    from oracle import pose_torch

    g = torch.Generator().manual_seed(5)
    n = 4099
    pred = torch.randn(n, dim, generator=g)
    gt = pred + 0.3 * torch.randn(n, dim, generator=g)
    if dim == 7:
        pred[:, 3:] *= 1.7                      # the conversion normalises: non-unit quaternions are legal inputs
        gt[10:20, 3:] = pred[10:20, 3:] * 0.5   # same rotation, different norm -> cos = 1 -> clamped
    else:
        pred[30:40, 3:] = 1e-8 * torch.randn(10, 3, generator=g)   # series branch of axis_angle_to_quaternion
        gt[30:35, 3:] = 0.0                                       # |a| = 0 exactly (d|a|/da taken as 0)
        gt[10:20, 3:] = pred[10:20, 3:]
    pred[50, 1] = float("nan"); gt[51, 4] = float("nan")
    p32, g32 = pred.to(dev).requires_grad_(True), gt.to(dev).requires_grad_(True)
    w = torch.randn(n, generator=g).to(dev)
    loss = dp_utils.se3_loss(p32, g32, 0.1)
    (loss * w).sum().backward()
    p64, g64 = pred.double().to(dev).requires_grad_(True), gt.double().to(dev).requires_grad_(True)
    ref = pose_torch.se3_loss(p64, g64, 0.1)
    ok = torch.ones(n, dtype=torch.bool, device=dev); ok[50] = False; ok[51] = False
    (ref * w.double()).sum().backward()
    assert loss[50].item() == 0.0 and loss[51].item() == 0.0
    # NaN inputs: the loss is 0 and the gradient is 0 x (the local derivative) -- NaN where that is not finite, as autograd makes of the
    # reference's `loss[nanid] = 0` (dp_utils.py:137): the same NaN pattern as autograd through the float64 restatement
    assert torch.equal(p32.grad.isnan(), p64.grad.isnan()) and torch.equal(g32.grad.isnan(), g64.grad.isnan())
    assert p32.grad[50].isnan().tolist() == [False, True] + [False] * (dim - 2) and g32.grad[51, 3:].isnan().all()
    assert torch.isfinite(loss).all() and torch.isfinite(p32.grad[ok]).all() and torch.isfinite(g32.grad[ok]).all()
    rel = lambda a, b: float((a.detach().double() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-30))
    assert rel(loss[ok], ref[ok]) < 2e-5
    assert rel(p32.grad[ok], p64.grad[ok]) < 5e-4 and rel(g32.grad[ok], g64.grad[ok]) < 5e-4
    # clamped entries: translation gradient only
    assert p32.grad[10:20, 3:].abs().max().item() == 0.0
    # shapes with leading dims and no-grad calls go through the same kernel
    l2 = dp_utils.se3_loss(pred.to(dev).reshape(1, n, dim), gt.to(dev).reshape(1, n, dim))
    assert l2.shape == (1, n) and torch.equal(l2.reshape(-1), loss.detach())


def test_more_contact_candidates_than_fit_at_the_default_width(dev, oracle_libs):
    """A robot whose contact tables + 16 envs per workgroup exceed the LDS gets a wider segment (fewer envs per workgroup)
    instead of an error: Laikago with its candidates duplicated (5 757 points) must build, pick 32 lanes, and match the oracle."""
    from diffphys_amd import hip_backend, robots, synth
    from oracle.ref_c import RefC

    tpl = dict(robots.load_template("laikago"))
    nc = len(tpl["contact_body"])
    rng = np.random.RandomState(3)
    extra = rng.choice(nc, nc // 2, replace=False)
    for k in ("contact_body", "contact_dist", "contact_material"):
        tpl[k] = np.concatenate([tpl[k], tpl[k][extra]])
    jitter = (rng.randn(len(extra), 3) * 2e-4).astype(np.float32)
    tpl["contact_point"] = np.concatenate([tpl["contact_point"], tpl["contact_point"][extra] + jitter]).astype(np.float32)
    dm = hip_backend.DeviceModel(tpl)
    assert dm.segment_width() == 32
    bs, T = 9, 12
    inp = synth.make_inputs(robots.load_template("laikago"), "laikago", bs=bs, nsteps=T, seed=4, steps_per_frame=5, penetration=0.003)
    out = gpu_rollout(dm, inp, dev)
    rc = RefC(tpl, np.float32)
    st = rc.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    gr = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
    assert np.abs(st["grf"]).max() > 1.0
    assert relmax(out["wp_pos"], st["wp_pos"]) < 5e-5 and relmax(out["grf"], st["grf"]) < 5e-3
    for k in ("q_init", "qd_init", "refs", "body_inv_mass"):
        assert relmax(out["grads"][k].reshape(gr[k].shape), gr[k]) < 2e-2, k


@pytest.mark.parametrize("name,bs", [("laikago", 37), ("human", 9), ("laikago", 6700)])  # 6700 x 5 entries: the reduce kernel's table no longer fits LDS
def test_traj_loss_inside_the_rollout_equals_forwardwarp_se3_loss_reduce_loss(name, bs, dev):
    """Row f4 at the C ABI: pd_rollout_forward_traj_loss / pd_rollout_backward_traj_loss against ForwardWarp.apply -> pd_se3_loss ->
    torch reduce_loss(clip=True) (dp_model.py:733-779, dp_utils.py:93-138) on the same inputs: the reduced loss, the [bs, F] table, the
    clip threshold / counts, the frame poses (bit-identical: same rollout arithmetic) and the gradients of all rollout inputs AND of
    the target poses.  Targets = the simulated poses + noise; one env gets a far target from frame 2 on (clipped there), one env has
    out-of-sequence frames, one target pose is NaN (se3_loss ignores it), frames include state T."""
    from diffphys_amd import dp_model, dp_utils, hip_backend, robots, synth

    tpl = robots.load_template(name)
    T = 70
    inp = synth.make_inputs(tpl, name, bs=bs, nsteps=T, seed=5, steps_per_frame=23, penetration=0.002)
    f2s = [0, 23, 46, 69, T]
    F, nb = len(f2s), int(tpl["nb"])

    class Host:
        pass

    h = Host()
    h.env = robots.env_from_template(name, bs, device=dev)
    h.num_envs, h.steps_idx, h.frame2step, h.dt = bs, range(T), f2s, inp["dt"]
    # (a small Laikago batch takes the quad-lane kernels in BOTH paths: their forward kernel has the loss-evaluating instantiation too)
    t = {k: torch.from_numpy(inp[k]).to(dev).requires_grad_(True) for k in synth.INPUT_NAMES}
    args = [t[k] for k in synth.INPUT_NAMES]
    with torch.no_grad():
        pos0, _ = dp_model.ForwardWarp.apply(*args, h)
    g = torch.Generator().manual_seed(11)
    tgt = pos0.reshape(F, bs, nb, 7).permute(1, 0, 2, 3).clone()
    tgt = tgt + (torch.randn(tgt.shape, generator=g) * 0.02).to(dev)
    tgt[3, 2:, :, :3] += 0.8                # env 3: far from its targets from frame 2 on => clipped at frame 2
    tgt[1, 4, 5, 2] = float("nan")           # one NaN target pose: its se3_loss is 0; its gradient is 0 x NaN = NaN like autograd's, which
                                             # seeds NaN into env 1's adjoint (scrubbed to 0 at the stores, as the reference's remove_nan does)
    outseq = torch.zeros(bs, F, dtype=torch.bool, device=dev)
    outseq[2, 3:] = True                     # env 2: frames 3.. belong to another clip
    tgt = tgt.contiguous().requires_grad_(True)
    wt = 0.37

    # the reference's sequence through torch (the fused se3_loss launch inside; its torch composition is tested elsewhere)
    pos, vel = dp_model.ForwardWarp.apply(*args, h)
    pos.retain_grad()
    sim = pos.reshape(F, bs, nb, 7).permute(1, 0, 2, 3)
    lt = dp_utils.se3_loss(sim, tgt).mean(-1)
    lt = torch.where(outseq, torch.zeros_like(lt), lt)
    table_ref = lt.detach().clone()
    loss_ref = dp_utils.reduce_loss(lt, clip=True)
    (loss_ref * wt).backward()
    ref = {k: t[k].grad.detach().clone() for k in synth.INPUT_NAMES}
    ref_tgt = tgt.grad.detach().clone()
    ref_seed = pos.grad.detach().clone()   # what autograd hands ForwardWarp.backward as adj_body_qs
    for k in synth.INPUT_NAMES:
        t[k].grad = None
    tgt.grad = None

    loss, pos_f, vel_f = dp_model.ForwardWarpTrajLoss.apply(*args, tgt, outseq, h)
    assert not pos_f.requires_grad and torch.equal(pos_f, pos.detach()) and torch.equal(vel_f, vel.detach())
    (loss * wt).backward()
    info = h.traj_loss_info.cpu().numpy()
    print("%s: loss_traj %.6e (torch %.6e), threshold %.3e, positives left %d, clipped envs %d" % (name, float(loss.detach()), float(loss_ref.detach()), info[1], info[2], info[3]))
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 1e-6 * abs(float(loss_ref.detach()))
    assert info[3] >= 1 and info[2] < bs * F
    # the table the kernel wrote = se3_loss(...).mean(-1) with the outseq entries zeroed (before the clip)
    dm = hip_backend.device_model(h.env)
    c = lambda x: x.detach().to(torch.float32).contiguous()
    out = dm.rollout_forward_traj_loss(bs, T, inp["dt"], *[c(t[k]) for k in FWD], frame2step=f2s, target_pos=c(tgt), outseq=outseq)
    table = out[5]["table"]
    assert float((table - table_ref).abs().max()) <= 2e-6 * float(table_ref.abs().max())
    # the seeds the adjoint kernel builds for itself = what autograd computed for adj_body_qs, to 1e-6 ...
    tl = out[5]
    seed = tl["seed_pos"].view(F, bs, nb, 7) * (tl["scale"].t() * (wt / nb))[:, :, None, None]
    nan_seed = ref_seed.isnan()
    assert int(nan_seed.sum()) == 1 and torch.equal(seed.view_as(ref_seed).isnan(), nan_seed)    # the z of body 5, env 1, frame 4
    ref_seed = ref_seed.nan_to_num(nan=0.0)
    assert float((seed.view_as(ref_seed).nan_to_num(nan=0.0) - ref_seed).abs().max()) <= 1e-6 * float(ref_seed.abs().max())
    # ... and the gradients they give, to 1e-4 of each tensor's max: the two seed tensors differ in their last bits (another order of
    # the scalings), and a 70-step adjoint amplifies that (tests/test_gpu_tight.py: ONE ulp on a stored Laikago state moves these
    # gradients by ~1e-2 over 100 steps); measured 8e-6 (Laikago), 2e-7 (human)
    for k in synth.INPUT_NAMES:
        a, b = t[k].grad, ref[k]
        sc = float(b.abs().max())
        print("   %-18s fused vs torch sequence: %.1e of the tensor's max" % (k, float((a - b).abs().max()) / (sc + 1e-30)))
        assert float((a - b).abs().max()) <= 1e-4 * sc + 1e-30, (k, float((a - b).abs().max()), sc)
    assert torch.equal(tgt.grad.isnan(), ref_tgt.isnan()) and int(ref_tgt.isnan().sum()) == 1 and bool(tgt.grad[1, 4, 5, 2].isnan())
    assert float((tgt.grad.nan_to_num(nan=0.0) - ref_tgt.nan_to_num(nan=0.0)).abs().max()) <= 1e-6 * float(ref_tgt.nan_to_num(nan=0.0).abs().max())
    assert float(tgt.grad[3, 2:].abs().max()) == 0 and float(tgt.grad[2, 3:].abs().max()) == 0
    assert float(t["q_init"].grad.view(bs, -1)[1].abs().max()) == 0, "the NaN seed spreads over env 1 and is scrubbed: that env's gradient is dropped"
    # other terms may still reach the poses: adj_pos / adj_vel rows are ADDED to the loss seeds
    ap = torch.from_numpy(inp["adj_pos"][:1].repeat(F, 0)).to(dev).contiguous()
    av = torch.from_numpy(inp["adj_vel"][:1].repeat(F, 0)).to(dev).contiguous()
    ins = [c(t[k]) for k in BWD]
    one = torch.ones(1, device=dev)
    g_both = dm.rollout_backward_traj_loss(bs, T, inp["dt"], *ins, f2s, out[4], out[5], one * wt, adj_pos=ap, adj_vel=av)
    g_seed = dm.rollout_backward_traj_loss(bs, T, inp["dt"], *ins, f2s, out[4], out[5], one * wt)
    g_adj = dm.rollout_backward(bs, T, inp["dt"], *ins, f2s, out[4], ap, av)
    # the seeds pd_rollout_backward_traj_loss built on the device for that last sweep = autograd's adj_body_qs, and no twist seeds
    work = out[5]["work"]
    assert torch.equal(work[: F * bs * nb * 7].view_as(ref_seed).isnan(), nan_seed)
    assert float((work[: F * bs * nb * 7].view_as(ref_seed).nan_to_num(nan=0.0) - ref_seed).abs().max()) <= 1e-6 * float(ref_seed.abs().max())
    assert float(work[F * bs * nb * 7:].abs().max()) == 0
    keep = torch.ones(bs, dtype=torch.bool, device=dev); keep[1] = False   # (env 1: NaN seed => its gradient is dropped whenever the loss seeds are in)
    envs = lambda x, k: (x.reshape(x.shape[0], bs, -1) if GRAD_LEAD[k] else x.reshape(1, bs, -1))[:, keep]
    for k in g_both:
        s_ = envs(g_seed[k] + g_adj[k], k)
        assert float((envs(g_both[k], k) - s_).abs().max()) <= 2e-4 * float(s_.abs().max()) + 1e-30, k   # linear in the seeds (fp32 sums in another order; measured 3e-5)
        assert float(envs(g_both[k], k).abs().max()) > 0, k


@pytest.mark.parametrize("name,bs", [("laikago", 37), ("human", 9), ("laikago", 6700), ("laikago", 2100)])
def test_fk_of_the_control_reference_rides_on_the_traj_loss_launches(name, bs, dev):
    """Row f4, second half: ForwardWarpTrajLossFK = ForwardWarpTrajLoss + ForwardKinematics.apply(queried_q, queried_qd, env)
    (dp_model.py:758 of the reference) with the FK chains as extra workgroups of the reduce_loss launch and their adjoint as extra
    workgroups of the seeds launch (C ABI pd_rollout_*_traj_loss_fk).  Same device code as the separate launches: poses, twists, the
    loss and EVERY gradient must be bit-identical, including ForwardKinematics.backward's NaN -> 0 / > 1 -> 1 post-processing; the
    [bs, F, ...] layout comes straight out of the kernel.  6700 envs: the reduce table does not fit LDS; 2100: lane-per-body kernels."""
    from diffphys_amd import dp_model, hip_backend, robots, synth

    tpl = robots.load_template(name)
    T = 40
    inp = synth.make_inputs(tpl, name, bs=bs, nsteps=T, seed=6, steps_per_frame=13, penetration=0.002)
    f2s = [0, 13, 26, 39]
    F, nb, nq, nqd = len(f2s), int(tpl["nb"]), int(tpl["nq"]), int(tpl["nqd"])

    class Host:
        pass

    h = Host()
    h.env = robots.env_from_template(name, bs, device=dev)
    h.num_envs, h.steps_idx, h.frame2step, h.dt = bs, range(T), f2s, inp["dt"]
    t = {k: torch.from_numpy(inp[k]).to(dev).requires_grad_(True) for k in synth.INPUT_NAMES}
    args = [t[k] for k in synth.INPUT_NAMES]
    with torch.no_grad():
        pos0, _ = dp_model.ForwardWarp.apply(*args, h)
    g = torch.Generator().manual_seed(12)
    tgt = (pos0.reshape(F, bs, nb, 7).permute(1, 0, 2, 3) + (torch.randn(bs, F, nb, 7, generator=g) * 0.02).to(dev)).contiguous()
    outseq = torch.zeros(bs, F, dtype=torch.bool, device=dev)
    # the control reference: the initial joint state of every env, perturbed per frame
    q0 = torch.from_numpy(inp["q_init"]).view(bs, nq)
    qq = (q0[None] + 0.1 * torch.randn(F, bs, nq, generator=g)).to(dev).requires_grad_(True)
    qqd = (0.5 * torch.randn(F, bs, nqd, generator=g)).to(dev).requires_grad_(True)
    w_q = (torch.randn(bs, F, nb, 7, generator=g) * 3.0).to(dev)   # some FK gradients land above 1: the clamp of :1109-1110 acts
    w_q[0, 1, 2, 0] = float("nan")                                   # ... and one is NaN -> 0
    w_qd = torch.randn(bs, F, nb, 6, generator=g).to(dev)
    leaves = args + [qq, qqd]

    def grads():
        out = [x.grad.detach().clone() for x in leaves]
        for x in leaves:
            x.grad = None
        return out

    # the separate launches
    loss, pos, vel = dp_model.ForwardWarpTrajLoss.apply(*args, tgt, outseq, h)
    qp, qv, pid = dp_model.ForwardKinematics.apply(qq, qqd, h.env)
    (loss * 0.37 + (qp * w_q).sum() + (qv * w_qd).sum()).backward()
    ref = grads()
    # riding along
    loss2, pos2, vel2, qp2, qv2, pid2 = dp_model.ForwardWarpTrajLossFK.apply(*args, tgt, outseq, qq, qqd, h)
    assert qp2.shape == (bs, F, nb, 7) and qv2.shape == (bs, F, nb, 6) and qp2.is_contiguous()
    assert torch.equal(qp2, qp) and torch.equal(qv2, qv) and torch.equal(pos2, pos) and torch.equal(vel2, vel) and torch.equal(loss2, loss)
    assert np.array_equal(np.stack(pid2, 0), np.stack(pid, 0))
    (loss2 * 0.37 + (qp2 * w_q).sum() + (qv2 * w_qd).sum()).backward()
    got = grads()
    names = list(synth.INPUT_NAMES) + ["queried_q", "queried_qd"]
    for k, a, b in zip(names, got, ref):
        assert torch.equal(a, b), (k, float((a - b).abs().max()))
    assert float(ref[-2].max()) == 1.0 and bool(torch.isfinite(ref[-2]).all())   # the clamp acted, the NaN weight left no NaN
    # only one of the two FK outputs used downstream: the other adjoint is zero
    loss3, _, _, qp3, qv3, _ = dp_model.ForwardWarpTrajLossFK.apply(*args, tgt, outseq, qq, qqd, h)
    (qv3 * w_qd).sum().backward()
    g3 = grads()
    qp4, qv4, _ = dp_model.ForwardKinematics.apply(qq, qqd, h.env)
    (qv4 * w_qd).sum().backward()
    assert torch.equal(g3[-1], qqd.grad) and torch.equal(g3[-2], qq.grad)
    # the C ABI takes any F x bs of chains, not only the rollout's: 3 frames of 5 envs beside this rollout, and beside an EMPTY rollout
    dm = hip_backend.device_model(h.env)
    c = lambda x: x.detach().to(torch.float32).contiguous()
    jq, jqd = c(qq[:3, :5]), c(qqd[:3, :5])
    want_q, want_qd = dm.fk_forward(jq.view(15, nq), jqd.view(15, nqd))
    out = dm.rollout_forward_traj_loss(bs, T, inp["dt"], *[c(t[k]) for k in FWD], frame2step=f2s, target_pos=c(tgt), outseq=outseq, fk=(jq, jqd))
    assert torch.equal(out[5]["fk_body_q"], want_q.view(3, 5, nb, 7).permute(1, 0, 2, 3)) and torch.equal(out[5]["fk_body_qd"], want_qd.view(3, 5, nb, 6).permute(1, 0, 2, 3))
    e = lambda *s: torch.zeros(*s, device=dev)
    out0 = dm.rollout_forward_traj_loss(0, T, inp["dt"], e(0), e(0), e(T, 0), e(T, 0, 6), e(T, 0), e(0), e(0), e(0), e(0, 3, 3), e(0, 3, 3),
                                        frame2step=f2s, target_pos=e(0, F, nb, 7), fk=(jq, jqd))
    assert torch.equal(out0[5]["fk_body_q"], out[5]["fk_body_q"]) and float(out0[5]["reduced"][0]) == 0.0


@pytest.mark.parametrize("family", [1, 2], ids=["lane-per-body", "quad-lane"])
def test_revolute_tree_with_six_children_on_the_root(family, dev, oracle_libs):
    """Both kernel families gather the first four children of a body with unrolled reads and the rest in a loop that no shipped robot
    enters (Laikago's root has exactly four).  A re-parented Laikago -- the root carries six revolute children (bodies 1, 2, 4, 5, 7,
    10), three chains of depth 2-3 hang below -- is still a revolute-only PLAIN model (quad-lane eligible) and must match the C oracle:
    poses, forces and all gradients at a short horizon, forward and adjoint."""
    from diffphys_amd import hip_backend, robots, synth
    from oracle.ref_c import RefC

    tpl = dict(robots.load_template("laikago"))
    parent = np.array([-1, 0, 0, 2, 0, 0, 5, 0, 7, 8, 0, 10, 11], np.int32)
    tpl["joint_parent"] = parent
    bs, T = 21, 24
    inp = synth.make_inputs(tpl, "laikago", bs=bs, nsteps=T, seed=9, steps_per_frame=11, penetration=0.004)
    rng = np.random.RandomState(9)
    inp["torques"] = (rng.randn(*inp["torques"].shape) * 0.2).astype(np.float32)
    inp["res_f"] = (rng.randn(*inp["res_f"].shape) * 0.2).astype(np.float32)
    dm = hip_backend.DeviceModel(tpl)
    dm.set_kernel_family(family)
    out = gpu_rollout(dm, inp, dev)
    info = dm.last_launch_info(0)
    assert info["envs_per_wg"] * 192 // info["threads_per_wg"] == (1 if family == 2 else 4)   # envs per wave triple (body, contact, cull): which family ran
    rc = RefC(tpl, np.float32)
    st = rc.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    gr = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
    assert np.abs(st["grf"]).max() > 1.0 and np.abs(st["jaf"]).max() > 1.0
    # (the re-parented joints start far from their anchors: the attach springs throw the bodies around, twists run into the +-10
    # clamps -- bars a few times looser than the robots'; a child missing from a gather is an O(1) error)
    errs = {k: relmax(out[k], st[k]) for k in ("wp_pos", "wp_vel", "grf", "jaf")}
    errs.update({"g_" + k: relmax(out["grads"][k].reshape(gr[k].shape), gr[k]) for k in GRADS})
    print("six children, family %d: " % family + "  ".join("%s %.1e" % kv for kv in errs.items()))
    assert errs["wp_pos"] < 2e-5 and errs["wp_vel"] < 5e-3 and errs["grf"] < 2e-2 and errs["jaf"] < 2e-2, errs
    for k in GRADS:
        assert np.isfinite(out["grads"][k]).all(), k
        assert errs["g_" + k] < 5e-2, (k, errs)
    # the tight bar: the float64 adjoint of the kernel's OWN trajectory (no rollout divergence in it), every env
    from helpers import own_trajectory_check

    own = own_trajectory_check(dm, tpl, inp, dev, abs_floor=1e-8)
    print("   own trajectory: worst env %.1e, median %.1e (one-ulp conditioning median %.1e, plain fp32 %.1e)" % (
        own["worst"].max(), np.median(own["worst"]), np.median(own["cond"]), np.median(own["fp32_atan2"])))
    assert own["worst"].max() < 1e-3, own["worst"]   # measured 1.7e-5 (both families)


@pytest.mark.parametrize("family", [1, 2], ids=["lane-per-body", "quad-lane"])
def test_revolute_joint_limits_engage(family, dev, oracle_libs):
    """The shipped robots run with limit_ke = limit_kd = 0 (dp_model.py:144-145 of the reference), so eval_joint_force's limit branch
    (integrator_euler.py:261-286) is dead for them.  Laikago with +-0.15 rad limits and limit_ke = 60, limit_kd = 1.5 on every revolute
    joint, started beyond the limits on most joints: forward and adjoint of both kernel families against the C oracle (short horizon)
    and the float64 adjoint of the kernel's own trajectory."""
    from diffphys_amd import hip_backend, robots, synth
    from helpers import own_trajectory_check
    from oracle.ref_c import RefC

    tpl = dict(robots.load_template("laikago"))
    nqd = int(tpl["nqd"])
    lo, up, lke, lkd = (np.array(tpl[k], np.float32).copy() for k in ("joint_limit_lower", "joint_limit_upper", "joint_limit_ke", "joint_limit_kd"))
    lo[6:], up[6:], lke[6:], lkd[6:] = -0.15, 0.15, 60.0, 1.5
    tpl.update(joint_limit_lower=lo, joint_limit_upper=up, joint_limit_ke=lke, joint_limit_kd=lkd)
    bs, T = 19, 30
    inp = synth.make_inputs(tpl, "laikago", bs=bs, nsteps=T, seed=10, steps_per_frame=14, penetration=0.003)
    dm = hip_backend.DeviceModel(tpl)
    dm.set_kernel_family(family)
    out = gpu_rollout(dm, inp, dev)
    q0 = inp["q_init"].reshape(bs, -1)[:, 7:]
    assert (np.abs(q0) > 0.15).mean() > 0.3                      # the limits are engaged from the first step on
    rc = RefC(tpl, np.float32)
    st = rc.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    gr = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
    rc0 = RefC(dict(tpl, joint_limit_ke=np.zeros(nqd, np.float32), joint_limit_kd=np.zeros(nqd, np.float32)), np.float32)
    st0 = rc0.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    assert relmax(st0["jaf"], st["jaf"]) > 1e-2                  # ... and they matter
    assert relmax(out["wp_pos"], st["wp_pos"]) < 2e-5 and relmax(out["wp_vel"], st["wp_vel"]) < 1e-3
    assert relmax(out["grf"], st["grf"]) < 5e-3 and relmax(out["jaf"], st["jaf"]) < 5e-3
    # against the oracle's own rollout: per-env medians (a joint angle within rounding of its limit takes the other branch in one of
    # the two rollouts: single envs legitimately disagree); the tight bar is the own-trajectory check below
    for k in GRADS:
        assert np.isfinite(out["grads"][k]).all(), k
        g, r = out["grads"][k].astype(np.float64), gr[k].astype(np.float64)
        g, r = (g.reshape(T, bs, -1).transpose(1, 0, 2).reshape(bs, -1), r.reshape(T, bs, -1).transpose(1, 0, 2).reshape(bs, -1)) if k in ("torques", "res_f", "refs") else (g.reshape(bs, -1), r.reshape(bs, -1))
        per_env = np.abs(g - r).max(1) / (np.abs(r).max(1) + 1e-12)
        assert np.median(per_env) < 2e-2, (k, np.sort(per_env))
    own = own_trajectory_check(dm, tpl, inp, dev, abs_floor=1e-8)
    print("limits, family %d: own trajectory worst env %.1e, median %.1e" % (family, own["worst"].max(), np.median(own["worst"])))
    assert own["worst"].max() < 1e-3, own["worst"]


@pytest.mark.parametrize("name,family", [("laikago", 1), ("laikago", 2), ("human", 0)])
def test_zero_step_rollout_is_fk_and_its_adjoint(name, family, dev, oracle_libs):
    """nsteps = 0 with a frame at state 0: the forward is eval_fk of (q_init, qd_init) (dp_model.py:1204), the backward its adjoint
    (:1294-1306) -- the path where the adjoint kernels have no staged records and rebuild state 0 themselves."""
    from diffphys_amd import hip_backend, robots, synth
    from oracle.ref_c import RefC

    tpl = robots.load_template(name)
    bs, nb = 5, int(tpl["nb"])
    inp = synth.make_inputs(tpl, name, bs=bs, nsteps=1, seed=13, penetration=0.002)
    for k in ("torques", "res_f", "refs"):
        inp[k] = inp[k][:0]
    inp["nsteps"] = 0
    rng = np.random.RandomState(1)
    inp["frame2step"] = [0]
    inp["qd_init"] = (rng.randn(*inp["qd_init"].shape) * 0.3).astype(np.float32)
    inp["adj_pos"] = (rng.randn(1, bs * nb, 7) * 1e-2).astype(np.float32)
    inp["adj_vel"] = (rng.randn(1, bs * nb, 6) * 1e-2).astype(np.float32)
    dm = hip_backend.DeviceModel(tpl)
    if family:
        dm.set_kernel_family(family)
    out = gpu_rollout(dm, inp, dev)
    rc = RefC(tpl, np.float32)
    st = rc.rollout_forward(inp, 0, [0], inp["dt"])
    gr = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
    assert relmax(out["wp_pos"], st["wp_pos"]) < 1e-6 and relmax(out["wp_vel"], st["wp_vel"]) < 1e-5
    assert np.abs(out["grf"]).max() == 0 and np.abs(out["jaf"]).max() == 0          # no step, no force snapshot: zero rows
    for k in ("q_init", "qd_init"):
        assert relmax(out["grads"][k].reshape(gr[k].shape), gr[k]) < 1e-4, k
    for k in ("target_ke", "target_kd", "body_inv_mass", "body_inertia", "body_inv_inertia"):
        assert np.abs(out["grads"][k]).max() == 0, k
    # ... and they are what the FK entry points give
    jq = torch.from_numpy(inp["q_init"]).to(dev).view(bs, -1)
    jqd = torch.from_numpy(inp["qd_init"]).to(dev).view(bs, -1)
    bq, bqd = dm.fk_forward(jq, jqd)
    # (to the last place or two: the FK inside the rollout kernels and k_fk contract their products differently)
    assert relmax(bq.cpu().numpy().reshape(1, bs * nb, 7), out["wp_pos"]) < 1e-6 and relmax(bqd.cpu().numpy().reshape(1, bs * nb, 6), out["wp_vel"]) < 1e-6


@pytest.mark.parametrize("name,family", [("laikago", 1), ("laikago", 2), ("human", 0), ("quad", 0)])
def test_edge_shapes_every_robot(name, family, dev, oracle_libs):
    """Shapes at the edges for every robot / kernel family: one env and one step with the only frame at the FINAL state; no frame at
    all (nothing to output, all-zero gradients); no force snapshots requested; a batch one env past a wave's worth."""
    from diffphys_amd import hip_backend, robots, synth
    from oracle.ref_c import RefC

    tpl = robots.load_template(name)
    nb = int(tpl["nb"])
    dm = hip_backend.DeviceModel(tpl)
    if family:
        dm.set_kernel_family(family)
    rc = RefC(tpl, np.float32)
    rng = np.random.RandomState(2)
    for bs, T, f2s in ((1, 1, [1]), (2, 3, []), (5, 4, [2, 4, 0])):
        inp = synth.make_inputs(tpl, name, bs=bs, nsteps=T, seed=14, penetration=0.002)
        F = len(f2s)
        inp["frame2step"] = f2s
        inp["adj_pos"] = (rng.randn(F, bs * nb, 7) * 1e-3).astype(np.float32)
        inp["adj_vel"] = (rng.randn(F, bs * nb, 6) * 1e-3).astype(np.float32)
        out = gpu_rollout(dm, inp, dev)
        assert out["wp_pos"].shape == (F, bs * nb, 7) and out["grf"].shape == (F, bs * nb, 6)
        if F == 0:
            for k, v in out["grads"].items():
                assert np.abs(v).max() == 0, (k, bs, T)
            continue
        st = rc.rollout_forward(inp, T, f2s, inp["dt"])
        gr = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
        assert relmax(out["wp_pos"], st["wp_pos"]) < 1e-5 and relmax(out["wp_vel"], st["wp_vel"]) < 5e-3, (bs, T)   # (velocities of 1e-2 m/s after 1-4 steps: fp32 round-off of the 16 kN/m attach springs)
        for k in ("q_init", "qd_init", "refs", "res_f", "body_inertia", "body_inv_mass"):
            ref = gr[k]
            if np.abs(ref).max() > 0:
                assert relmax(out["grads"][k].reshape(ref.shape), ref) < 2e-2, (k, bs, T)
            else:
                assert np.abs(out["grads"][k]).max() == 0, (k, bs, T)
    # no force snapshots: same poses, no grf / jaf buffers touched
    inp = synth.make_inputs(tpl, name, bs=3, nsteps=5, seed=15, penetration=0.002)
    t = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=np.float32)).to(dev) for k in FWD}
    a = dm.rollout_forward(3, 5, inp["dt"], *[t[k] for k in FWD], frame2step=[0, 5])
    b = dm.rollout_forward(3, 5, inp["dt"], *[t[k] for k in FWD], frame2step=[0, 5], want_forces=False)
    assert b[2] is None and b[3] is None and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert float(a[2][1].abs().max()) == 0.0   # the frame at state T has no force snapshot (dp_model.py:1225-1228): zero rows


@pytest.mark.parametrize("name,family", [("laikago", 1), ("laikago", 2), ("human", 0)])
def test_every_state_is_a_frame(name, family, dev, oracle_libs):
    """frame2step = 0 .. T (every state gathered, seeds at every step of the reverse sweep, the final state included), in a
    shuffled order -- the frame paths of the loops run on every iteration instead of 4 in 100."""
    from diffphys_amd import hip_backend, robots, synth
    from oracle.ref_c import RefC

    tpl = robots.load_template(name)
    nb, bs, T = int(tpl["nb"]), 7, 12
    rng = np.random.RandomState(3)
    f2s = [int(x) for x in rng.permutation(T + 1)]
    inp = synth.make_inputs(tpl, name, bs=bs, nsteps=T, seed=16, penetration=0.003)
    inp["frame2step"] = f2s
    inp["adj_pos"] = (rng.randn(T + 1, bs * nb, 7) * 1e-3).astype(np.float32)
    inp["adj_vel"] = (rng.randn(T + 1, bs * nb, 6) * 1e-3).astype(np.float32)
    dm = hip_backend.DeviceModel(tpl)
    if family:
        dm.set_kernel_family(family)
    out = gpu_rollout(dm, inp, dev)
    rc = RefC(tpl, np.float32)
    st = rc.rollout_forward(inp, T, f2s, inp["dt"])
    gr = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
    assert relmax(out["wp_pos"], st["wp_pos"]) < 1e-5 and relmax(out["wp_vel"], st["wp_vel"]) < 5e-3
    assert relmax(out["grf"], st["grf"]) < 5e-3 and relmax(out["jaf"], st["jaf"]) < 5e-3
    for k in GRADS:
        assert relmax(out["grads"][k].reshape(gr[k].shape), gr[k]) < 2e-2, k
    # the same through the loss-evaluating entries: table = se3_loss of every state's poses, seeds at every step
    from diffphys_amd import dp_utils

    t = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=np.float32)).to(dev) for k in FWD}
    F = T + 1
    pos = torch.from_numpy(out["wp_pos"]).to(dev)
    tgt = (pos.view(F, bs, nb, 7).permute(1, 0, 2, 3) + 0.01 * torch.randn(bs, F, nb, 7, device=dev)).contiguous()
    o = dm.rollout_forward_traj_loss(bs, T, inp["dt"], *[t[k] for k in FWD], frame2step=f2s, target_pos=tgt)
    assert torch.equal(o[0], pos)
    want = dp_utils.se3_loss(pos.view(F, bs, nb, 7).permute(1, 0, 2, 3).contiguous(), tgt).mean(-1)
    assert float((o[5]["table"] - want).abs().max()) <= 2e-6 * float(want.abs().max())
    one = torch.ones(1, device=dev)
    g = dm.rollout_backward_traj_loss(bs, T, inp["dt"], *[t[k] for k in BWD], f2s, o[4], o[5], one)
    seeds = o[5]["seed_pos"].view(F, bs, nb, 7) * (o[5]["scale"].t() / nb)[:, :, None, None]
    g2 = dm.rollout_backward(bs, T, inp["dt"], *[t[k] for k in BWD], f2s, o[4], seeds.reshape(F, bs * nb, 7).contiguous(), torch.zeros(F, bs * nb, 6, device=dev))
    for k in g2:
        assert float((g[k] - g2[k]).abs().max()) <= 1e-5 * float(g2[k].abs().max()) + 1e-30, k


def test_mixed_families_between_the_two_thresholds(dev, oracle_libs):
    """Between 2 x CUs and 4 x CUs envs (513 .. 1024 on MI355X) the automatic choice runs the quad-lane FORWARD kernel and the
    lane-per-body ADJOINT on the trajectory and hit log it saved.  640 envs: the forward is bit-identical to family 2's, the gradients
    pass the own-trajectory check and equal family 1's adjoint of that same workspace."""
    from diffphys_amd import hip_backend, robots, synth
    from helpers import own_trajectory_check

    tpl = robots.load_template("laikago")
    bs, T = 640, 40
    inp = synth.make_inputs(tpl, "laikago", bs=bs, nsteps=T, seed=17, steps_per_frame=13, penetration=0.003)
    dm = hip_backend.DeviceModel(tpl)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    if not (2 * cus < bs <= 4 * cus):
        pytest.skip("batch is not between the two thresholds on this device")
    t = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=np.float32)).to(dev) for k in INPUT_NAMES + ("adj_pos", "adj_vel")}
    f2s = list(inp["frame2step"])
    fwd = lambda: dm.rollout_forward(bs, T, inp["dt"], *[t[k] for k in FWD], frame2step=f2s)
    bwd = lambda ws: dm.rollout_backward(bs, T, inp["dt"], *[t[k] for k in BWD], f2s, ws, t["adj_pos"], t["adj_vel"])
    auto = fwd()
    info_f = dm.last_launch_info(0)
    g_auto = bwd(auto[4])
    info_b = dm.last_launch_info(1)
    assert info_f["envs_per_wg"] * 192 // info_f["threads_per_wg"] == 1 and info_b["envs_per_wg"] * 128 // info_b["threads_per_wg"] == 4   # (forward: body, contact, cull wave)
    dm.set_kernel_family(2)
    q = fwd()
    assert all(torch.equal(a, b) for a, b in zip(auto[:4], q[:4]))
    # (the saved trajectory, decoded: the raw workspace also holds hit-log slots past each count, which nobody writes)
    assert all(torch.equal(a, b) for a, b in zip(dm.saved_trajectory(auto[4], bs, T), dm.saved_trajectory(q[4], bs, T)))
    dm.set_kernel_family(1)
    g1 = bwd(auto[4])                      # family 1's adjoint on the quad-lane forward's workspace
    assert all(torch.equal(g_auto[k], g1[k]) for k in g1)
    dm.set_kernel_family(0)
    own = own_trajectory_check(dm, tpl, inp, dev, abs_floor=1e-8)
    print("mixed families, 640 envs: own trajectory worst env %.1e, median %.1e" % (own["worst"].max(), np.median(own["worst"])))
    assert (own["worst"] <= np.maximum(1e-3, own["cond"])).all()


@pytest.mark.parametrize("name,family", [("laikago", 1), ("laikago", 2), ("human", 0), ("quad", 0)])
def test_traj_loss_and_fk_ride_at_the_edges(name, family, dev):
    """The row-f4 entries at the edges: a zero-step rollout (the loss is se3_loss of FK(q_init) against the target, its gradient goes
    through the FK adjoint), and a rollout without frames (loss 0, zero gradients) -- both with FK chains riding along, whose results
    must still be those of pd_fk_forward / pd_fk_backward."""
    from diffphys_amd import dp_utils, hip_backend, robots, synth

    tpl = robots.load_template(name)
    nb, nq, nqd, bs = int(tpl["nb"]), int(tpl["nq"]), int(tpl["nqd"]), 6
    dm = hip_backend.DeviceModel(tpl)
    if family:
        dm.set_kernel_family(family)
    g = torch.Generator().manual_seed(21)
    for T, f2s in ((0, [0]), (3, [])):
        inp = synth.make_inputs(tpl, name, bs=bs, nsteps=max(T, 1), seed=18, penetration=0.002)
        for k in ("torques", "res_f", "refs"):
            inp[k] = inp[k][:T]
        t = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=np.float32)).to(dev) for k in FWD}
        F = len(f2s)
        jq = (t["q_init"].view(1, bs, nq) + 0.1 * torch.randn(2, bs, nq, generator=g).to(dev)).contiguous()
        jqd = (0.3 * torch.randn(2, bs, nqd, generator=g)).to(dev).contiguous()
        tgt = torch.randn(bs, F, nb, 7, generator=g).to(dev)
        o = dm.rollout_forward_traj_loss(bs, T, inp["dt"], *[t[k] for k in FWD], frame2step=f2s, target_pos=tgt, fk=(jq, jqd))
        wq, wqd = dm.fk_forward(jq.view(2 * bs, nq), jqd.view(2 * bs, nqd))
        assert torch.equal(o[5]["fk_body_q"], wq.view(2, bs, nb, 7).permute(1, 0, 2, 3)) and torch.equal(o[5]["fk_body_qd"], wqd.view(2, bs, nb, 6).permute(1, 0, 2, 3))
        aq = torch.randn(bs, 2, nb, 7, generator=g).to(dev)
        aqd = torch.randn(bs, 2, nb, 6, generator=g).to(dev)
        gr = dm.rollout_backward_traj_loss(bs, T, inp["dt"], *[t[k] for k in BWD], f2s, o[4], o[5], torch.ones(1, device=dev), fk=(jq, jqd, aq, aqd))
        wgq, wgqd = dm.fk_backward(jq.view(2 * bs, nq), jqd.view(2 * bs, nqd), aq.permute(1, 0, 2, 3).contiguous(), aqd.permute(1, 0, 2, 3).contiguous())
        assert torch.equal(gr["fk_joint_q"].view(2 * bs, nq), wgq) and torch.equal(gr["fk_joint_qd"].view(2 * bs, nqd), wgqd)
        red = o[5]["reduced"].cpu().numpy()
        if F == 0:
            assert red[0] == 0.0 and all(float(gr[k].abs().max()) == 0.0 for k in ("q_init", "qd_init", "refs", "target_ke", "body_inertia"))
        else:
            pose0, _ = dm.fk_forward(t["q_init"].view(bs, nq), t["qd_init"].view(bs, nqd))
            assert float((o[0].view(bs, nb, 7) - pose0).abs().max()) <= 1e-6 * float(pose0.abs().max())
            want = dp_utils.se3_loss(o[0].view(1, bs, nb, 7).permute(1, 0, 2, 3).contiguous(), tgt).mean(-1)
            assert float((o[5]["table"] - want).abs().max()) <= 2e-6 * float(want.abs().max())
            assert np.isfinite(red[0]) and red[0] > 0 and float(gr["q_init"].abs().max()) > 0 and bool(torch.isfinite(gr["q_init"]).all())
            assert float(gr["refs"].numel()) == 0 and float(gr["target_ke"].abs().max()) == 0.0


@pytest.mark.parametrize("n", [40, 64])
def test_forty_link_chain(n, dev, oracle_libs, tmp_path):
    """A serial chain of 40 / 64 bodies (root FREE + revolute joints, alternating axes, box collisions): the deepest trees a 64-lane
    segment can hold (64 = every lane a body) -- n FK levels, one env per wave, every body's parent is its neighbour lane -- through the model
    compiler and the revolute kernels, against the C oracle and the float64 adjoint of the kernel's own trajectory."""
    from diffphys_amd import hip_backend, sim
    from diffphys_amd.import_urdf import parse_urdf
    from helpers import build_template, own_trajectory_check
    from oracle.ref_c import RefC

    links = "".join('<link name="l%d"><collision><origin xyz="0.05 0 0"/><geometry><box size="0.10 0.05 0.05"/></geometry></collision></link>\n' % i for i in range(n))
    joints = "".join('<joint name="j%d" type="continuous"><parent link="l%d"/><child link="l%d"/><axis xyz="%s"/><origin xyz="0.10 0 0" rpy="0 0 0"/>'
                     '<limit effort="1" velocity="1"/></joint>\n' % (i, i - 1, i, "0 0 1" if i % 2 else "0 1 0") for i in range(1, n))
    (tmp_path / "snake.urdf").write_text('<?xml version="1.0"?>\n<robot name="snake">\n' + links + joints + "</robot>\n")
    b = sim.ModelBuilder()
    parse_urdf(str(tmp_path / "snake.urdf"), b, xform=sim.transform((0, 0.05, 0), sim.quat_identity()), floating=True, density=1000.0,
               armature=0.001, stiffness=20.0, damping=0.2, shape_ke=1e4, shape_kd=0.0, shape_kf=1e2, shape_mu=1.0, limit_ke=0.0, limit_kd=0.0)
    tpl = build_template(b, attach_ke=4000.0, attach_kd=40.0)   # (0.25 kg links: the reference robots' 16 kN/m would sit at the explicit scheme's stability edge)
    nb, nq, nqd = int(tpl["nb"]), int(tpl["nq"]), int(tpl["nqd"])
    assert nb == n and sorted(set(int(t) for t in tpl["joint_type"])) == [1, 4] and list(tpl["joint_parent"]) == [-1] + list(range(n - 1))
    bs, T = 5, 24
    rng = np.random.RandomState(6)
    q = np.tile(tpl["joint_q"].astype(np.float64), (bs, 1))
    q[:, 1] = 0.021 + rng.rand(bs) * 0.004               # the boxes' lower corners are in the ground
    q[:, 7:] = rng.uniform(-0.01, 0.01, (bs, nq - 7))   # nearly straight: 4 m of chain stay within a few mm of the ground plane
    mass = np.tile(tpl["body_mass"].astype(np.float64), bs)
    inertia = np.tile(tpl["body_inertia"].astype(np.float64), (bs, 1, 1))
    ke = np.tile(np.r_[np.zeros(6), np.full(nqd - 6, 20.0)], bs)
    inp = dict(q_init=q.reshape(-1), qd_init=rng.randn(bs * nqd) * 0.02, torques=rng.randn(T, bs * nqd) * 0.02,
               res_f=rng.randn(T, bs * nb, 6) * 0.02, refs=rng.uniform(-0.2, 0.2, (T, bs * nqd)), target_ke=ke, target_kd=ke * 0.01,
               body_mass=mass, body_inv_mass=1 / mass, body_inertia=inertia, body_inv_inertia=np.linalg.inv(inertia),
               adj_pos=rng.randn(3, bs * nb, 7) * 1e-3, adj_vel=rng.randn(3, bs * nb, 6) * 1e-3)
    inp = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in inp.items()}
    inp.update(frame2step=[0, 11, 24], nsteps=T, dt=5e-4)
    dm = hip_backend.DeviceModel(tpl)
    assert dm.segment_width() == 64
    out = gpu_rollout(dm, inp, dev)
    rc = RefC(tpl, np.float32)
    st = rc.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    gr = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
    assert np.abs(st["grf"]).max() > 0.1 and np.abs(st["jaf"]).max() > 0.1
    # yardstick: what the fp32 C oracle loses against the float64 one on the same inputs (39 coupled stiff joints)
    st64 = RefC(tpl, np.float64).rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    for k, floor in (("wp_pos", 2e-5), ("wp_vel", 5e-3), ("grf", 1e-2), ("jaf", 1e-2)):
        e, y = relmax(out[k], st64[k]), relmax(st[k], st64[k])
        assert e < max(floor, 3.0 * y), (k, e, y)
    for k in GRADS:
        assert np.isfinite(out["grads"][k]).all(), k
    own = own_trajectory_check(dm, tpl, inp, dev, abs_floor=1e-8)
    print("%d-link chain: own trajectory worst env %.1e, median %.1e; vs oracle rollout q_init %.1e" % (
        n, own["worst"].max(), np.median(own["worst"]), relmax(out["grads"]["q_init"].reshape(gr["q_init"].shape), gr["q_init"])))
    # per env: 1e-3, or three times what a plain fp32 evaluation of the same adjoint on the same trajectory loses (64 coupled joints)
    assert (own["worst"] <= np.maximum(1e-3, 3.0 * own["fp32_atan2"])).all(), (own["worst"], own["fp32_atan2"])


@pytest.mark.parametrize("n", [30, 45])
def test_thirty_compound_joints_in_a_chain(n, dev, oracle_libs, tmp_path):
    """31 bodies joined by 30 COMPOUND joints in series (the *_R / *_P / *_Y triples of the reference's URDF convention,
    import_urdf.py:177-196): the compound kernels (k_rollout_fwd unsplit / split, k_rollout_bwd3) on a 64-lane segment with a deep tree
    -- the shipped compound robots have 19 and 26 bodies at depth <= 5."""
    from diffphys_amd import hip_backend, sim
    from diffphys_amd.import_urdf import parse_urdf
    from helpers import build_template, own_trajectory_check
    from oracle.ref_c import RefC

    box = '<collision><origin xyz="0.05 0 0"/><geometry><box size="0.10 0.05 0.05"/></geometry></collision>'
    links = '<link name="base">%s</link>\n' % box
    joints = ""
    for i in range(n):
        par = "base" if i == 0 else "s%d_Y" % (i - 1)
        links += '<link name="s%d_R"/>\n<link name="s%d_P"/>\n<link name="s%d_Y">%s</link>\n' % (i, i, i, box)
        joints += ('<joint name="j%d_R" type="revolute"><parent link="%s"/><child link="s%d_R"/><axis xyz="1 0 0"/><origin xyz="0.10 0 0"/><limit lower="-1.5" upper="1.5" effort="1" velocity="1"/></joint>\n'
                   '<joint name="j%d_P" type="revolute"><parent link="s%d_R"/><child link="s%d_P"/><axis xyz="0 1 0"/></joint>\n'
                   '<joint name="j%d_Y" type="revolute"><parent link="s%d_P"/><child link="s%d_Y"/><axis xyz="0 0 1"/></joint>\n') % (i, par, i, i, i, i, i, i, i)
    (tmp_path / "worm.urdf").write_text('<?xml version="1.0"?>\n<robot name="worm">\n' + links + joints + "</robot>\n")
    b = sim.ModelBuilder()
    parse_urdf(str(tmp_path / "worm.urdf"), b, xform=sim.transform((0, 0.05, 0), sim.quat_identity()), floating=True, density=1000.0,
               armature=0.001, stiffness=20.0, damping=0.2, shape_ke=1e4, shape_kd=0.0, shape_kf=1e2, shape_mu=1.0, limit_ke=0.0, limit_kd=0.0)
    tpl = build_template(b, attach_ke=4000.0, attach_kd=40.0)
    nb, nq, nqd = int(tpl["nb"]), int(tpl["nq"]), int(tpl["nqd"])
    assert nb == n + 1 and sorted(set(int(t) for t in tpl["joint_type"])) == [4, 5] and nqd == 6 + 3 * n
    bs, T = 5, 20
    rng = np.random.RandomState(8)
    q = np.tile(tpl["joint_q"].astype(np.float64), (bs, 1))
    q[:, 1] = 0.021 + rng.rand(bs) * 0.004
    q[:, 7:] = rng.uniform(-0.01, 0.01, (bs, nq - 7))
    mass = np.tile(tpl["body_mass"].astype(np.float64), bs)
    inertia = np.tile(tpl["body_inertia"].astype(np.float64), (bs, 1, 1))
    ke = np.tile(np.r_[np.zeros(6), np.full(nqd - 6, 20.0)], bs)
    inp = dict(q_init=q.reshape(-1), qd_init=rng.randn(bs * nqd) * 0.02, torques=rng.randn(T, bs * nqd) * 0.02,
               res_f=rng.randn(T, bs * nb, 6) * 0.02, refs=rng.uniform(-0.05, 0.05, (T, bs * nqd)), target_ke=ke, target_kd=ke * 0.01,
               body_mass=mass, body_inv_mass=1 / mass, body_inertia=inertia, body_inv_inertia=np.linalg.inv(inertia),
               adj_pos=rng.randn(3, bs * nb, 7) * 1e-3, adj_vel=rng.randn(3, bs * nb, 6) * 1e-3)
    inp = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in inp.items()}
    inp.update(frame2step=[0, 9, 20], nsteps=T, dt=5e-4)
    dm = hip_backend.DeviceModel(tpl)
    out = gpu_rollout(dm, inp, dev)
    st = RefC(tpl, np.float32).rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    st64 = RefC(tpl, np.float64).rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    assert np.abs(st["grf"]).max() > 0.1 and np.abs(st["jaf"]).max() > 0.1
    for k, floor in (("wp_pos", 2e-5), ("wp_vel", 5e-3), ("grf", 1e-2), ("jaf", 1e-2)):
        e, y = relmax(out[k], st64[k]), relmax(st[k], st64[k])
        assert e < max(floor, 3.0 * y), (k, e, y)
    for k in GRADS:
        assert np.isfinite(out["grads"][k]).all(), k
    own = own_trajectory_check(dm, tpl, inp, dev, abs_floor=1e-8)
    print("%d compound joints: segw %d, own trajectory worst env %.1e, median %.1e" % (n, dm.segment_width(), own["worst"].max(), np.median(own["worst"])))
    assert (own["worst"] <= np.maximum(1e-3, 3.0 * own["fp32_atan2"])).all(), (own["worst"], own["fp32_atan2"])


@pytest.mark.parametrize("name,family", [("laikago", 1), ("laikago", 2), ("human", 0)])
def test_bodies_without_inverse_mass(name, family, dev, oracle_libs):
    """integrate_bodies switches gravity off for a body whose inverse mass is zero (integrator_euler.py:61-62: g * nonzero(inv_m)) --
    no shipped configuration has one.  The root of every second env gets inv_mass = 0 and inv_inertia = 0 (a pinned base that the
    joint and contact forces cannot move), one limb body inv_mass = 0 only: forward and gradients against the C oracle and the
    own-trajectory adjoint."""
    from diffphys_amd import hip_backend, robots, synth
    from helpers import own_trajectory_check
    from oracle.ref_c import RefC

    tpl = robots.load_template(name)
    nb, bs, T = int(tpl["nb"]), 8, 20
    inp = synth.make_inputs(tpl, name, bs=bs, nsteps=T, seed=19, steps_per_frame=9, penetration=0.003)
    im = inp["body_inv_mass"].reshape(bs, nb).copy()
    ii = inp["body_inv_inertia"].reshape(bs, nb, 3, 3).copy()
    im[::2, 0] = 0.0
    ii[::2, 0] = 0.0
    im[1::2, 3] = 0.0
    inp["body_inv_mass"], inp["body_inv_inertia"] = im.reshape(-1), ii.reshape(bs * nb, 3, 3)
    dm = hip_backend.DeviceModel(tpl)
    if family:
        dm.set_kernel_family(family)
    out = gpu_rollout(dm, inp, dev)
    rc = RefC(tpl, np.float32)
    st = rc.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    pos = out["wp_pos"].reshape(-1, bs, nb, 7)
    assert np.abs(pos[-1, ::2, 0, :3] - pos[0, ::2, 0, :3]).max() < 1e-6            # the pinned roots did not move ((x + R com) - R com: an ulp)
    assert np.abs(pos[-1, 1::2, 0, :3] - pos[0, 1::2, 0, :3]).max() > 0.0
    assert relmax(out["wp_pos"], st["wp_pos"]) < 2e-5 and relmax(out["wp_vel"], st["wp_vel"]) < 5e-3
    assert relmax(out["grf"], st["grf"]) < 5e-3 and relmax(out["jaf"], st["jaf"]) < 5e-3
    own = own_trajectory_check(dm, tpl, inp, dev, abs_floor=1e-8)
    print("%s family %d, zero inverse masses: own trajectory worst env %.1e" % (name, family, own["worst"].max()))
    assert (own["worst"] <= np.maximum(1e-3, 3.0 * own["fp32_atan2"])).all(), own["worst"]


@pytest.mark.parametrize("name,family", [("laikago", 1), ("laikago", 2), ("quad", 0)])
def test_other_model_constants(name, family, dev, oracle_libs):
    """Model constants away from the shipped robots' values: tilted gravity, a twice longer time step, contact damping (kd > 0: the
    `fd = min(vn, 0) kd step(c)` term is identically zero for every shipped robot), weaker friction (mu = 0.4, kf = 40, different per
    shape), softer attach springs, contact thickness (contact_dist) on half the points."""
    from diffphys_amd import hip_backend, robots, synth
    from helpers import own_trajectory_check
    from oracle.ref_c import RefC

    tpl = dict(robots.load_template(name))
    nb, bs, T = int(tpl["nb"]), 9, 16
    rng = np.random.RandomState(20)
    tpl["gravity"] = np.array([1.5, -7.0, -2.0], np.float32)
    mats = np.array(tpl["shape_materials"], np.float32).copy()
    mats[:, 1] = 30.0 + 10.0 * rng.rand(len(mats))          # kd
    mats[:, 2] = 40.0 + 20.0 * rng.rand(len(mats))          # kf
    mats[:, 3] = 0.4 + 0.2 * rng.rand(len(mats))            # mu
    tpl["shape_materials"] = mats
    tpl["joint_attach_ke"], tpl["joint_attach_kd"] = np.float32(5000.0), np.float32(60.0)
    cd = np.array(tpl["contact_dist"], np.float32).copy()
    cd[::2] += 0.002
    tpl["contact_dist"] = cd
    inp = synth.make_inputs(tpl, name, bs=bs, nsteps=T, seed=21, steps_per_frame=7, penetration=0.003)
    inp["dt"] = 1e-3
    inp["qd_init"] = (rng.randn(*inp["qd_init"].shape) * 0.3).astype(np.float32)   # normal velocities into the ground: the damping term acts
    dm = hip_backend.DeviceModel(tpl)
    if family:
        dm.set_kernel_family(family)
    out = gpu_rollout(dm, inp, dev)
    rc = RefC(tpl, np.float32)
    st = rc.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    st64 = RefC(tpl, np.float64).rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    base = RefC(robots.load_template(name), np.float32).rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    assert relmax(base["grf"], st["grf"]) > 5e-2 and relmax(base["wp_pos"], st["wp_pos"]) > 1e-4     # the constants matter
    for k, floor in (("wp_pos", 2e-5), ("wp_vel", 5e-3), ("grf", 5e-3), ("jaf", 5e-3)):
        e, y = relmax(out[k], st64[k]), relmax(st[k], st64[k])
        assert e < max(floor, 3.0 * y), (k, e, y)
    own = own_trajectory_check(dm, tpl, inp, dev, abs_floor=1e-8)
    print("%s family %d, other constants: own trajectory worst env %.1e" % (name, family, own["worst"].max()))
    assert (own["worst"] <= np.maximum(1e-3, 3.0 * own["fp32_atan2"])).all(), own["worst"]


def test_reduce_loss_inside_the_launch_on_awkward_tables(dev):
    """reduce_loss(clip=True) as the launch after the rollout does it (pd_trajloss.h), against the reference's per-env loop
    (oracle/pose_torch.reduce_loss_loop = dp_utils.py:93-110, itself held to the reference's own outputs in tests/test_ref_fixtures.py)
    on the SAME table, for tables built to be awkward: env 0 without a positive entry (outseq everywhere / NaN targets: the reference's
    threshold is then NaN and NOTHING is clipped in the whole batch, however large another env's loss), the same batch with env 0 in
    sequence (then that env IS clipped), every entry out of sequence (the "mean of all entries" branch), one env, a
    threshold that clips many envs at different frames, 40 frames.  Also the shares: scale = d loss / d table entry (autograd on
    the loop)."""
    from diffphys_amd import hip_backend, robots, synth
    from oracle.pose_torch import reduce_loss_loop

    tpl = robots.load_template("laikago")
    nb = int(tpl["nb"])
    dm = hip_backend.DeviceModel(tpl)
    g = torch.Generator().manual_seed(31)
    cases = [("first envs empty", 12, 6), ("first envs empty but env 0", 12, 6), ("all out of sequence", 5, 4), ("one env", 1, 7), ("many clipped", 40, 8), ("forty frames", 6, 40), ("far first env", 9, 5)]
    for tag, bs, F in cases:
        T = F - 1
        inp = synth.make_inputs(tpl, "laikago", bs=bs, nsteps=T, seed=23, steps_per_frame=1, penetration=0.002)
        t = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=np.float32)).to(dev) for k in FWD}
        f2s = list(range(F))
        pos0 = dm.rollout_forward(bs, T, inp["dt"], *[t[k] for k in FWD], frame2step=f2s)[0]
        tgt = (pos0.view(F, bs, nb, 7).permute(1, 0, 2, 3) + 0.01 * torch.randn(bs, F, nb, 7, generator=g).to(dev)).contiguous()
        outseq = torch.zeros(bs, F, dtype=torch.bool, device=dev)
        if tag.startswith("first envs empty"):
            outseq[0] = tag == "first envs empty"
            tgt[1] = float("nan")
            outseq[2, : F - 1] = True
            tgt[5, 3:, :, :3] += 1.0
        elif tag == "all out of sequence":
            outseq[:] = True
        elif tag == "many clipped":
            for e in range(1, bs, 2):
                tgt[e, e % F:, :, :3] += 0.5 + 0.1 * e
        elif tag == "far first env":
            tgt[0, :, :, :3] += 2.0       # the threshold comes from env 0's (large) losses: nothing is clipped
            tgt[4, 2:, :, :3] += 0.7
        o = dm.rollout_forward_traj_loss(bs, T, inp["dt"], *[t[k] for k in FWD], frame2step=f2s, target_pos=tgt, outseq=outseq)
        tl = o[5]
        table = tl["table"].detach().clone().double().requires_grad_(True)
        ref = reduce_loss_loop(table.clone(), clip=True)
        (gref,) = torch.autograd.grad(ref, table, allow_unused=True)
        gref = torch.zeros_like(table) if gref is None else gref
        red = tl["reduced"].cpu().numpy()
        print("%-20s bs=%d F=%d: loss %.6e (loop %.6e) th %.3e positives %d clipped envs %d" % (tag, bs, F, red[0], float(ref), red[1], red[2], red[3]))
        assert abs(red[0] - float(ref)) <= 2e-6 * abs(float(ref)) + 1e-12, tag
        assert float((tl["scale"].double() - gref).abs().max()) <= 1e-6 * float(gref.abs().max()) + 1e-12, tag
        if tag == "first envs empty":   # env 0 has no positive entry: NaN threshold, no env clipped (dp_utils.py:98-103 on torch >= 1.8)
            assert np.isnan(red[1]) and red[3] == 0 and float(tl["table"][5].max()) > 0.5
        if tag == "first envs empty but env 0":
            assert np.isfinite(red[1]) and red[3] >= 1
        if tag == "all out of sequence":
            assert red[0] == 0.0 and red[2] == 0
        if tag == "many clipped":
            assert red[3] >= 5
        if tag == "far first env":
            assert red[3] == 0


def test_generic_robot_through_the_loss_and_fk_entries(dev, tmp_path):
    """The toy robot (free + revolute + compound + fixed joints: the GENERIC kernel instantiation) through the row-f4 entries: the
    loss table is se3_loss of the gathered poses, the self-seeded adjoint equals the plain adjoint fed the same seeds, the FK chains
    that ride along are pd_fk_forward / pd_fk_backward's -- for a 6-step rollout and for a zero-step one."""
    from test_host import OBJ, URDF
    from diffphys_amd import dp_utils, hip_backend, sim
    from diffphys_amd.import_urdf import parse_urdf
    from helpers import build_template

    (tmp_path / "toy.urdf").write_text(URDF)
    (tmp_path / "tet.obj").write_text(OBJ)
    b = sim.ModelBuilder()
    parse_urdf(str(tmp_path / "toy.urdf"), b, xform=sim.transform((0, 0.5, 0), sim.quat_identity()), floating=True, density=1000.0,
               armature=0.01, stiffness=220.0, damping=2.0, shape_ke=1e4, shape_kd=10.0, shape_kf=1e2, shape_mu=0.7, limit_ke=50.0, limit_kd=1.0)
    tpl = build_template(b, attach_ke=8000.0, attach_kd=200.0)
    nb, nq, nqd, bs = int(tpl["nb"]), int(tpl["nq"]), int(tpl["nqd"]), 5
    dm = hip_backend.DeviceModel(tpl)
    rng = np.random.RandomState(0)
    g = torch.Generator().manual_seed(41)
    for T, f2s in ((6, [0, 3, 6]), (0, [0])):
        F = len(f2s)
        q = np.tile(tpl["joint_q"].astype(np.float64), (bs, 1))
        q[:, 1] = 0.13 + rng.rand(bs) * 0.02
        q[:, 7:] = rng.uniform(-0.4, 0.4, (bs, nq - 7))
        mass = np.tile(tpl["body_mass"].astype(np.float64), bs)
        inertia = np.tile(tpl["body_inertia"].astype(np.float64), (bs, 1, 1))
        ke = np.tile(np.r_[np.zeros(6), np.full(nqd - 6, 60.0)], bs)
        inp = dict(q_init=q.reshape(-1), qd_init=rng.randn(bs * nqd) * 0.2, torques=rng.randn(T, bs * nqd) * 0.3, res_f=rng.randn(T, bs * nb, 6) * 0.3,
                   refs=rng.uniform(-0.3, 0.3, (T, bs * nqd)), target_ke=ke, target_kd=ke * 0.02, body_inv_mass=1 / mass, body_inertia=inertia,
                   body_inv_inertia=np.linalg.inv(inertia))
        t = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(dev) for k, v in inp.items()}
        jq = (t["q_init"].view(1, bs, nq) + 0.05 * torch.randn(3, bs, nq, generator=g).to(dev)).contiguous()
        jqd = (0.2 * torch.randn(3, bs, nqd, generator=g)).to(dev).contiguous()
        pos0 = dm.rollout_forward(bs, T, 5e-4, *[t[k] for k in FWD], frame2step=f2s)[0]
        tgt = (pos0.view(F, bs, nb, 7).permute(1, 0, 2, 3) + 0.02 * torch.randn(bs, F, nb, 7, generator=g).to(dev)).contiguous()
        o = dm.rollout_forward_traj_loss(bs, T, 5e-4, *[t[k] for k in FWD], frame2step=f2s, target_pos=tgt, fk=(jq, jqd))
        assert torch.equal(o[0], pos0)
        want = dp_utils.se3_loss(pos0.view(F, bs, nb, 7).permute(1, 0, 2, 3).contiguous(), tgt).mean(-1)
        assert float((o[5]["table"] - want).abs().max()) <= 2e-6 * float(want.abs().max())
        wq, wqd = dm.fk_forward(jq.view(3 * bs, nq), jqd.view(3 * bs, nqd))
        assert torch.equal(o[5]["fk_body_q"], wq.view(3, bs, nb, 7).permute(1, 0, 2, 3)) and torch.equal(o[5]["fk_body_qd"], wqd.view(3, bs, nb, 6).permute(1, 0, 2, 3))
        aq, aqd = torch.randn(bs, 3, nb, 7, generator=g).to(dev), torch.randn(bs, 3, nb, 6, generator=g).to(dev)
        one = torch.ones(1, device=dev)
        gr = dm.rollout_backward_traj_loss(bs, T, 5e-4, *[t[k] for k in BWD], f2s, o[4], o[5], one, fk=(jq, jqd, aq, aqd))
        wgq, wgqd = dm.fk_backward(jq.view(3 * bs, nq), jqd.view(3 * bs, nqd), aq.permute(1, 0, 2, 3).contiguous(), aqd.permute(1, 0, 2, 3).contiguous())
        assert torch.equal(gr["fk_joint_q"].view(3 * bs, nq), wgq) and torch.equal(gr["fk_joint_qd"].view(3 * bs, nqd), wgqd)
        seeds = (o[5]["seed_pos"].view(F, bs, nb, 7) * (o[5]["scale"].t() / nb)[:, :, None, None]).reshape(F, bs * nb, 7).contiguous()
        g2 = dm.rollout_backward(bs, T, 5e-4, *[t[k] for k in BWD], f2s, o[4], seeds, torch.zeros(F, bs * nb, 6, device=dev))
        for k in g2:
            if g2[k].numel() == 0:   # (the per-step gradients of a zero-step rollout)
                continue
            assert bool(torch.isfinite(gr[k]).all()), k
            assert float((gr[k] - g2[k]).abs().max()) <= 1e-5 * float(g2[k].abs().max()) + 1e-30, (k, T)


@pytest.mark.parametrize("bs", [64, 2100])
def test_graph_capture_of_the_fused_iteration(bs, dev):
    """The row-f4 form of an iteration -- rollout with the loss inside, reduce_loss + FK forward in one launch, seeds + FK backward in
    one launch, adjoint rollout: four launches -- captured into ONE HIP graph and replayed bit for bit (quad-lane kernels at 64 envs,
    lane per body at 2100)."""
    from diffphys_amd import hip_backend, robots, synth

    tpl = robots.load_template("laikago")
    dm = hip_backend.DeviceModel(tpl)
    T, nb, nq, nqd = 34, 13, 19, 18
    inp = synth.make_inputs(tpl, "laikago", bs=bs, nsteps=T, seed=2, penetration=0.002)
    t = {k: torch.from_numpy(inp[k]).to(dev) for k in INPUT_NAMES}
    f2s = list(inp["frame2step"])
    F = len(f2s)
    g = torch.Generator().manual_seed(5)
    pos0 = dm.rollout_forward(bs, T, inp["dt"], *[t[k] for k in FWD], frame2step=f2s)[0]
    tgt = (pos0.view(F, bs, nb, 7).permute(1, 0, 2, 3) + 0.02 * torch.randn(bs, F, nb, 7, generator=g).to(dev)).contiguous()
    jq = (t["q_init"].view(1, bs, nq) + 0.05 * torch.randn(F, bs, nq, generator=g).to(dev)).contiguous()
    jqd = torch.zeros(F, bs, nqd, device=dev)
    aq, aqd = torch.randn(bs, F, nb, 7, generator=g).to(dev), torch.randn(bs, F, nb, 6, generator=g).to(dev)
    gain = torch.full((1,), 0.37, device=dev)

    def run():
        o = dm.rollout_forward_traj_loss(bs, T, inp["dt"], *[t[k] for k in FWD], frame2step=f2s, target_pos=tgt, fk=(jq, jqd))
        gr = dm.rollout_backward_traj_loss(bs, T, inp["dt"], *[t[k] for k in BWD], f2s, o[4], o[5], gain, fk=(jq, jqd, aq, aqd))
        return o[0], o[5]["reduced"], o[5]["fk_body_q"], gr["q_init"], gr["refs"], gr["fk_joint_q"]

    ref = [x.clone() for x in run()]
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
        with torch.cuda.graph(graph, stream=side):
            cap = run()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        for x in cap:
            x.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(cap, ref))
    assert float(ref[1][0]) > 0 and float(ref[3].abs().max()) > 0 and float(ref[5].abs().max()) > 0


@pytest.mark.parametrize("name,family", [("laikago", 1), ("laikago", 2), ("human", 0)])
def test_a_nan_env_stays_in_its_lanes(name, family, dev):
    """Envs share wavefronts (four Laikago envs per body wave, votes and compaction across the wave): an env whose inputs are NaN / inf
    must not change ANY bit of its neighbours' results, must not hang the speculative sweep, and its own gradients come back scrubbed
    (NaN -> 0, the boundary's remove_nan)."""
    from diffphys_amd import hip_backend, robots, synth

    tpl = robots.load_template(name)
    nb, bs, T = int(tpl["nb"]), 11, 30
    inp = synth.make_inputs(tpl, name, bs=bs, nsteps=T, seed=24, steps_per_frame=9, penetration=0.003)
    dm = hip_backend.DeviceModel(tpl)
    if family:
        dm.set_kernel_family(family)
    clean = gpu_rollout(dm, inp, dev)
    bad = {k: np.array(v, copy=True) if isinstance(v, np.ndarray) else v for k, v in inp.items()}
    nq, nqd = int(tpl["nq"]), int(tpl["nqd"])
    bad["q_init"].reshape(bs, nq)[2, 8] = np.nan                 # a joint angle
    bad["qd_init"].reshape(bs, nqd)[5, 1] = np.inf               # a root velocity
    bad["refs"].reshape(T, bs, nqd)[7, 6, 9] = np.nan            # a control, from step 7 on
    bad["body_inv_mass"].reshape(bs, nb)[9, 3] = np.nan
    out = gpu_rollout(dm, bad, dev)
    good = [e for e in range(bs) if e not in (2, 5, 6, 9)]
    F = clean["wp_pos"].shape[0]
    for k, w in (("wp_pos", 7), ("wp_vel", 6), ("grf", 6), ("jaf", 6)):
        a, b = clean[k].reshape(F, bs, nb, w)[:, good], out[k].reshape(F, bs, nb, w)[:, good]
        assert np.array_equal(a, b), k
    for k, v in out["grads"].items():
        assert not np.isnan(v).any(), k                            # scrubbed at the stores (inf may remain, as in the reference)
    per_env = lambda k, v: v.reshape(T, bs, -1).transpose(1, 0, 2).reshape(bs, -1) if k in ("torques", "res_f", "refs") else v.reshape(bs, -1)
    for k in out["grads"]:
        assert np.array_equal(per_env(k, clean["grads"][k])[good], per_env(k, out["grads"][k])[good]), k


@pytest.mark.parametrize("R,T,f2s,need_gb", [(256, 8, [3, 8], 30), (2500, 2, [1], 110)], ids=["1M-envs-8-steps", "10M-envs-2-steps"])
def test_a_million_envs_past_four_gib(R, T, f2s, need_gb, dev):
    """Maximum sizes.  (a) 1 048 576 Laikago envs -- 256 copies of 4 096 distinct ones -- x 8 steps, 2 frames.  13.6 M bodies; the saved
    trajectory + hit log is 9.8 GB, so every step base from the fourth on lies past 4 GiB (64-bit bases under the kernels' 32-bit lane
    offsets), the frame outputs and per-step gradients are 0.3-0.6 GB each.  Envs are independent: every copy must reproduce copy 0 BIT
    FOR BIT (outputs and all 10 gradients, compared on the device), and copy 0 must be the 4 096-env launch on its own.  The library
    refuses a batch whose lane offsets would not fit 32 bits (bs x bodies >= 2^27 or bs >= 2^24) instead of wrapping around.
    (b) 10 240 000 envs x 2 steps, 133 M bodies: just under that limit -- the largest lane offsets the kernels ever form (2.1e9 bytes
    into a step's planes, 1.3e9 into its hit log), 12 GB per step of workspace, ~80 GB in all."""
    from diffphys_amd import hip_backend, robots, synth

    free, _ = torch.cuda.mem_get_info()
    if free < need_gb * 1e9:
        pytest.skip("needs ~%d GB of device memory" % need_gb)
    tpl = robots.load_template("laikago")
    bs0 = 4096
    inp = synth.make_inputs(tpl, "laikago", bs=bs0, nsteps=T, seed=12, seqs=("mi-trot", "mi-spin"), penetration=0.003)
    inp["frame2step"] = f2s
    F, nb = len(f2s), 13
    rng = np.random.RandomState(5)
    inp["adj_pos"] = (rng.randn(F, bs0 * nb, 7) * 1e-2).astype(np.float32)
    inp["adj_vel"] = (rng.randn(F, bs0 * nb, 6) * 1e-2).astype(np.float32)
    lead = dict(GRAD_LEAD, body_mass=0, adj_pos=1, adj_vel=1)
    t0 = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=np.float32)).to(dev) for k in INPUT_NAMES + ("adj_pos", "adj_vel")}

    def tiled(k):   # the env axis (after the step / frame axis where there is one) repeated R times
        x = t0[k]
        if lead[k]:
            y = x.reshape(x.shape[0], bs0, -1).repeat(1, R, 1)
            return y.reshape((x.shape[0], R * x.shape[1]) + tuple(x.shape[2:]))
        y = x.reshape(bs0, -1).repeat(R, 1)
        return y.reshape((R * x.shape[0],) + tuple(x.shape[1:]))

    dm = hip_backend.DeviceModel(tpl)
    pos0, vel0, grf0, jaf0, ws0 = dm.rollout_forward(bs0, T, inp["dt"], *[t0[k] for k in FWD], frame2step=f2s)
    g0 = dm.rollout_backward(bs0, T, inp["dt"], *[t0[k] for k in BWD], f2s, ws0, t0["adj_pos"], t0["adj_vel"])
    assert float(grf0.abs().max()) > 1.0, "contacts must be active"
    t = {k: tiled(k) for k in t0}
    bs = bs0 * R
    assert dm.workspace_floats(bs, T) * 4 > 2 * 2 ** 32 and bs * nb < 2 ** 27
    pos, vel, grf, jaf, ws = dm.rollout_forward(bs, T, inp["dt"], *[t[k] for k in FWD], frame2step=f2s)
    g = dm.rollout_backward(bs, T, inp["dt"], *[t[k] for k in BWD], f2s, ws, t["adj_pos"], t["adj_vel"])
    torch.cuda.synchronize()

    def copies_equal(big, small, has_lead, what):
        a = big.reshape(big.shape[0], R, -1) if has_lead else big.reshape(R, -1)
        b = small.reshape(small.shape[0], 1, -1) if has_lead else small.reshape(1, -1)
        same = (a == b) | (torch.isnan(a) & torch.isnan(b))
        assert bool(same.all()), (what, int((~same).sum()))

    for name, big, small in (("wp_pos", pos, pos0), ("wp_vel", vel, vel0), ("grf", grf, grf0), ("jaf", jaf, jaf0)):
        copies_equal(big, small, True, name)
    for k in g0:
        assert bool(torch.isfinite(g[k]).all()), k
        copies_equal(g[k], g0[k], bool(GRAD_LEAD[k]), "grad " + k)
    del pos, vel, grf, jaf, ws, g, t
    torch.cuda.empty_cache()
    # past the supported range: refused with a message before anything is looked at or launched
    import ctypes

    lib = hip_backend.lib()
    f2s_c = (ctypes.c_int * 1)(0)
    for too_many in ((2 ** 27 + nb - 1) // nb, 2 ** 24):
        assert lib.pd_rollout_forward(dm.h, too_many, 1, ctypes.c_float(5e-4), *([None] * 10), 0, f2s_c, *([None] * 5), None) != 0
        assert "batch too large" in lib.pd_last_error().decode()
        assert lib.pd_rollout_backward(dm.h, too_many, 1, ctypes.c_float(5e-4), *([None] * 9), 0, f2s_c, *([None] * 13), None) != 0
        assert "batch too large" in lib.pd_last_error().decode()


def test_a_million_envs_through_the_fused_loss_and_fk_entries(dev):
    """Maximum sizes, row f4's entries: 1 048 576 envs (256 copies of 4 096) x 8 steps through pd_rollout_forward_traj_loss_fk /
    pd_rollout_backward_traj_loss_fk -- the [bs, F] loss table (8 MB: the global-memory path of the reduce launch, not the LDS one),
    2 M FK chains riding on the reduce / seeds launches.  Every copy equals copy 0 bit for bit (table, seeds, FK poses, the 10 + 2
    gradients); the threshold is env 0's -- the same as the 4 096-env launch's; the reduced loss agrees with it to summation order;
    the gradients are the small launch's / 256 exactly (the mean's 1 / entries, a power of two)."""
    from diffphys_amd import hip_backend, robots, synth

    free, _ = torch.cuda.mem_get_info()
    if free < 40e9:
        pytest.skip("needs ~30 GB of device memory")
    tpl = robots.load_template("laikago")
    bs0, R, T, f2s = 4096, 256, 8, [3, 8]
    inp = synth.make_inputs(tpl, "laikago", bs=bs0, nsteps=T, seed=13, seqs=("mi-trot", "mi-spin"), penetration=0.003)
    F, nb, nq, nqd = len(f2s), 13, 19, 18
    rng = np.random.RandomState(6)
    t0 = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=np.float32)).to(dev) for k in INPUT_NAMES}
    dm = hip_backend.DeviceModel(tpl)
    pos_plain = dm.rollout_forward(bs0, T, inp["dt"], *[t0[k] for k in FWD], frame2step=f2s)[0]
    tgt0 = pos_plain.reshape(F, bs0, nb, 7).permute(1, 0, 2, 3).contiguous().clone()
    tgt0[..., :3] += torch.from_numpy((rng.randn(bs0, F, nb, 3) * 0.02).astype(np.float32)).to(dev)
    tgt0[7, :, :, 1] += 3.0          # one env far off: clipped by reduce_loss
    outseq0 = torch.zeros(bs0, F, dtype=torch.bool, device=dev)
    outseq0[11, 1] = True
    jq0 = torch.from_numpy(np.tile(inp["q_init"].reshape(1, bs0, nq), (F, 1, 1)).astype(np.float32)).to(dev)
    jq0[1, :, 7:] += 0.05
    jqd0 = torch.from_numpy((rng.randn(F, bs0, nqd) * 0.1).astype(np.float32)).to(dev)
    aq0 = torch.from_numpy((rng.randn(bs0, F, nb, 7) * 1e-2).astype(np.float32)).to(dev)
    aqd0 = torch.from_numpy((rng.randn(bs0, F, nb, 6) * 1e-2).astype(np.float32)).to(dev)
    g_loss = torch.ones(1, device=dev)

    def run(bs, t, tgt, outseq, jq, jqd, aq, aqd):
        pos, vel, grf, jaf, ws, tl = dm.rollout_forward_traj_loss(bs, T, inp["dt"], *[t[k] for k in FWD], frame2step=f2s, target_pos=tgt, outseq=outseq,
                                                                  fk=(jq, jqd))
        g = dm.rollout_backward_traj_loss(bs, T, inp["dt"], *[t[k] for k in BWD], f2s, ws, tl, g_loss, fk=(jq, jqd, aq, aqd))
        return pos, tl, g

    pos_s, tl_s, g_s = run(bs0, t0, tgt0, outseq0, jq0, jqd0, aq0, aqd0)
    red_s = tl_s["reduced"].cpu().numpy()
    assert red_s[3] >= 1 and np.isfinite(red_s).all()       # something was clipped in the small launch
    lead = dict(GRAD_LEAD, body_mass=0)

    def tiled(x, has_lead):
        if has_lead:
            y = x.reshape(x.shape[0], bs0, -1).repeat(1, R, 1)
            return y.reshape((x.shape[0], R * x.shape[1]) + tuple(x.shape[2:]))
        y = x.reshape(bs0, -1).repeat(R, 1)
        return y.reshape((R * x.shape[0],) + tuple(x.shape[1:]))

    t = {k: tiled(t0[k], bool(lead[k])) for k in t0}
    bs = bs0 * R
    pos, tl, g = run(bs, t, tiled(tgt0, False), tiled(outseq0.to(torch.uint8), False).to(torch.bool), tiled(jq0, True), tiled(jqd0, True),
                     tiled(aq0, False), tiled(aqd0, False))
    torch.cuda.synchronize()

    def copies_equal(big, small, has_lead, what, factor=1.0):
        a = big.reshape(big.shape[0], R, -1) if has_lead else big.reshape(R, -1)
        b = (small.reshape(small.shape[0], 1, -1) if has_lead else small.reshape(1, -1)) * factor
        same = (a == b) | (torch.isnan(a) & torch.isnan(b))
        assert bool(same.all()), (what, int((~same).sum()))

    copies_equal(pos, pos_s, True, "wp_pos")
    copies_equal(tl["table"], tl_s["table"], False, "loss table")
    copies_equal(tl["seed_pos"], tl_s["seed_pos"], True, "seed_pos")
    copies_equal(tl["seed_gt"], tl_s["seed_gt"], False, "seed_gt")
    copies_equal(tl["fk_body_q"], tl_s["fk_body_q"], False, "fk body_q")
    copies_equal(tl["fk_body_qd"], tl_s["fk_body_qd"], False, "fk body_qd")
    red = tl["reduced"].cpu().numpy()
    assert red[1] == red_s[1] and red[3] == R * red_s[3] and red[2] == R * red_s[2]      # env 0's threshold; counts scale with the copies
    assert abs(red[0] - red_s[0]) <= 1e-5 * abs(red_s[0])
    copies_equal(tl["scale"], tl_s["scale"], False, "scale", 1.0 / R)
    for k in GRAD_LEAD:
        copies_equal(g[k], g_s[k], bool(GRAD_LEAD[k]), "grad " + k, 1.0 / R)
    # (the FK adjoint's seeds do not go through the mean: the same values in every copy)
    copies_equal(g["fk_joint_q"], g_s["fk_joint_q"], True, "fk grad q")
    copies_equal(g["fk_joint_qd"], g_s["fk_joint_qd"], True, "fk grad qd")
