"""Row f4 held to the ORACLE directly, not to the product's own unfused path:

    float64 C oracle rollout  ->  oracle/pose_torch.se3_loss  ->  loss_traj[outseq] = 0  ->  the reference's literal reduce_loss loop
    (held to the reference's own outputs in tests/test_ref_fixtures.py)  ->  autograd's d loss / d pose as the seeds of the float64 C
    oracle's adjoint rollout;  the oracle's FK and FK adjoint with ForwardKinematics.backward's post-processing

against pd_rollout_forward_traj_loss_fk / pd_rollout_backward_traj_loss_fk through the C ABI: loss, the [bs, F] table, threshold and
counts, each entry's share, the seeds, the 10 rollout gradients, the FK poses / twists and the FK gradients.  Reference lines:
dp_model.py:733-779 (ForwardWarp, se3_loss(..).mean(-1), outseq, reduce_loss(clip=True)), :758 + :1022-1130 (the control reference's
FK), dp_utils.py:93-138.  A NaN target: autograd through the reference's `loss[nanid] = 0` yields 0 x NaN = NaN in that pose's seed,
the adjoint spreads it over the env and remove_nan zeroes what it reached -- the env's gradient is dropped; the same must happen here
(NaN pattern of the seeds identical; the env's q_init gradient exactly zero on both sides).  Tolerances are those of test_gpu_parity.test_vs_c_oracle_fresh_seed (gradients: max(2e-2, 2 x the fp32 C
oracle through the same pipeline)); cases: a clipped env, an out-of-sequence env, a NaN target, and env 0 EMPTY (no env clipped)."""
import numpy as np
import pytest
import torch

from helpers import relmax, toy_inputs, toy_template
from test_gpu_parity import BWD, FWD, GRADS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "run with -m gpu on a GPU box"
    return torch.device("cuda:0")


def _oracle_pipeline(rc, inp, tgt, outseq, wt, qq, qqd, w_q, w_qd):
    """everything the fused entries compute, by the oracle in rc's precision (targets / weights are float64 numpy)"""
    from oracle import pose_torch

    nb = rc.nb
    T, f2s = inp["nsteps"], inp["frame2step"]
    F, bs = len(f2s), inp["q_init"].size // rc.nq
    st = rc.rollout_forward(inp, T, f2s, inp["dt"])
    tdt = torch.float64 if rc.dtype == np.float64 else torch.float32
    pos = torch.tensor(np.asarray(st["wp_pos"]), dtype=tdt, requires_grad=True)         # [F, bs*nb, 7]
    tg = torch.tensor(tgt, dtype=tdt, requires_grad=True)
    sim = pos.reshape(F, bs, nb, 7).permute(1, 0, 2, 3)
    lt = pose_torch.se3_loss(sim, tg).mean(-1)                                            # dp_model.py:777
    lt = torch.where(torch.tensor(outseq), torch.zeros_like(lt), lt)                      # :778  loss_traj[outseq_idx] = 0
    table = lt.detach().clone()
    work = lt * 1.0
    loss = pose_torch.reduce_loss_loop(work, clip=True)                                   # :779, the reference's loop
    share = torch.autograd.grad(loss, lt, retain_graph=True)[0]
    (loss * wt).backward()
    clipped = int(((work.detach() == 0) & (table != 0)).any(1).sum())
    row0 = table[0][table[0] > 0]
    th = float(row0.median() * 10) if row0.numel() else float("nan")
    adj_pos = pos.grad.numpy()
    g = rc.rollout_backward(st, adj_pos, np.zeros((F, bs * nb, 6)))
    g = {k: np.nan_to_num(np.asarray(v, np.float64), nan=0.0) for k, v in g.items()}          # remove_nan, dp_model.py:1294-1384
    # the control reference's FK: chains come frame-major [F, bs, .], rows env-major [bs, F, nb, .]
    bq, bqd = rc.fk_forward(qq.reshape(F * bs, -1), qqd.reshape(F * bs, -1))
    perm = lambda a: a.reshape(F, bs, nb, -1).transpose(1, 0, 2, 3)
    aq = np.ascontiguousarray(w_q.transpose(1, 0, 2, 3)).reshape(F * bs, nb, 7)
    aqd = np.ascontiguousarray(w_qd.transpose(1, 0, 2, 3)).reshape(F * bs, nb, 6)
    gq, gqd = rc.fk_backward(qq.reshape(F * bs, -1), qqd.reshape(F * bs, -1), bq, aq, aqd)   # (the NaN seed goes in: it spreads, then NaN -> 0)
    post = lambda a: np.minimum(np.nan_to_num(a, nan=0.0), 1.0)                            # dp_model.py:1109-1110, 1122-1123
    return dict(loss=float(loss.detach()), table=table.numpy(), th=th, positives=int((work.detach() > 0).sum()), clipped=clipped,
                share=share.numpy(), seed=adj_pos, grads=g, g_tgt=tg.grad.numpy(), wp_pos=np.asarray(st["wp_pos"]),
                fk_q=perm(bq), fk_qd=perm(bqd), fk_gq=post(gq).reshape(F, bs, -1), fk_gqd=post(gqd).reshape(F, bs, -1))


@pytest.mark.parametrize("case", ["laikago-37", "laikago-37-env0-empty", "human-9", "generic-11", "generic-11-env0-empty"])
def test_fused_entries_vs_float64_oracle_pipeline(case, dev, oracle_libs, tmp_path):
    from diffphys_amd import hip_backend, robots, synth
    from oracle.ref_c import RefC

    name, bs = case.split("-")[0], int(case.split("-")[1])
    env0_empty = case.endswith("env0-empty")
    T, f2s = 31, [0, 10, 20, 30, 31]
    F = len(f2s)
    if name == "generic":
        tpl = toy_template(tmp_path)
        inp = toy_inputs(tpl, bs, T, f2s, seed=3)
    else:
        tpl = robots.load_template(name)
        inp = synth.make_inputs(tpl, name, bs=bs, nsteps=T, seed=21, steps_per_frame=10, penetration=0.002)
        inp["frame2step"] = f2s
        rng0 = np.random.RandomState(2)
        inp["qd_init"] = (rng0.randn(*inp["qd_init"].shape) * 0.1).astype(np.float32)
        inp["torques"] = (rng0.randn(*inp["torques"].shape) * 0.5).astype(np.float32)
    nb, nq, nqd = int(tpl["nb"]), int(tpl["nq"]), int(tpl["nqd"])
    rc64, rc32 = RefC(tpl, np.float64), RefC(tpl, np.float32)
    if name == "generic":  # FIXED joint: the scale-invariant evaluation on both sides (PD_NUM_STABLE; helpers.own_trajectory_check)
        rc64.set_twist_eval(True); rc32.set_twist_eval(True)
    try:
        rng = np.random.RandomState(17)
        pos64 = np.asarray(rc64.rollout_forward(inp, T, f2s, inp["dt"])["wp_pos"]).reshape(F, bs, nb, 7).transpose(1, 0, 2, 3)
        tgt = pos64 + rng.randn(bs, F, nb, 7) * 0.02
        tgt[3, 2:, :, :3] += 0.8                   # env 3: far from its targets from frame 2 on => clipped there (unless env 0 is empty)
        tgt[1, 4, 2, 2] = np.nan                   # one NaN target pose: se3_loss ignores it
        outseq = np.zeros((bs, F), dtype=bool)
        outseq[2, 3:] = True                       # env 2: frames 3.. belong to another clip
        if env0_empty:
            outseq[0] = True                       # env 0 entirely out of sequence: NaN threshold, NOTHING is clipped (dp_utils.py:98-103)
        tgt = np.ascontiguousarray(tgt.astype(np.float32)).astype(np.float64)   # what the device sees
        wt = 0.37
        q0 = inp["q_init"].reshape(bs, nq).astype(np.float64)
        qq = np.ascontiguousarray((q0[None] + 0.1 * rng.randn(F, bs, nq)).astype(np.float32))
        qqd = np.ascontiguousarray((0.5 * rng.randn(F, bs, nqd)).astype(np.float32))
        w_q = (rng.randn(bs, F, nb, 7) * 3.0).astype(np.float32)    # some FK gradients land above 1: the clamp acts
        w_q[0, 1, 2, 0] = np.nan                                      # ... and a NaN seed
        w_qd = rng.randn(bs, F, nb, 6).astype(np.float32)
        o64 = _oracle_pipeline(rc64, inp, tgt, outseq, wt, qq.astype(np.float64), qqd.astype(np.float64), w_q.astype(np.float64), w_qd.astype(np.float64))
        o32 = _oracle_pipeline(rc32, inp, tgt, outseq, wt, qq, qqd, w_q, w_qd)
    finally:
        rc64.set_twist_eval(False); rc32.set_twist_eval(False)
    assert o64["clipped"] == (0 if env0_empty else 1), o64["clipped"]
    assert np.isnan(o64["th"]) == env0_empty

    # ---- the fused entries through the C ABI
    dm = hip_backend.DeviceModel(tpl)
    t = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=np.float32)).to(dev) for k in set(FWD) | set(BWD)}
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    tgt_d, outseq_d = d(tgt), torch.from_numpy(outseq).to(dev)
    qq_d, qqd_d = d(qq), d(qqd)
    pos, vel, grf, jaf, ws, tl = dm.rollout_forward_traj_loss(bs, T, inp["dt"], *[t[k] for k in FWD], frame2step=f2s, target_pos=tgt_d,
                                                              outseq=outseq_d, rot_ratio=0.1, want_seed_gt=True, fk=(qq_d, qqd_d))
    red = tl["reduced"].cpu().numpy()
    print("%s: loss %.6e (oracle %.6e), threshold %.4e (oracle %.4e), positives %d (%d), clipped envs %d (%d)" % (
        case, red[0], o64["loss"], red[1], o64["th"], red[2], o64["positives"], red[3], o64["clipped"]))
    assert relmax(pos.cpu().numpy(), o64["wp_pos"]) < 1e-5
    # the table: se3_loss(...).mean(-1) of fp32 poses; rotation part = 0.1 x acos near small angles
    assert relmax(tl["table"].cpu().numpy(), o64["table"]) < max(2e-4, 3 * relmax(o32["table"], o64["table"]))
    assert abs(red[0] - o64["loss"]) <= max(2e-4, 3 * abs(o32["loss"] - o64["loss"]) / abs(o64["loss"])) * abs(o64["loss"])
    assert int(red[2]) == o64["positives"] and int(red[3]) == o64["clipped"]
    if env0_empty:
        assert np.isnan(red[1])
    else:
        assert abs(red[1] - o64["th"]) <= 3e-4 * o64["th"]
    assert relmax(tl["scale"].cpu().numpy(), o64["share"]) < 1e-6          # 1 / N_pos, or 0 for assigned entries
    assert relmax(tl["fk_body_q"].cpu().numpy(), o64["fk_q"]) < 2e-6 and relmax(tl["fk_body_qd"].cpu().numpy(), o64["fk_qd"]) < 2e-6
    aq, aqd = d(w_q), d(w_qd)
    g = dm.rollout_backward_traj_loss(bs, T, inp["dt"], *[t[k] for k in BWD], f2s, ws, tl, torch.full((1,), wt, device=dev), fk=(qq_d, qqd_d, aq, aqd))
    # the seeds the device built for its adjoint = autograd's d (wt loss) / d pose through the oracle pipeline
    seed = tl["work"][: F * bs * nb * 7].view(F, bs * nb, 7).cpu().numpy()
    assert np.array_equal(np.isnan(seed), np.isnan(o64["seed"])) and np.isnan(seed).sum() == 1     # the NaN target's pose: 0 x NaN, as autograd
    assert np.isnan(seed.reshape(F, bs, nb, 7)[4, 1, 2, 2])
    z = lambda a: np.nan_to_num(a, nan=0.0)
    seed, s64, s32 = z(seed), z(o64["seed"]), z(o32["seed"])
    assert relmax(seed, s64) < max(5e-4, 3 * relmax(s32, s64)), (relmax(seed, s64), relmax(s32, s64))
    assert float(tl["work"][F * bs * nb * 7:].abs().max()) == 0
    e3 = np.abs(seed.reshape(F, bs, nb, 7)[2:, 3]).max()
    assert (e3 == 0) != env0_empty, "env 3 is clipped from frame 2 on -- unless env 0 is empty, then nothing is"
    assert np.abs(seed.reshape(F, bs, nb, 7)[3:, 2]).max() == 0                # out of sequence
    # env 1 carries the NaN seed at the final state: its gradient is dropped wherever the NaN reached (everything but the last steps'
    # entries of bodies far from body 2) -- compared exactly where it must be zero, excluded from the tolerance comparison otherwise
    assert np.abs(o64["grads"]["q_init"].reshape(bs, -1)[1]).max() == 0 and float(g["q_init"].view(bs, -1)[1].abs().max()) == 0
    assert np.abs(o64["grads"]["target_ke"].reshape(bs, -1)[1]).max() == 0 and float(g["target_ke"].view(bs, -1)[1].abs().max()) == 0
    from helpers import GRAD_LEAD
    others = np.ones(bs, dtype=bool); others[1] = False
    envs = lambda a, k: (a.reshape(a.shape[0], bs, -1) if GRAD_LEAD[k] else a.reshape(1, bs, -1))[:, others]
    for k in GRADS:
        ref = envs(o64["grads"][k], k)
        got, c32 = envs(g[k].cpu().numpy().astype(np.float64), k), envs(np.asarray(o32["grads"][k], np.float64), k)
        e, e32 = relmax(got, ref), relmax(c32, ref)
        # per env on the tensor's scale: a Laikago rollout is chaotic at its stiff contacts (one env may sit on a contact edge: the bar
        # for the worst env is the fresh-seed bar, max(2e-2, 2 x fp32 oracle) -- for Laikago 5e-2), the typical env must be tight
        pe = np.abs(got - ref).max((0, 2)) / np.abs(ref).max()
        print("   %-18s fused entries vs float64 pipeline %.1e (per env median %.1e)   fp32 oracle pipeline %.1e" % (k, e, np.median(pe), e32))
        assert np.isfinite(e) and e < max(5e-2 if name == "laikago" else 2e-2, 2 * e32), (k, e, e32)
        assert np.median(pe) < (2e-3 if name == "laikago" else 2e-4), (k, float(np.median(pe)))
    assert relmax(g["fk_joint_q"].cpu().numpy(), o64["fk_gq"]) < 5e-6 and relmax(g["fk_joint_qd"].cpu().numpy(), o64["fk_gqd"]) < 5e-6
    assert float(o64["fk_gq"].max()) == 1.0, "the FK clamp must act in this test"
    # d loss / d target as the autograd boundary returns it (dp_model._traj_loss_backward: share x seed_gt / nb)
    k = (tl["scale"] * (wt / nb))[:, :, None, None]
    g_tgt = torch.where(k != 0, tl["seed_gt"] * k, torch.zeros_like(tl["seed_gt"])).cpu().numpy()
    assert np.array_equal(np.isnan(g_tgt), np.isnan(o64["g_tgt"])) and np.isnan(g_tgt).sum() == 1     # d / d (the NaN target component)
    ref_t = np.nan_to_num(o64["g_tgt"], nan=0.0)
    assert relmax(np.nan_to_num(g_tgt, nan=0.0), ref_t) < max(5e-4, 3 * relmax(np.nan_to_num(o32["g_tgt"], nan=0.0), ref_t))
