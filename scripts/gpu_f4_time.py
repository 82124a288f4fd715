#!/usr/bin/env python3
"""Row f4 timing: one rollout iteration (forward, trajectory loss, backward) through the reference's sequence ForwardWarp -> se3_loss
-> reduce_loss(clip=True) -> autograd, against ForwardWarpTrajLoss (loss inside the rollout, self-seeding adjoint).  Device time per
iteration from events around back-to-back iterations, and the launch durations of the two rollout kernels in each mode.
usage: gpu_f4_time.py [robot:bs ...]   (default laikago:4096 laikago:512 human:1024)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ppr-diffphys_amd")]
import numpy as np, torch
from diffphys_amd import dp_model, dp_utils, hip_backend, robots, synth

dev = torch.device("cuda:0")
for cfg in (sys.argv[1:] or ["laikago:4096", "laikago:512", "human:1024"]):
    name, bs = cfg.split(":"); bs = int(bs)
    tpl = robots.load_template(name)
    T = 100
    inp = synth.make_inputs(tpl, name, bs=bs, nsteps=T, seed=0, seqs=("mi-trot", "mi-spin"))
    f2s = list(inp["frame2step"]); F = len(f2s); nb = int(tpl["nb"])

    class Host: pass
    h = Host()
    h.env = robots.env_from_template(name, bs, device=dev)
    h.num_envs, h.steps_idx, h.frame2step, h.dt = bs, range(T), f2s, inp["dt"]
    t = {k: torch.from_numpy(inp[k]).to(dev).requires_grad_(True) for k in synth.INPUT_NAMES}
    args = [t[k] for k in synth.INPUT_NAMES]
    with torch.no_grad():
        pos0, _ = dp_model.ForwardWarp.apply(*args, h)
    tgt = (pos0.reshape(F, bs, nb, 7).permute(1, 0, 2, 3) + 0.02 * torch.randn(bs, F, nb, 7, device=dev)).contiguous()
    outseq = torch.zeros(bs, F, dtype=torch.bool, device=dev)
    dm = hip_backend.device_model(h.env)

    # the control reference whose FK the reference evaluates beside the rollout (dp_model.py:758): F x bs chains
    nq, nqd = int(tpl["nq"]), int(tpl["nqd"])
    qq = (torch.from_numpy(inp["q_init"]).view(1, bs, nq) + 0.05 * torch.randn(F, bs, nq)).to(dev).requires_grad_(True)
    qqd = (0.1 * torch.randn(F, bs, nqd)).to(dev).requires_grad_(True)
    w_q, w_qd = torch.randn(bs, F, nb, 7, device=dev) * 1e-3, torch.randn(bs, F, nb, 6, device=dev) * 1e-3
    args_fk = args + [qq, qqd]

    def seq_torch():
        pos, vel = dp_model.ForwardWarp.apply(*args, h)
        lt = dp_utils.se3_loss(pos.reshape(F, bs, nb, 7).permute(1, 0, 2, 3), tgt).mean(-1)
        lt = torch.where(outseq, torch.zeros_like(lt), lt)
        (dp_utils.reduce_loss(lt, clip=True) * 0.1).backward()

    def seq_fused():
        loss, _, _ = dp_model.ForwardWarpTrajLoss.apply(*args, tgt, outseq, h)
        (loss * 0.1).backward()

    def seq_fused_fk_apart():   # loss inside the rollout, ForwardKinematics as its own launches (+ its permuted copies)
        loss, _, _ = dp_model.ForwardWarpTrajLoss.apply(*args, tgt, outseq, h)
        qp, qv, _ = dp_model.ForwardKinematics.apply(qq, qqd, h.env)
        (loss * 0.1 + (qp * w_q).sum() + (qv * w_qd).sum()).backward()

    def seq_fused_fk_riding():  # ForwardWarpTrajLossFK: the FK chains ride on the reduce / seeds launches
        loss, _, _, qp, qv, _ = dp_model.ForwardWarpTrajLossFK.apply(*args, tgt, outseq, qq, qqd, h)
        (loss * 0.1 + (qp * w_q).sum() + (qv * w_qd).sum()).backward()

    res = {}
    for tag, fn in (("torch sequence", seq_torch), ("fused", seq_fused), ("fused + FK apart", seq_fused_fk_apart), ("fused + FK riding", seq_fused_fk_riding)) * 2:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        dm.set_timing(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            for v in args_fk: v.grad = None
            fn()
        e1.record(); torch.cuda.synchronize()
        res.setdefault(tag, []).append((e0.elapsed_time(e1) / n, dm.last_kernel_ms(0), dm.last_kernel_ms(1)))
        dm.set_timing(False)
    for tag, v in res.items():
        it, kf, kb = np.min([x[0] for x in v]), np.min([x[1] for x in v]), np.min([x[2] for x in v])
        print("F4TIME %-8s bs=%-5d %-18s iteration %.3f ms   (rollout forward launch %.3f ms, adjoint launch %.3f ms)" % (name, bs, tag, it, kf, kb), flush=True)
