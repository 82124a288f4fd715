#!/usr/bin/env python3
"""Randomised sweep of the row-f4 entries (pd_rollout_*_traj_loss[_fk]): robots x batch sizes x horizons x frame lists x out-of-sequence /
NaN / far targets x kernel families.  Per case: the loss table = se3_loss of the gathered poses; reduce_loss + shares = the reference's
per-env loop on that table (oracle/pose_torch.py); the self-seeded adjoint = the plain adjoint fed the same seeds; the FK chains that ride
along = pd_fk_forward / pd_fk_backward bit for bit.  Usage: gpu_stress_f4.py [ncases] [seed]; exits non-zero on a violation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ppr-diffphys_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from diffphys_amd import dp_utils, hip_backend, robots, synth
from oracle.pose_torch import reduce_loss_loop

FWD = ("q_init", "qd_init", "torques", "res_f", "refs", "target_ke", "target_kd", "body_inv_mass", "body_inertia", "body_inv_inertia")
BWD = ("q_init", "qd_init", "torques", "refs", "target_ke", "target_kd", "body_inv_mass", "body_inertia", "body_inv_inertia")
dev = torch.device("cuda:0")
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
tpls = {n: robots.load_template(n) for n in ("laikago", "human", "quad")}
dms = {}
bad = 0
for case in range(ncases):
    name = ("laikago", "laikago", "human", "quad")[rng.randint(4)]
    tpl = tpls[name]; nb, nq, nqd = int(tpl["nb"]), int(tpl["nq"]), int(tpl["nqd"])
    fam = int(rng.randint(0, 3)) if name == "laikago" else 0
    bs = int(rng.choice([1, 2, 3, 5, 8, 17, 33, 64, 130]))
    T = int(rng.choice([0, 1, 2, 5, 12, 34]))
    F = int(rng.randint(0, min(T + 1, 6) + 1))
    f2s = sorted(int(x) for x in rng.choice(T + 1, size=F, replace=False)) if F else []
    if rng.rand() < 0.3: rng.shuffle(f2s)
    Ff, bsf = int(rng.randint(1, 4)), int(rng.choice([1, bs, 7]))
    key = (name, fam)
    if key not in dms:
        dms[key] = hip_backend.DeviceModel(tpl)
        if fam: dms[key].set_kernel_family(fam)
    dm = dms[key]
    inp = synth.make_inputs(tpl, name, bs=bs, nsteps=max(T, 1), seed=int(rng.randint(1 << 30)), steps_per_frame=max(1, T // 3), penetration=float(rng.choice([0.0, 0.003])))
    for k in ("torques", "res_f", "refs"): inp[k] = inp[k][:T]
    t = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=np.float32)).to(dev) for k in FWD}
    g = torch.Generator().manual_seed(case)
    pos0 = dm.rollout_forward(bs, T, inp["dt"], *[t[k] for k in FWD], frame2step=f2s)[0]
    tgt = (pos0.view(F, bs, nb, 7).permute(1, 0, 2, 3) + 0.02 * torch.randn(bs, F, nb, 7, generator=g).to(dev)).contiguous()
    outseq = (torch.rand(bs, F, generator=g) < (0.0, 0.2, 1.0)[rng.randint(3)]).to(dev)
    if F and rng.rand() < 0.5: tgt[rng.randint(bs), rng.randint(F)] = float("nan")
    if F and rng.rand() < 0.5: tgt[rng.randint(bs), rng.randint(F):, :, :3] += 1.0
    jq = (torch.from_numpy(np.tile(tpl["joint_q"].astype(np.float32), (Ff, bsf, 1))) + 0.1 * torch.randn(Ff, bsf, nq, generator=g)).to(dev).contiguous()
    jqd = (0.3 * torch.randn(Ff, bsf, nqd, generator=g)).to(dev).contiguous()
    aq, aqd = torch.randn(bsf, Ff, nb, 7, generator=g).to(dev), torch.randn(bsf, Ff, nb, 6, generator=g).to(dev)
    why = []
    try:
        o = dm.rollout_forward_traj_loss(bs, T, inp["dt"], *[t[k] for k in FWD], frame2step=f2s, target_pos=tgt, outseq=outseq, fk=(jq, jqd))
        tl = o[5]
        if not torch.equal(o[0], pos0): why.append("poses")
        if F:
            want = dp_utils.se3_loss(pos0.view(F, bs, nb, 7).permute(1, 0, 2, 3).contiguous(), tgt).mean(-1)
            want = torch.where(outseq, torch.zeros_like(want), want)
            if float((tl["table"] - want).abs().max()) > 2e-6 * float(want.abs().max()) + 1e-12: why.append("table")
        table = tl["table"].detach().clone().double().requires_grad_(True)
        ref = reduce_loss_loop(table.clone(), clip=True) if F else torch.zeros((), dtype=torch.float64)
        gref = torch.autograd.grad(ref, table, allow_unused=True)[0] if F and ref.requires_grad else None
        gref = torch.zeros_like(table) if gref is None else gref
        red = tl["reduced"].cpu().numpy()
        refv = float(ref) if np.isfinite(float(ref)) else 0.0
        if abs(red[0] - refv) > 2e-6 * abs(refv) + 1e-12: why.append("loss %g vs %g" % (red[0], refv))
        if F and float((tl["scale"].double() - gref).abs().max()) > 1e-6 * float(gref.abs().max()) + 1e-12: why.append("shares")
        wq, wqd = dm.fk_forward(jq.view(Ff * bsf, nq), jqd.view(Ff * bsf, nqd))
        if not (torch.equal(tl["fk_body_q"], wq.view(Ff, bsf, nb, 7).permute(1, 0, 2, 3)) and torch.equal(tl["fk_body_qd"], wqd.view(Ff, bsf, nb, 6).permute(1, 0, 2, 3))): why.append("fk fwd")
        gain = torch.full((1,), 0.37, device=dev)
        gr = dm.rollout_backward_traj_loss(bs, T, inp["dt"], *[t[k] for k in BWD], f2s, o[4], tl, gain, fk=(jq, jqd, aq, aqd))
        wgq, wgqd = dm.fk_backward(jq.view(Ff * bsf, nq), jqd.view(Ff * bsf, nqd), aq.permute(1, 0, 2, 3).contiguous(), aqd.permute(1, 0, 2, 3).contiguous())
        if not (torch.equal(gr["fk_joint_q"].view(Ff * bsf, nq), wgq) and torch.equal(gr["fk_joint_qd"].view(Ff * bsf, nqd), wgqd)): why.append("fk bwd")
        k_ = (tl["scale"].t() * (0.37 / nb))[:, :, None, None]   # a zero share is an assignment (nothing flows, not 0 x NaN: pd_trajloss.h); a NaN
        sp_ = tl["seed_pos"].view(F, bs, nb, 7)                   # target's own seed IS NaN (0 x NaN, as autograd in the reference) and goes in
        seeds = torch.where(k_ != 0, sp_ * k_, torch.zeros_like(sp_)).reshape(F, bs * nb, 7).contiguous()
        g2 = dm.rollout_backward(bs, T, inp["dt"], *[t[k] for k in BWD], f2s, o[4], seeds, torch.zeros(F, bs * nb, 6, device=dev))
        for k in g2:
            if g2[k].numel() == 0: continue
            if not bool(torch.isfinite(gr[k]).all()): why.append("nonfinite " + k)
            elif float((gr[k] - g2[k]).abs().max()) > 2e-5 * float(g2[k].abs().max()) + 1e-30: why.append("grad " + k)
    except Exception as e:
        why.append("EXC %r" % (e,))
    bad += bool(why)
    print("%s %-8s fam=%d bs=%-3d T=%-2d frames=%s fk=%dx%d  loss %.3e clipped %d  %s" % ("BAD" if why else "ok ", name, fam, bs, T, f2s, Ff, bsf, red[0] if not why or "EXC" not in why[0] else float("nan"), int(red[3]) if not why or "EXC" not in why[0] else -1, "; ".join(why)), flush=True)
print("f4 stress: %d cases, %d failures" % (ncases, bad))
sys.exit(1 if bad else 0)
