#!/usr/bin/env python3
"""Big-batch sanity: 150 001 Laikago envs x 40 steps (6.5 GB workspace) and 60 001 human envs x 20 steps -- copies of a 257-env batch inside
the big one must come out bit-identical to the small batch (poses and gradients), lane-per-body kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ppr-diffphys_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from diffphys_amd import hip_backend, robots, synth
dev = torch.device("cuda:0")
FWD = ("q_init", "qd_init", "torques", "res_f", "refs", "target_ke", "target_kd", "body_inv_mass", "body_inertia", "body_inv_inertia")
BWD = ("q_init", "qd_init", "torques", "refs", "target_ke", "target_kd", "body_inv_mass", "body_inertia", "body_inv_inertia")
for name, bs, T in (("laikago", 150001, 40), ("human", 60001, 20)):
    tpl = robots.load_template(name); nb = int(tpl["nb"])
    small = synth.make_inputs(tpl, name, bs=257, nsteps=T, seed=5, steps_per_frame=13, penetration=0.003)
    rep = (bs + 256) // 257
    def tile(k, v):
        v = np.asarray(v)
        if k in ("torques", "refs"): return np.tile(v.reshape(T, 257, -1), (1, rep, 1))[:, :bs].reshape(T, -1)
        if k == "res_f": return np.tile(v.reshape(T, 257, nb, 6), (1, rep, 1, 1))[:, :bs].reshape(T, bs * nb, 6)
        if k in ("adj_pos", "adj_vel"): F = v.shape[0]; return np.tile(v.reshape(F, 257, nb, -1), (1, rep, 1, 1))[:, :bs].reshape(F, bs * nb, -1)
        per = v.reshape(257, -1); return np.tile(per, (rep, 1))[:bs].reshape((-1,) + v.shape[1:]) if v.ndim > 1 else np.tile(per, (rep, 1))[:bs].reshape(-1)
    f2s = list(small["frame2step"])
    dm = hip_backend.DeviceModel(tpl); dm.set_kernel_family(1)
    def run(inp, n):
        t = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=np.float32)).to(dev) for k in FWD + ("adj_pos", "adj_vel")}
        o = dm.rollout_forward(n, T, small["dt"], *[t[k] for k in FWD], frame2step=f2s)
        g = dm.rollout_backward(n, T, small["dt"], *[t[k] for k in BWD], f2s, o[4], t["adj_pos"], t["adj_vel"])
        torch.cuda.synchronize()
        return o, g
    big = {k: tile(k, small[k]) for k in FWD + ("adj_pos", "adj_vel")}
    ob, gb = run(big, bs)
    os_, gs = run(small, 257)
    F = len(f2s)
    last = (rep - 1) * 257  # the last full copy starts here (if it fits)
    ok = True
    for lo in (0, 257 * (rep // 2)):
        a = ob[0].view(F, bs, nb, 7)[:, lo:lo + 257]; b = os_[0].view(F, 257, nb, 7)
        ok &= torch.equal(a, b)
        ok &= torch.equal(gb["q_init"].view(bs, -1)[lo:lo + 257], gs["q_init"].view(257, -1))
        ok &= torch.equal(gb["refs"].view(T, bs, -1)[:, lo:lo + 257], gs["refs"].view(T, 257, -1))
        ok &= torch.equal(gb["body_inertia"].view(bs, -1)[lo:lo + 257], gs["body_inertia"].view(257, -1))
    print("BIG %s bs=%d T=%d: workspace %.2f GB, copies bit-identical to the 257-env batch: %s, finite: %s" % (name, bs, T, ob[4].numel() * 4 / 2**30, ok, bool(torch.isfinite(gb["q_init"]).all())))
    del ob, gb, big
    torch.cuda.empty_cache()
