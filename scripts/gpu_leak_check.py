#!/usr/bin/env python3
"""Device-memory leak check: 180 create / roll out (five frame tables each) / destroy cycles over the three robots; free device memory before and after."""
import os, sys, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ppr-diffphys_amd")]
import numpy as np, torch
from diffphys_amd import hip_backend, robots, synth
dev = torch.device("cuda:0")
FWD = ("q_init", "qd_init", "torques", "res_f", "refs", "target_ke", "target_kd", "body_inv_mass", "body_inertia", "body_inv_inertia")
BWD = ("q_init", "qd_init", "torques", "refs", "target_ke", "target_kd", "body_inv_mass", "body_inertia", "body_inv_inertia")
tpls = {n: robots.load_template(n) for n in ("laikago", "human", "quad")}
inps = {n: synth.make_inputs(tpls[n], n, bs=8, nsteps=10, seed=1) for n in tpls}
ts = {n: {k: torch.from_numpy(np.ascontiguousarray(inps[n][k], dtype=np.float32)).to(dev) for k in FWD + ("adj_pos", "adj_vel")} for n in tpls}
def cycle(n_frames_variants):
    for name in tpls:
        dm = hip_backend.DeviceModel(tpls[name])
        t = ts[name]
        for v in range(n_frames_variants):
            f2s = [0, 3 + v % 5, 10]
            o = dm.rollout_forward(8, 10, 5e-4, *[t[k] for k in FWD], frame2step=f2s)
            ap = torch.zeros(3, o[0].shape[1], 7, device=dev); av = torch.zeros(3, o[0].shape[1], 6, device=dev)
            dm.rollout_backward(8, 10, 5e-4, *[t[k] for k in BWD], f2s, o[4], ap, av)
        del dm
for _ in range(3): cycle(5)
gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
free0 = torch.cuda.mem_get_info()[0]
for _ in range(60): cycle(5)
gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
free1 = torch.cuda.mem_get_info()[0]
print("LEAK 180 model create / use / destroy cycles: device memory free before %.1f MB, after %.1f MB, difference %.2f MB" % (free0 / 2**20, free1 / 2**20, (free0 - free1) / 2**20))
