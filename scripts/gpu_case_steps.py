#!/usr/bin/env python3
"""Per-step divergence of one env of a saved stress case: GPU vs C oracle fp32 vs C oracle fp64, frames at EVERY step (forward only).
Usage: gpu_case_steps.py file.npz env"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))
import numpy as np, torch
from diffphys_amd import dp_model, hip_backend, robots
from oracle import ref_c
from oracle.ref_c import RefC
z = np.load(sys.argv[1], allow_pickle=True); e = int(sys.argv[2])
name = str(z["name"]); tpl = robots.load_template(name)
inp = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
T = int(inp["nsteps"]); dt = float(inp["dt"]); f2s = list(range(T))
inp["nsteps"], inp["dt"], inp["frame2step"] = T, dt, f2s
dev = torch.device("cuda:0")
dm = hip_backend.DeviceModel(tpl)
if int(z["segw"]):
    dm.set_segment_width(int(z["segw"]))
FWD = ("q_init", "qd_init", "torques", "res_f", "refs", "target_ke", "target_kd", "body_inv_mass", "body_inertia", "body_inv_inertia")
t = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=np.float32)).to(dev) for k in FWD}
bs = inp["q_init"].size // dm.nq; nb = dm.nb
fos = list(f2s)
pos, vel, grf, jaf, ws = dm.rollout_forward(bs, T, dt, *[t[k] for k in FWD], frame2step=fos)
ref_c.build()
s32 = RefC(tpl, np.float32).rollout_forward(inp, T, f2s, dt)
s64 = RefC(tpl, np.float64).rollout_forward(inp, T, f2s, dt)
gv = vel.cpu().numpy().reshape(T, bs, nb, 6)[:, e]; v32 = s32["wp_vel"].reshape(T, bs, nb, 6)[:, e]; v64 = s64["wp_vel"].reshape(T, bs, nb, 6)[:, e]
gg = grf.cpu().numpy().reshape(T, bs, nb, 6)[:, e]; g32 = s32["grf"].reshape(T, bs, nb, 6)[:, e]; g64 = s64["grf"].reshape(T, bs, nb, 6)[:, e]
print("step | vel err gpu-c32  gpu-c64  c32-c64 | grf err gpu-c32 c32-c64 | max|grf|")
for s in range(T):
    print("%3d  | %.2e %.2e %.2e | %.2e %.2e | %.1f" % (s, np.abs(gv[s] - v32[s]).max(), np.abs(gv[s] - v64[s]).max(), np.abs(v32[s] - v64[s]).max(),
                                                   np.abs(gg[s] - g32[s]).max(), np.abs(g32[s] - g64[s]).max(), np.abs(g64[s]).max()))
