#!/usr/bin/env python3
"""Re-runs a case saved by gpu_stress.py (gpurun_out/stress_fail_N.npz) and prints per-frame errors of one env against the saved oracle.
Usage: gpu_case.py file.npz [env]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))
import numpy as np, torch
from diffphys_amd import dp_model, hip_backend, robots
z = np.load(sys.argv[1], allow_pickle=True)
name = str(z["name"]); tpl = robots.load_template(name)
inp = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
T = int(inp["nsteps"]); dt = float(inp["dt"]); f2s = [int(x) for x in inp["frame2step"]]
dev = torch.device("cuda:0")
dm = hip_backend.DeviceModel(tpl)
if int(z["segw"]):
    dm.set_segment_width(int(z["segw"]))
FWD = ("q_init", "qd_init", "torques", "res_f", "refs", "target_ke", "target_kd", "body_inv_mass", "body_inertia", "body_inv_inertia")
t = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=np.float32)).to(dev) for k in FWD}
bs = inp["q_init"].size // dm.nq
fos = list(f2s)
pos, vel, grf, jaf, ws = dm.rollout_forward(bs, T, dt, *[t[k] for k in FWD], frame2step=fos)
F, nb = len(f2s), dm.nb
e = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = grf.cpu().numpy().reshape(F, bs, nb, 6); r = z["ref_grf"].reshape(F, bs, nb, 6)
v = vel.cpu().numpy().reshape(F, bs, nb, 6); rv = z["ref_wp_vel"].reshape(F, bs, nb, 6)
print("lib:", os.environ.get("PPR_DIFFPHYS_LIB", "default"), " max vel err over envs:", float(np.abs(v - rv).max()), " env", int(np.abs(v - rv).max((0, 2, 3)).argmax()))
for f in range(F):
    print("frame %2d step %2d  vel err %.2e  grf |f| per body gpu-ref: %s" % (f, f2s[f], np.abs(v[f, e] - rv[f, e]).max(),
          np.round(np.abs(g[f, e, :, 3:]).sum(1) - np.abs(r[f, e, :, 3:]).sum(1), 2)[[3, 6, 9, 12]]))
