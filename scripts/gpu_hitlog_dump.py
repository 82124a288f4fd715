#!/usr/bin/env python3
"""Diagnostic: forward rollout of the golden Laikago input (or a synthetic batch), dumps hit log + frame poses to an .npz (A/B of two
library builds via PPR_DIFFPHYS_LIB): gpu_hitlog_dump.py out.npz [bs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from diffphys_amd import robots, synth, hip_backend
from helpers import golden_inputs, load_golden

dev = torch.device("cuda:0")
tpl = robots.load_template("laikago")
if len(sys.argv) > 2:
    bs = int(sys.argv[2]); inp = synth.make_inputs(tpl, "laikago", bs=bs, nsteps=100, seed=0, seqs=("mi-trot", "mi-spin"))
else:
    inp = golden_inputs(load_golden("laikago"))
dm = hip_backend.DeviceModel(tpl)
bs = inp["q_init"].size // dm.nq; T = inp["nsteps"] if "nsteps" in inp else 100
t = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=np.float32)).to(dev) for k in synth.INPUT_NAMES}
fa = [t[k] for k in ("q_init","qd_init","torques","res_f","refs","target_ke","target_kd","body_inv_mass","body_inertia","body_inv_inertia")]
out = dm.rollout_forward(bs, T, inp["dt"], *fa, frame2step=list(inp["frame2step"]))
torch.cuda.synchronize()
nb = dm.nb
ws = out[4]
log = ws[T * 20 * bs * nb:].view(torch.int32).view(T, bs, -1).cpu().numpy()
traj = ws[:T * 20 * bs * nb].view(T, 5, bs * nb, 4).cpu().numpy()
np.savez(sys.argv[1], log=log, traj=traj, pos=out[0].cpu().numpy())
print("saved", sys.argv[1], "bs", bs, "T", T)
