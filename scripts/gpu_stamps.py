#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares inside one sim step (needs `make -C ppr-diffphys_amd/csrc stamps`).
Read the SHARES, not the absolute time: the stamped build forbids overlaps the real kernel has."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["PPR_DIFFPHYS_LIB"] = os.path.join(ROOT, "ppr-diffphys_amd/diffphys_amd/lib/libpprdiffphys_hip_stamps.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))
import numpy as np, torch
from diffphys_amd import robots, synth, hip_backend, dp_model

name, bs, segw = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("laikago", 4096, 16)
dev = torch.device("cuda:0")
tpl = robots.load_template(name); T = 100
inp = synth.make_inputs(tpl, name, bs=bs, nsteps=T, seed=0, seqs=("mi-trot", "mi-spin"))
dm = hip_backend.DeviceModel(tpl); dm.set_segment_width(segw)
t = {k: torch.from_numpy(inp[k]).to(dev) for k in synth.INPUT_NAMES}
f2s = inp["frame2step"]; fos = dp_model.frame_of_step_tensor(T, f2s, dev)
fa = [t[k] for k in ("q_init","qd_init","torques","res_f","refs","target_ke","target_kd","body_inv_mass","body_inertia","body_inv_inertia")]
ba = [t[k] for k in ("q_init","qd_init","torques","refs","target_ke","target_kd","body_inv_mass","body_inertia","body_inv_inertia")]
ap = torch.from_numpy(inp["adj_pos"]).to(dev); av = torch.from_numpy(inp["adj_vel"]).to(dev)
nw = (bs * segw // 64 + 3) // 4 * 4 + 64
dbg = torch.zeros(nw * 16, dtype=torch.int64, device=dev)
L = hip_backend.lib()
L.pd_debug_set_buffer(ctypes.c_void_p(dbg.data_ptr()))
for it in range(2):
    out = dm.rollout_forward(bs, T, inp["dt"], *fa, frame_of_step=fos, nframes=len(f2s))
torch.cuda.synchronize()
f = dbg.view(-1, 16).cpu().numpy().astype(np.float64); f = f[f.sum(1) > 0]
dbg.zero_()
g = dm.rollout_backward(bs, T, inp["dt"], *ba, fos, len(f2s), out[4], ap, av)
torch.cuda.synchronize()
b = dbg.view(-1, 16).cpu().numpy().astype(np.float64); b = b[b.sum(1) > 0]
fn = ["top: controls+spill stores", "contact sweep + facc", "joints fwd + pcon write", "child gather + traj_f/frame stores", "integrate", "stage record"]
bn = ["top: seeds+unpack+prefetch+stage", "integrate adj + g_res_f", "joints adj + stores", "contact sweep adj", "gathers + tail"]
for lab, arr, names in (("FWD", f, fn), ("BWD", b, bn)):
    tot = arr.sum(1).mean()
    print("%s  waves=%d  cycles/step (100 MHz memtime ticks x?) mean total per wave per step = %.0f" % (lab, len(arr), tot / T))
    for i, n in enumerate(names):
        print("   %-40s %6.1f%%   %8.0f ticks/step" % (n, 100 * arr[:, i].mean() / tot, arr[:, i].mean() / T))
    for i, n in ((8, "sweep: L1 body cull"), (9, "sweep: L2 tile cull (body loop)"), (10, "sweep: L3 point cull"), (11, "sweep: hit pass")):
        print("      (inside) %-32s %6.1f%%   %8.0f ticks/step" % (n, 100 * arr[:, i].mean() / tot, arr[:, i].mean() / T))
