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
f2s = inp["frame2step"]; fos = list(f2s)
fa = [t[k] for k in ("q_init","qd_init","torques","res_f","refs","target_ke","target_kd","body_inv_mass","body_inertia","body_inv_inertia")]
ba = [t[k] for k in ("q_init","qd_init","torques","refs","target_ke","target_kd","body_inv_mass","body_inertia","body_inv_inertia")]
ap = torch.from_numpy(inp["adj_pos"]).to(dev); av = torch.from_numpy(inp["adj_vel"]).to(dev)
nw = ((bs * segw // 64 + 3) // 4) * 8 + 64
dbg = torch.zeros(nw * 16, dtype=torch.int64, device=dev)
L = hip_backend.lib()
L.pd_debug_set_buffer(ctypes.c_void_p(dbg.data_ptr()))
for it in range(2):
    out = dm.rollout_forward(bs, T, inp["dt"], *fa, frame2step=fos)
torch.cuda.synchronize()
f_all = dbg.view(-1, 16).cpu().numpy().astype(np.float64)[: (nw // 8) * 8]
dbg.zero_()
g = dm.rollout_backward(bs, T, inp["dt"], *ba, fos, out[4], ap, av)
torch.cuda.synchronize()
b_all = dbg.view(-1, 16).cpu().numpy().astype(np.float64)[: (nw // 8) * 8]
def rows(arr, contact):
    r = arr.reshape(-1, 8, 16)[:, 4:] if contact else arr.reshape(-1, 8, 16)[:, :4]
    r = r.reshape(-1, 16)
    return r[r[:, :13].sum(1) > 0]

fn = [(0, "top: controls + spill stores"), (1, "wait at barrier A"), (2, "joints fwd + pcon write"), (6, "child gather"),
      (3, "wait at barrier B + facc + traj_f/frame stores"), (4, "integrate"), (5, "stage record")]
bn = [(0, "top: seeds + unpack + prefetch + stage"), (1, "integrate adj + g_res_f + adjf"), (2, "wait A + joints adj + stores"),
      (3, "child gather"), (4, "wait B + cacc gather")]
cn = [(7, "wait at barrier A/A1 (idle)"), (8, "L1 body cull"), (9, "L2 tile cull"), (10, "L3 point cull"), (11, "hit pass"), (12, "tail")]
cb = [(8, "joint state-only half -> LDS"), (7, "prefetch issue"), (9, "wait at barrier A"), (10, "contact adjoint per hit"),
      (11, "per-body sums"), (12, "tail / generic sweep")]
for lab, arr in (("FWD", f_all), ("BWD", b_all)):
    for who, names, contact in (("body wave", fn if lab == "FWD" else bn, False), ("contact wave", cn if lab == "FWD" else cb, True)):
        r = rows(arr, contact)
        tot = r[:, :13].sum(1).mean()
        print("%s %-12s waves=%d  cycles per step = %.0f" % (lab, who, len(r), tot / T))
        for i, n in names:
            print("      %-48s %6.1f%%   %8.0f cycles/step" % (n, 100 * r[:, i].mean() / tot, r[:, i].mean() / T))

r = rows(f_all, True)
print("FWD contact wave: exact cull redone in %.2f%% of wave-steps; speculated candidates (env 0 of the wave) %.2f per step" % (
    100 * r[:, 13].mean() / T, r[:, 14].mean() / T))
