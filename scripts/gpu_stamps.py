#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares inside one sim step (needs `make -C ppr-diffphys_amd/csrc stamps`).
Read the SHARES, not the absolute time: the stamped build forbids overlaps the real kernel has."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["PPR_DIFFPHYS_LIB"] = os.path.join(ROOT, "ppr-diffphys_amd/diffphys_amd/lib/libpprdiffphys_hip_stamps.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))
import numpy as np, torch
from diffphys_amd import robots, synth, hip_backend

name, bs, segw = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("laikago", 4096, 16)
dev = torch.device("cuda:0")
tpl = robots.load_template(name); T = 100
inp = synth.make_inputs(tpl, name, bs=bs, nsteps=T, seed=0, seqs=("mi-trot", "mi-spin"))
dm = hip_backend.DeviceModel(tpl); dm.set_segment_width(segw)
t = {k: torch.from_numpy(inp[k]).to(dev) for k in synth.INPUT_NAMES}
f2s = inp["frame2step"]
fa = [t[k] for k in ("q_init","qd_init","torques","res_f","refs","target_ke","target_kd","body_inv_mass","body_inertia","body_inv_inertia")]
ba = [t[k] for k in ("q_init","qd_init","torques","refs","target_ke","target_kd","body_inv_mass","body_inertia","body_inv_inertia")]
ap = torch.from_numpy(inp["adj_pos"]).to(dev); av = torch.from_numpy(inp["adj_vel"]).to(dev)
G = int(os.environ.get("PD_GROUPS", "4"))  # env groups per workgroup (the role-by-row mapping below follows it)
nblk = (bs * segw // 64 + G - 1) // G
dbg = torch.zeros((nblk * 12 + 64) * 16, dtype=torch.int64, device=dev)
L = hip_backend.lib()
L.pd_debug_set_buffer(ctypes.c_void_p(dbg.data_ptr()))
L.pd_debug_set_groups(G)
for it in range(2):
    out = dm.rollout_forward(bs, T, inp["dt"], *fa, frame2step=f2s)
torch.cuda.synchronize()
wf = dm.last_launch_info(0)["threads_per_wg"] // 64
f_all = dbg.view(-1, 16).cpu().numpy().astype(np.float64)[: nblk * wf].reshape(nblk, wf, 16)
dbg.zero_()
g = dm.rollout_backward(bs, T, inp["dt"], *ba, f2s, out[4], ap, av)
torch.cuda.synchronize()
wb = dm.last_launch_info(1)["threads_per_wg"] // 64
b_all = dbg.view(-1, 16).cpu().numpy().astype(np.float64)[: nblk * wb].reshape(nblk, wb, 16)


def rows(arr, role):  # role r = waves G r .. G r + G - 1 of a workgroup
    r = arr[:, G * role:G * role + G].reshape(-1, 16)
    return r[r[:, :13].sum(1) > 0]


def report(lab, who, r, names):
    tot = r[:, :13].sum(1).mean()
    print("%s %-14s waves=%d  cycles per step = %.0f" % (lab, who, len(r), tot / T))
    for i, n in names:
        print("      %-52s %6.1f%%   %8.0f cycles/step" % (n, 100 * r[:, i].mean() / tot, r[:, i].mean() / T))


fn = [(0, "signal A"), (1, "vmcnt wait, control unpack, next controls issued"), (2, "joints fwd + pcon write"), (6, "child gather"),
      (8, "trajectory / frame stores issued"), (9, "wait at barrier B"), (3, "facc read + wrench sums + force snapshots"), (4, "integrate"),
      (5, "stage record")]
cn = [(7, "wait at barrier A/A1 (idle)"), (8, "L1 body cull"), (9, "L2 tile cull"), (10, "L3 point cull"), (11, "hit pass"), (12, "tail")]
report("FWD", "body wave", rows(f_all, 0), fn)
if wf >= 2 * G:
    r = rows(f_all, 1)
    report("FWD", "contact wave", r, cn)
    print("FWD contact wave: exact cull redone in %.2f%% of wave-steps; speculated candidates (env 0 of the wave) %.2f per step" % (
        100 * r[:, 13].mean() / T, r[:, 14].mean() / T))
if wf >= 3 * G:  # forward with the cull wave (revolute-only robots): per-step averages of work done once per epoch of 4 steps
    r = rows(f_all, 2)
    report("FWD", "cull wave", r, [(7, "wait for the epoch's vectors (hand-over A of step e K)"), (8, "L1 body cull"), (9, "L2 tile cull"),
                                  (10, "L3 point cull -> candidate list, hand-over C")])
    print("FWD cull wave: culls dropped (tile list over its capacity) %.0f of %.0f; tiles after L2 (env 0 of the wave) %.1f per cull" % (
        r[:, 13].sum(), len(r) * max(1, (T - 2 + 3) // 4), r[:, 14].mean() / max(1, (T - 2 + 3) // 4)))
if segw == 64 and name == "laikago" and os.environ.get("PD_FAMILY", "") != "1":  # quad-lane adjoint (small batches): body, contact and state wave
    report("BWD", "body wave (quad)", rows(b_all, 0), [(0, "seeds added"), (1, "integrate adj phase 1 (reverse part) + adjf + signal A"),
                                                      (5, "g_res_f stores + integrate adj phase 2"),
                                                      (2, "LDS reads + rev_adjoint"), (7, "rotm adjoint + slots + control-gradient stores"),
                                                      (3, "child gather"), (9, "look-ahead: seeds / target requested, wait S, the state wave's values taken"),
                                                      (8, "wait B"), (4, "cacc")])
    report("BWD", "contact wave", rows(b_all, 1), [(7, "prefetch issue"), (9, "wait at hand-over A"),
                                                  (10, "contact adjoint per hit"), (11, "per-body sums"), (12, "tail / generic sweep")])
    print("BWD state wave: not stamped (it runs up to two steps ahead of the other two)")
elif wb == 3 * G:  # 3-role adjoint
    report("BWD", "integrate wave", rows(b_all, 0), [(0, "top: seeds + unpack + stage"), (1, "integrate adj (phase 1, signal A, phase 2) + g_res_f"),
                                                    (2, "wait J"), (3, "own + child gather"), (4, "wait C + cacc")])
    report("BWD", "contact wave", rows(b_all, 1), [(7, "prefetch issue"), (9, "wait A"), (10, "contact adjoint per hit"), (11, "per-body sums"),
                                                  (12, "tail / generic sweep")])
    report("BWD", "joint wave", rows(b_all, 2), [(8, "rev_forward (state-only half)"), (7, "prefetch issue"), (9, "wait A"),
                                                (10, "LDS reads + rev_adjoint + slots"), (11, "signal J + control-gradient stores")])
elif wb == 2 * G and name != "laikago":  # 2-role k_rollout_bwd3: integrate (+ contacts) wave, joint wave
    report("BWD", "integrate wave", rows(b_all, 0), [(0, "top: seeds + unpack + stage + signal S"), (1, "integrate adj (phase 1, signal A, phase 2) + g_res_f"),
                                                    (10, "inline contacts: replay setup"), (11, "inline contacts: hit pass"), (5, "inline contacts: rest"),
                                                    (2, "wait J"), (3, "own + child gather"), (4, "cacc")])
    report("BWD", "joint wave", rows(b_all, 1), [(7, "top: prefetch controls"), (8, "wait S + joint_adj_prep"), (9, "wait A"),
                                                (10, "joint_adj_apply + slots"), (11, "signal J + control-gradient stores")])
else:
    report("BWD", "body wave", rows(b_all, 0), [(0, "top: seeds + unpack + prefetch + stage"), (1, "integrate adj + g_res_f + adjf"),
                                               (2, "wait A + joints adj + stores"), (3, "child gather"), (4, "wait B + cacc gather")])
    if wb == 2 * G:
        report("BWD", "contact wave", rows(b_all, 1), [(8, "joint state-only half -> LDS"), (7, "prefetch issue"), (9, "wait at hand-over A"),
                                                      (10, "contact adjoint per hit"), (11, "per-body sums"), (12, "tail / generic sweep")])
