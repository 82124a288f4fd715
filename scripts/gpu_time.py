#!/usr/bin/env python3
"""Per-kernel device timing of rollout fwd / adjoint for a list of (robot, bs, segw) configs (HIP events around the last
launches of back-to-back batches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))
import numpy as np, torch
from diffphys_amd import robots, synth, hip_backend, dp_model

cfgs = [("laikago", 4096, 16), ("laikago", 4096, 32), ("laikago", 4096, 64), ("laikago", 512, 16), ("laikago", 16384, 16),
        ("human", 1024, 32), ("quad", 8192, 32)]
if len(sys.argv) > 1:
    cfgs = [(a.split(":")[0], int(a.split(":")[1]), int(a.split(":")[2])) for a in sys.argv[1:]]
dev = torch.device("cuda:0")
for name, bs, segw in cfgs:
    tpl = robots.load_template(name)
    T = 100
    inp = synth.make_inputs(tpl, name, bs=bs, nsteps=T, seed=0, seqs=("mi-trot", "mi-spin"))
    dm = hip_backend.DeviceModel(tpl); dm.set_segment_width(segw); dm.set_timing(True)
    if "PD_FAMILY" in os.environ:  # 1 = lane per body always, 2 = quad-lane wherever eligible (default: by batch size)
        dm.set_kernel_family(int(os.environ["PD_FAMILY"]))
    t = {k: torch.from_numpy(inp[k]).to(dev) for k in synth.INPUT_NAMES}
    f2s = inp["frame2step"]; fos = list(f2s)
    fa = [t[k] for k in ("q_init","qd_init","torques","res_f","refs","target_ke","target_kd","body_inv_mass","body_inertia","body_inv_inertia")]
    ba = [t[k] for k in ("q_init","qd_init","torques","refs","target_ke","target_kd","body_inv_mass","body_inertia","body_inv_inertia")]
    ap = torch.from_numpy(inp["adj_pos"]).to(dev); av = torch.from_numpy(inp["adj_vel"]).to(dev)
    # batches of steps enqueued back to back (like bench.py's timed region: no idle gaps for the clocks to sag in), the durations
    # of the last forward / adjoint launch of each batch are read; the first batches warm the clocks up
    fs, bs_ = [], []
    bufs = dm.alloc_rollout(bs, T, len(fos), dev)
    for it in range(9):
        for _ in range(10):
            out = dm.rollout_forward(bs, T, inp["dt"], *fa, frame2step=fos, out=bufs)
            g = dm.rollout_backward(bs, T, inp["dt"], *ba, fos, out[4], ap, av, out=bufs)
        torch.cuda.synchronize()
        if it >= 4:
            fs.append(dm.last_kernel_ms(0)); bs_.append(dm.last_kernel_ms(1))
    f, b = np.median(fs), np.median(bs_)
    nb, nqd = int(tpl["nb"]), int(tpl["nqd"]); C = 2 * nqd + 6 * nb; B = 4 * (26 * nb + 3 * C)
    print("TIMING g%s f%s %-8s bs=%-6d segw=%-2d fwd %.3f ms bwd %.3f ms -> %.3e env-steps/s  (%.2f%% of 8 TB/s at %d B/env-step)" % (
        os.environ.get("PD_GROUPS", "a"), os.environ.get("PD_FAMILY", "a"), name, bs, segw, f, b, bs * T / ((f + b) * 1e-3), 100 * bs * T / ((f + b) * 1e-3) * B / 8e12, B), flush=True)
