#!/usr/bin/env python3
"""The own-trajectory gradient check at the reference's REAL horizons: 10 envs x 760 steps (mi-pace window) and 10 x 1 910 (the 83-step
clips), 24 frames, both Laikago kernel families.  Prints the per-env worst relative error against the float64 adjoint of the kernel's own
trajectory, next to what a plain fp32 evaluation of the same adjoint loses."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ppr-diffphys_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from diffphys_amd import hip_backend, robots, synth
from helpers import own_trajectory_check
dev = torch.device("cuda:0")
tpl = robots.load_template("laikago")
for T, spf in ((760, 33), (1910, 83)):
    inp = synth.make_inputs(tpl, "laikago", bs=10, nsteps=T, seed=31, steps_per_frame=spf, seqs=("mi-pace",), penetration=0.002)
    for fam in (1, 2):
        dm = hip_backend.DeviceModel(tpl); dm.set_kernel_family(fam)
        own = own_trajectory_check(dm, tpl, inp, dev, hitlog_check=False, abs_floor=1e-8)
        w, f = np.sort(own["worst"]), np.sort(own["fp32_atan2"])
        print("LONG T=%4d family %d: worst-tensor error per env, sorted: %s | plain fp32 on the same trajectory: %s | one-ulp conditioning median %.1e" % (
            T, fam, " ".join("%.0e" % x for x in w), " ".join("%.0e" % x for x in f), np.median(own["cond"])), flush=True)
