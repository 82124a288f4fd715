#!/usr/bin/env python3
"""Diagnostic: distributions of tests/helpers.own_trajectory_check (kernel gradients vs the float64 adjoint of the kernel's own
trajectory with its decisions forced) for the BASELINE configs.  usage: gpu_own_traj.py [C2 C3 C4 C5 ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ppr-diffphys_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from helpers import GRAD_LEAD, own_trajectory_check
from diffphys_amd import hip_backend, robots, synth
from oracle import ref_c
ref_c.build()
dev = torch.device("cuda:0")

def cfg_inputs(cfg):
    if ":" in cfg:  # C4:16 = the C4 batch over a 16-step horizon, frames at states 0 and 16
        base, T = cfg.split(":"); T = int(T)
        name, bs, seqs, seed = {"C2": ("laikago", 256, ("mi-pace",), 4), "C3": ("human", 1024, ("mi-pace",), 12),
                                "C4": ("laikago", 4096, ("mi-trot", "mi-spin"), 9), "C5": ("quad", 8192, ("mi-pace",), 31)}[base]
        tpl = robots.load_template(name)
        inp = synth.make_inputs(tpl, name, bs=bs, nsteps=T, seed=seed, penetration=0.002, seqs=seqs)
        rng = np.random.RandomState(5)
        inp["qd_init"] = (rng.randn(*inp["qd_init"].shape) * 0.3).astype(np.float32)  # kicked: feet leave and hit the ground within the horizon
        inp["frame2step"] = [0, T]
        nb = int(tpl["nb"])
        inp["adj_pos"] = (rng.randn(2, bs * nb, 7) * 1e-3).astype(np.float32); inp["adj_vel"] = (rng.randn(2, bs * nb, 6) * 1e-3).astype(np.float32)
        return name, tpl, inp
    if cfg == "C5":
        tpl = robots.load_template("quad")
        inp = synth.make_inputs(tpl, "quad", bs=8192, nsteps=34, seed=31, steps_per_frame=33, penetration=0.004)
        rng = np.random.RandomState(2)
        inp["qd_init"] = (rng.randn(*inp["qd_init"].shape) * 0.1).astype(np.float32)
        inp["torques"] = (rng.randn(*inp["torques"].shape) * 0.5).astype(np.float32)
        return "quad", tpl, inp
    name, bs, seqs, seed = {"C2": ("laikago", 256, ("mi-pace",), 4), "C3": ("human", 1024, ("mi-pace",), 12),
                            "C4": ("laikago", 4096, ("mi-trot", "mi-spin"), 9)}[cfg]
    tpl = robots.load_template(name)
    return name, tpl, synth.make_inputs(tpl, name, bs=bs, nsteps=100, seed=seed, penetration=0.002, seqs=seqs)

for cfg in (sys.argv[1:] or ["C2", "C3", "C4", "C5"]):
    name, tpl, inp = cfg_inputs(cfg)
    dm = hip_backend.DeviceModel(tpl)
    if "PD_FAMILY" in os.environ:  # 1 = lane per body always, 2 = quad-lane wherever eligible
        dm.set_kernel_family(int(os.environ["PD_FAMILY"]))
    r = own_trajectory_check(dm, tpl, inp, dev)
    print("launch geometry fwd / bwd:", dm.last_launch_info(0), dm.last_launch_info(1))
    w = r["worst"]; bs = len(w)
    print("%s %s: %d envs; worst-tensor error per env: median %.2e p90 %.2e p99 %.2e p99.5 %.2e max %.2e" % (
        cfg, name, bs, np.median(w), np.percentile(w, 90), np.percentile(w, 99), np.percentile(w, 99.5), w.max()))
    for th in (1e-4, 2e-4, 1e-3, 1e-2):
        print("   envs above %.0e: %d" % (th, (w > th).sum()))
    for k in GRAD_LEAD:
        e = r["errs"][k]
        print("   %-18s median %.2e p99 %.2e max %.2e" % (k, np.median(e), np.percentile(e, 99), e.max()))
    print("   hit log: %d touches restated, %d missing from the log, %d log entries, %d overflowed env-steps" % (
        r["touches"], r["hitlog_missing"], r["log_entries"], r["hitlog_overflow"]))
    c = r["cond"]; ratio = w / np.maximum(c, 1e-7)
    print("   one-ulp conditioning of the fixed-trajectory adjoint: median %.2e p99 %.2e max %.2e; kernel error / conditioning: median %.2f p90 %.2f p99 %.2f max %.2f" % (
        np.median(c), np.percentile(c, 99), c.max(), np.median(ratio), np.percentile(ratio, 90), np.percentile(ratio, 99), ratio.max()))
    for tag in ("fp32_acos", "fp32_atan2"):
        e = r[tag]; rt = w / np.maximum(e, 1e-7)
        print("   fp32 C oracle (%s) on the same trajectory: median %.2e p90 %.2e p99 %.2e max %.2e, above 1e-3: %d; kernel / it: median %.2f p90 %.2f p99 %.2f max %.2f; envs with kernel > max(2e-4, 2x / 3x / 4x it): %d / %d / %d" % (
            tag, np.median(e), np.percentile(e, 90), np.percentile(e, 99), e.max(), (e > 1e-3).sum(), np.median(rt), np.percentile(rt, 90), np.percentile(rt, 99), rt.max(),
            (w > np.maximum(2e-4, 2 * e)).sum(), (w > np.maximum(2e-4, 3 * e)).sum(), (w > np.maximum(2e-4, 4 * e)).sum()))
    for K in (2, 4, 8):
        for fl in (1e-4, 2e-4):
            print("   envs with error > max(%.0e, %d x conditioning): %d" % (fl, K, (w > np.maximum(fl, K * c)).sum()))
    bar = 1e-3 if name == "laikago" else 2e-4
    bad = np.nonzero(w > bar)[0]
    print("   above the bar %.0e: %d envs; of these coulomb<1e-3: %d, force clamp<2e-2: %d, height<1e-6: %d, none: %d" % (
        bar, len(bad), (r["coulomb"][bad] < 1e-3).sum(), (r["force_clamp"][bad] < 2e-2).sum(), (r["height"][bad] < 1e-6).sum(),
        ((r["coulomb"][bad] >= 1e-3) & (r["force_clamp"][bad] >= 2e-2)).sum()))
    for i in bad[:12]:
        print("      env %d worst %.2e cond %.2e coulomb %.2e fclamp %.2e height %.2e vclamp %.2e  per tensor: %s" % (
            i, w[i], c[i], r["coulomb"][i], r["force_clamp"][i], r["height"][i], r["clamp_dist"][i],
            " ".join("%s=%.1e" % (k[:6], r["errs"][k][i]) for k in GRAD_LEAD)))
    tc = r["touch_counts"]
    print("   envs whose number of touching candidates changes within the horizon: %d; env-steps with a change: %d" % ((tc != tc[:1]).any(0).sum(), (tc[1:] != tc[:-1]).sum()))
    print("   all envs: coulomb<1e-3: %d, force clamp<2e-2: %d, height<1e-6: %d" % ((r["coulomb"] < 1e-3).sum(), (r["force_clamp"] < 2e-2).sum(), (r["height"] < 1e-6).sum()), flush=True)
