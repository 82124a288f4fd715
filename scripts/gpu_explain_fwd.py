#!/usr/bin/env python3
"""Diagnostic: a saved stress case (gpurun_out/stress_fail_N.npz) env by env -- forward error against the float64 oracle at every frame,
gradient error, measured conditioning, and the first step at which the kernel took a discrete branch (touch counts, sliding counts,
clamp masks) differently from float64.  An env is EXPLAINED if it is inside the bars, or has a branch difference, or its error is within
30 x its conditioning.  usage: gpu_explain_fwd.py case.npz [...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import first_branch_difference, grad_env_errors, oracle_bundle, GRAD_LEAD
from test_gpu_parity import gpu_rollout
from diffphys_amd import hip_backend, robots

for path in sys.argv[1:]:
    z = np.load(path, allow_pickle=True)
    name = str(z["name"]); tpl = robots.load_template(name)
    inp = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    inp["nsteps"] = int(inp["nsteps"]); inp["dt"] = float(inp["dt"]); inp["frame2step"] = [int(x) for x in inp["frame2step"]]
    T = inp["nsteps"]; bs = inp["q_init"].size // int(tpl["nq"]); F = len(inp["frame2step"]); nb = int(tpl["nb"])
    dm = hip_backend.DeviceModel(tpl)
    segw = int(z["segw"])
    if segw: dm.set_segment_width(segw)
    out = gpu_rollout(dm, inp, torch.device("cuda:0"), keep_traj=True)
    ob = oracle_bundle(tpl, inp, bs)
    st = ob["st64"]
    def per_env(a, r, w):
        a = np.asarray(a, np.float64).reshape(F, bs, nb, w); r = np.asarray(r, np.float64).reshape(F, bs, nb, w)
        return np.abs(a - r).max((0, 2, 3)) / (np.abs(r).max() + 1e-30)
    ep, ev, eg = per_env(out["wp_pos"], st["wp_pos"], 7), per_env(out["wp_vel"], st["wp_vel"], 6), per_env(out["grf"], st["grf"], 6)
    e = grad_env_errors(out["grads"], ob["g64"], bs)
    w = np.max(np.stack([e[k] for k in GRAD_LEAD]), 0)
    first = first_branch_difference(ob["rc64"], ob["st64"], out["traj"], inp, bs)
    off = (ep > 5e-5) | (ev > 2e-3) | (eg > 5e-3) | (w > 1e-3)
    unexpl = off & (first >= T) & (np.maximum(np.maximum(ep / 5e-5, ev / 2e-3), np.maximum(eg / 5e-3, w / 1e-3)) > 30 * np.maximum(ob["cond"], 1e-30) / 1e-3)
    print("%s: %s bs=%d T=%d kind=%s segw=%d: envs off the bars %d, with a branch difference %d, UNEXPLAINED %d" % (
        os.path.basename(path), name, bs, T, str(z["kind"]), segw, off.sum(), (off & (first < T)).sum(), unexpl.sum()))
    for i in np.argsort(-np.maximum(ev / 2e-3, w / 1e-3))[:4]:
        print("   env %3d pos %.1e vel %.1e grf %.1e grad %.1e cond %.1e first branch difference at step %d" % (i, ep[i], ev[i], eg[i], w[i], ob["cond"][i], first[i]))
