#!/usr/bin/env python3
"""Diagnostic behind tests/test_gpu_tight.py::test_config_size_gradients_every_tensor_per_env: per-env gradient error of the kernel
against the float64 oracle, the conditioning scale of each env and the first branch difference.  usage: gpu_explain.py C2|C3|C4"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import first_branch_difference, grad_env_errors, oracle_bundle, GRAD_LEAD
from test_gpu_parity import gpu_rollout
from test_gpu_tight import _config_inputs
from diffphys_amd import hip_backend

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
name, tpl, bs, inp = _config_inputs(cfg)
out = gpu_rollout(hip_backend.DeviceModel(tpl), inp, torch.device("cuda:0"), keep_traj=True)
ob = oracle_bundle(tpl, inp, bs)
e = grad_env_errors(out["grads"], ob["g64"], bs)
w = np.max(np.stack([e[k] for k in GRAD_LEAD]), 0)
cond = ob["cond"]
first = first_branch_difference(ob["rc64"], ob["st64"], out["traj"], inp, bs)
reg = first >= inp["nsteps"]
pc = lambda x, q: " ".join("%.1e" % v for v in np.percentile(x, q)) if len(x) else "-"
print("%s kernel vs float64, worst tensor per env 50/90/99/100: %s   conditioning scale: %s" % (cfg, pc(w, [50, 90, 99, 100]), pc(cond, [50, 90, 99, 100])))
print("   envs without a branch difference: %d of %d, error %s" % (reg.sum(), bs, pc(w[reg], [50, 90, 99, 100])))
print("   envs with one: error %s, first-difference step 10/50/90: %s" % (pc(w[~reg], [50, 90, 100]), " ".join("%d" % v for v in np.percentile(first[~reg], [10, 50, 90])) if (~reg).any() else "-"))
for K in (10, 30, 100):
    bad = w > np.maximum(K * cond, 1e-3)
    print("   e > max(%d cond, 1e-3): %d envs, %d of them without a branch difference" % (K, bad.sum(), (bad & reg).sum()))
for thr in (3e-5, 1e-4, 3e-4):
    c_ = reg & (ob["e_round"] < thr)
    print("   regular envs with e_round < %.0e: %d, kernel error max %.1e p99 %.1e" % (thr, c_.sum(), w[c_].max() if c_.any() else 0, np.percentile(w[c_], 99) if c_.any() else 0))
calm = reg & (ob["e_round"] < 3e-5)
print("   e_round 50/90/99: %s" % pc(ob["e_round"], [50, 90, 99]))
print("   calm envs (no branch difference, e_round < 3e-5): %d, error max %.1e" % (calm.sum(), w[calm].max() if calm.any() else 0))
