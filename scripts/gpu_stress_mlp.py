#!/usr/bin/env python3
"""Randomised sweep of the two host-side library kernels of the time-MLPs' backward: pd_linear_wgrad (weight + bias gradient on the fp32
matrix cores) and pd_colsum, on random shapes against float64, each twice for bit-wise repeatability.
    python scripts/gpu_stress_mlp.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))
import numpy as np
import torch
from diffphys_amd import hip_backend

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
bad = 0
worst_w = worst_b = worst_c = 0.0
for c in range(cases):
    n = int(rng.choice([1, 2, 3, 15, 16, 17, 119, 120, 121, int(rng.randint(1, 2000)), int(rng.randint(2000, 40000))]))
    m = int(rng.choice([128, 256, 384]))
    kin = int(rng.choice([128, 256, 384, 512, 640]))
    scale = float(10.0 ** rng.uniform(-3, 2))
    g = torch.from_numpy((rng.randn(n, m) * scale).astype(np.float32)).to(dev)
    x = torch.from_numpy((rng.randn(n, kin) + rng.uniform(-1, 1)).astype(np.float32)).to(dev)
    gw, gb = hip_backend.linear_wgrad(g, x)
    gw2, gb2 = hip_backend.linear_wgrad(g, x)
    ew = float(((gw.double() - g.double().t() @ x.double()).abs() / (g.abs().double().t() @ x.abs().double() + 1e-300)).max())
    eb = float(((gb.double() - g.double().sum(0)).abs() / (g.abs().double().sum(0) + 1e-300)).max())
    k2 = int(rng.randint(1, 700))
    y = torch.from_numpy(rng.randn(n, k2).astype(np.float32)).to(dev)
    cs, cs2 = hip_backend.colsum(y), hip_backend.colsum(y)
    ec = float(((cs.double() - y.double().sum(0)).abs() / (y.abs().double().sum(0) + 1e-300)).max())
    ok = ew < 1e-6 and eb < 1e-6 and ec < 2e-6 and torch.equal(gw, gw2) and torch.equal(gb, gb2) and torch.equal(cs, cs2)
    worst_w, worst_b, worst_c = max(worst_w, ew), max(worst_b, eb), max(worst_c, ec)
    if not ok:
        bad += 1
        print("FAIL n=%d m=%d kin=%d k2=%d  wgrad %.2e bias %.2e colsum %.2e" % (n, m, kin, k2, ew, eb, ec))
print("mlp stress: %d cases, %d failures; worst relative-to-sum|terms| error: wgrad %.2e, bias %.2e, colsum %.2e" % (cases, bad, worst_w, worst_b, worst_c))
sys.exit(1 if bad else 0)
