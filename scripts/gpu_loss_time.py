#!/usr/bin/env python3
"""Device + launch time of se3_loss forward+backward: fused HIP kernel vs the torch composition."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))
import torch
from diffphys_amd import dp_utils
from oracle import pose_torch

dev = torch.device("cuda:0")
for dim in (7, 6):
    n = 4096 * 4 * 13  # bench batch: 4096 envs x 4 frames x 13 bodies
    g = torch.Generator().manual_seed(0)
    pred = torch.randn(n, dim, generator=g).to(dev).requires_grad_(True)
    gt = (pred.detach() + 0.3 * torch.randn(n, dim, generator=g).to(dev)).requires_grad_(True)
    for name, fn in (("torch composition", pose_torch.se3_loss), ("fused pd_se3_loss", dp_utils.se3_loss)):
        for it in range(3):
            fn(pred, gt, 0.1).mean().backward()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 50
        for it in range(K):
            pred.grad = None; gt.grad = None
            fn(pred, gt, 0.1).mean().backward()
        torch.cuda.synchronize()
        print("se3_loss dim=%d n=%d  %-18s %.3f ms per forward+backward" % (dim, n, name, (time.perf_counter() - t0) / K * 1e3), flush=True)
