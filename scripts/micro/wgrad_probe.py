#!/usr/bin/env python3
"""pd_linear_wgrad against the BLAS path it replaces, inside a captured graph (50 calls per replay): us per call.
    python scripts/micro/wgrad_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))
import torch
from diffphys_amd import hip_backend
from diffphys_amd.time_mlp import _gemm_long_k

for n, m, kin in ((7600, 256, 256), (7600, 256, 512), (1255, 256, 256), (25600, 256, 256)):
    g = torch.randn(n, m, device="cuda"); x = torch.randn(n, kin, device="cuda")
    for name, f in (("pd_linear_wgrad (gw + gb)", lambda: hip_backend.linear_wgrad(g, x)),
                    ("torch BLAS g^T x + pd_colsum", lambda: (_gemm_long_k(g.t(), x), hip_backend.colsum(g)))):
        gr = torch.cuda.CUDAGraph()
        s0 = torch.cuda.Stream(); s0.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s0):
            f(); s0.synchronize()
            with torch.cuda.graph(gr, stream=s0):
                for _ in range(50):
                    o = f()
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        gr.replay(); s.record(); gr.replay(); e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 50 * 1e3
        print("n=%-6d m=%d kin=%d  %-28s %6.1f us per call  (%.1f TFLOP/s)" % (n, m, kin, name, us, 2.0 * n * m * kin / us * 1e-6))
