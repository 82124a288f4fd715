#!/usr/bin/env python3
"""Records what the CURRENT library computes on the committed golden inputs and on the 8-env x 100-step bench-like batch:
tests/golden/<tag>_bits_<robot>.npz (raw fp32 outputs and gradients), the way the round-1 library's outputs were recorded
(r01_bits_*: from a build of the r01 commit's sources; that one-off script left the tree in round 5).  tests/test_gpu_tight.py::test_against_round1_bits freezes arithmetic against such fixtures: Laikago against r01;
human / quad against r02 (their forward pass was restructured in round 2 -- same terms without the products with exact
zeros of identity frames and basis vectors -- and differs from r01 by 1 ulp).  Run on the GPU box; fixtures are data.
usage: make_bits.py <tag> [robot ...]   ->  gpurun_out/<tag>_bits/"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import golden_inputs, load_golden
from test_gpu_parity import gpu_rollout
from diffphys_amd import hip_backend, robots, synth

tag = sys.argv[1]
names = sys.argv[2:] or ["human", "quad"]
dev = torch.device("cuda:0")
out_dir = os.path.join(ROOT, "gpurun_out", tag + "_bits")
os.makedirs(out_dir, exist_ok=True)
for name in names:
    tpl = robots.load_template(name)
    dm = hip_backend.DeviceModel(tpl)
    dm.set_kernel_family(1)   # the lane-per-body kernels, as test_against_frozen_bits pins them
    rec = {}
    for t, inp in (("golden", golden_inputs(load_golden(name))),
                   ("bench8", synth.make_env_inputs(tpl, name, range(8), 100, seed=77, seqs=("mi-trot", "mi-spin"), penetration=0.002))):
        out = gpu_rollout(dm, inp, dev)
        again = gpu_rollout(dm, inp, dev)
        for k in ("wp_pos", "wp_vel", "grf", "jaf"):
            assert np.array_equal(out[k], again[k])
            rec["%s_%s" % (t, k)] = out[k]
        for k, v in out["grads"].items():
            assert np.array_equal(v, again["grads"][k])
            rec["%s_grad_%s" % (t, k)] = v
    np.savez_compressed(os.path.join(out_dir, "%s_bits_%s.npz" % (tag, name)), **rec)
    print("recorded", name, sorted(rec)[:4], "...")
