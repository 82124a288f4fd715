#!/usr/bin/env python3
"""Diagnostic: the quad-lane (four lanes per body) kernels against the lane-per-body kernels on the same inputs: outputs, gradients,
hit logs.  usage: gpu_quad_check.py [bs] [T]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ppr-diffphys_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from helpers import relmax
from test_gpu_parity import gpu_rollout
from diffphys_amd import hip_backend, robots, synth

dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 37
T = int(sys.argv[2]) if len(sys.argv) > 2 else 34
tpl = robots.load_template("laikago")
inp = synth.make_inputs(tpl, "laikago", bs=bs, nsteps=T, seed=3, penetration=0.002, seqs=("mi-trot", "mi-spin"))
rng = np.random.RandomState(0)
inp["res_f"] = (rng.randn(*inp["res_f"].shape) * 0.3).astype(np.float32)
inp["torques"] = (rng.randn(*inp["torques"].shape) * 0.3).astype(np.float32)
inp["qd_init"] = (rng.randn(*inp["qd_init"].shape) * 0.2).astype(np.float32)
dm = hip_backend.DeviceModel(tpl)
print("family, eligible:", dm.kernel_family())
dm.set_kernel_family(1)
a = gpu_rollout(dm, inp, dev, keep_traj=True)
print("lane per body launch:", dm.last_launch_info(0))
dm.set_kernel_family(2)
b = gpu_rollout(dm, inp, dev, keep_traj=True)
print("quad-lane launch:    ", dm.last_launch_info(0))
for k in ("wp_pos", "wp_vel", "grf", "jaf"):
    print("%-8s relmax %.2e  finite %s" % (k, relmax(b[k], a[k]), np.isfinite(b[k]).all()))
for k in ("states_q", "states_qd", "states_f"):
    print("traj %-10s relmax %.2e" % (k, relmax(b["traj"][k], a["traj"][k])))
print("clamp masks equal:", np.array_equal(a["traj"]["clamp"], b["traj"]["clamp"]))
for k, v in a["grads"].items():
    print("grad %-18s relmax %.2e" % (k, relmax(b["grads"][k], v)))
