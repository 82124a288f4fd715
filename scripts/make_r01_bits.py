#!/usr/bin/env python3
"""Records what the ROUND-1 library (build_r01/, rebuilt from the r01 commit's sources; scratch, git-ignored) computes on the
committed golden inputs and on a 8-env x 100-step bench-like batch: tests/golden/r01_bits_<robot>.npz hold the raw fp32
outputs and gradients.  tests/test_gpu_tight.py::test_against_round1_bits compares the current library with them, so a
kernel restructure shows whether it changed any bit (and by how many ulp).  Run on the GPU box; fixtures are data."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["PPR_DIFFPHYS_LIB"] = os.path.join(ROOT, "build_r01", "libpprdiffphys_r01.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
spec = importlib.util.spec_from_file_location("hip_backend_r01", os.path.join(ROOT, "build_r01", "hip_backend_r01.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
from helpers import INPUT_NAMES, golden_inputs, load_golden
from diffphys_amd import robots, synth

FWD = ("q_init", "qd_init", "torques", "res_f", "refs", "target_ke", "target_kd", "body_inv_mass", "body_inertia", "body_inv_inertia")
BWD = ("q_init", "qd_init", "torques", "refs", "target_ke", "target_kd", "body_inv_mass", "body_inertia", "body_inv_inertia")
dev = torch.device("cuda:0")
out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r01_bits")
os.makedirs(out_dir, exist_ok=True)


def run(dm, inp):
    bs = inp["q_init"].size // dm.nq
    T, f2s = inp["nsteps"], inp["frame2step"]
    t = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=np.float32)).to(dev) for k in INPUT_NAMES + ("adj_pos", "adj_vel")}
    fos = np.full(T + 1, -1, np.int32)
    for f, s in enumerate(f2s):
        fos[s] = f
    fos = torch.from_numpy(fos).to(dev)
    pos, vel, grf, jaf, ws = dm.rollout_forward(bs, T, inp["dt"], *[t[k] for k in FWD], frame_of_step=fos, nframes=len(f2s))
    g = dm.rollout_backward(bs, T, inp["dt"], *[t[k] for k in BWD], fos, len(f2s), ws, t["adj_pos"], t["adj_vel"])
    o = dict(wp_pos=pos, wp_vel=vel, grf=grf, jaf=jaf)
    o.update({"grad_" + k: v for k, v in g.items()})
    return {k: v.cpu().numpy() for k, v in o.items()}


for name in ("laikago", "human", "quad"):
    tpl = robots.load_template(name)
    dm = hb.DeviceModel(tpl)
    a = run(dm, golden_inputs(load_golden(name)))
    b = run(dm, synth.make_env_inputs(tpl, name, range(8), 100, seed=77, seqs=("mi-trot", "mi-spin"), penetration=0.002))
    np.savez_compressed(os.path.join(out_dir, "r01_bits_%s.npz" % name), **{"golden_" + k: v for k, v in a.items()},
                        **{"bench8_" + k: v for k, v in b.items()})
    print("recorded", name, {k: v.shape for k, v in a.items() if k.startswith("wp")})
