#!/usr/bin/env python3
"""cProfile of phys_model iterations (host side): which Python call sites cost the time.  Usage: gpu_iter_cprofile.py [num_envs]"""
import os, sys, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))
import importlib.util
import numpy as np, torch
from diffphys_amd.dataloader import DataLoader
from diffphys_amd.phys_model import phys_model

nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 256
spec = importlib.util.spec_from_file_location("pd_main", os.path.join(ROOT, "ppr-diffphys_amd", "main.py"))
pd_main = importlib.util.module_from_spec(spec); spec.loader.exec_module(pd_main)
opts = pd_main.get_opts(["--seqname", "mi-pace", "--urdf_template", "laikago", "--logroot", "/tmp/pprdp_prof/", "--logname", "p",
                         "--num_envs", str(nenv), "--frames_per_wdw", "4"])
torch.manual_seed(0); np.random.seed(0)
model = phys_model(opts, DataLoader(opts)).cuda(); model.train()
model.reinit_envs(nenv, frames_per_wdw=4)
fs = (torch.arange(nenv, device=model.device) * 3) % 40
def one(it):
    model.set_progress(it)
    out = model.forward(frame_start=fs)
    model.backward(out["total_loss"])
    model.update()
for it in range(5):
    one(it)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for it in range(10):
    one(5 + it)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue())
