#!/usr/bin/env python3
"""Where one optimisation iteration of phys_model (row f3) spends its time on the GPU: torch profiler, top ops by self time.
Usage: gpu_iter_profile.py [num_envs]"""
import os, sys, time
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
t0 = time.perf_counter()
K = 20
for it in range(K):
    one(5 + it)
torch.cuda.synchronize()
print("ITER num_envs=%d: %.2f ms per iteration (forward + backward + update)" % (nenv, (time.perf_counter() - t0) / K * 1e3), flush=True)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for it in range(5):
        one(30 + it)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=18, max_name_column_width=60))
ev = prof.key_averages()
print("kernel launches per iteration ~ %d" % (sum(e.count for e in ev if e.device_type.name == "CUDA" or e.self_device_time_total > 0) // 5))
