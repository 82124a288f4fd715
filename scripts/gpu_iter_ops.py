#!/usr/bin/env python3
"""How many aten ops one phys_model iteration dispatches, per section (forward pieces / backward / update), and the wall time
of each section with a synchronise after it (diagnostic for the host-plumbing rows f2-f4).  Usage: gpu_iter_ops.py [num_envs]"""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))
import importlib.util
import numpy as np, torch
from torch.utils._python_dispatch import TorchDispatchMode
from diffphys_amd.dataloader import DataLoader
from diffphys_amd import phys_model as pm

nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 256
spec = importlib.util.spec_from_file_location("pd_main", os.path.join(ROOT, "ppr-diffphys_amd", "main.py"))
pd_main = importlib.util.module_from_spec(spec); spec.loader.exec_module(pd_main)
opts = pd_main.get_opts(["--seqname", "mi-pace", "--urdf_template", "laikago", "--logroot", "/tmp/pprdp_prof/", "--logname", "p",
                         "--num_envs", str(nenv), "--frames_per_wdw", "4"])
torch.manual_seed(0); np.random.seed(0)
model = pm.phys_model(opts, DataLoader(opts)).cuda(); model.train()
model.reinit_envs(nenv, frames_per_wdw=4)
fs = (torch.arange(nenv, device=model.device) * 3) % 40


class Counter(TorchDispatchMode):
    def __init__(self):
        super().__init__(); self.section = "?"; self.n = collections.Counter(); self.ops = collections.defaultdict(collections.Counter)
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        self.n[self.section] += 1; self.ops[self.section][func.__name__] += 1
        return func(*args, **(kwargs or {}))


cnt = Counter()
wall = collections.Counter()


def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        prev = cnt.section; cnt.section = label
        torch.cuda.synchronize(); t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            torch.cuda.synchronize(); wall[label] += time.perf_counter() - t0; cnt.section = prev
    setattr(obj, name, g)


wrap(model, "get_mocap_tensors", "fwd: mocap query"); wrap(model, "get_net_pred", "fwd: time MLPs"); wrap(model, "fk_pos_vel", "fwd: FK of targets")
wrap(model, "get_foot_height", "fwd: foot height"); wrap(pm, "compose_delta", "fwd: compose_delta"); wrap(pm, "rotate_frame", "fwd: rotate_frame")
wrap(pm, "rotate_frame_vel", "fwd: rotate_frame_vel"); wrap(pm, "se3_loss", "fwd: se3_loss"); wrap(pm, "reduce_loss", "fwd: reduce_loss")
wrap(model, "forward", "fwd: rest"); wrap(model, "backward", "backward"); wrap(model, "update", "update")


def one(it):
    model.set_progress(it)
    out = model.forward(frame_start=fs)
    model.backward(out["total_loss"])
    model.update()


for it in range(4):
    one(it)
cnt.n.clear(); cnt.ops.clear(); wall.clear()
K = 5
with cnt:
    for it in range(K):
        one(4 + it)
tot_w = sum(wall.values()) - wall["fwd: rest"]  # "rest" wraps the others
print("%-26s %8s %10s" % ("section", "aten ops", "wall ms (synchronised, inclusive)"))
for k in sorted(cnt.n, key=lambda k: -cnt.n[k]):
    print("%-26s %8d %10.2f   top: %s" % (k, cnt.n[k] // K, wall[k] / K * 1e3, ", ".join("%s %d" % (a, b // K) for a, b in cnt.ops[k].most_common(6))))
print("total aten ops per iteration: %d" % (sum(cnt.n.values()) // K))
