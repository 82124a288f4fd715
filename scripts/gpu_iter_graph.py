#!/usr/bin/env python3
"""One optimisation iteration of phys_model on the reference's own training window (main.py:86: 10 envs x 760 steps, 24 frames):
eager forward() + backward() + update() against phys_model.iteration() replaying the captured HIP graph (capture_iteration) + update().
Prints ms per iteration of both, and whether the losses of 20 iterations and the parameters after them are bit-identical.
    python scripts/gpu_iter_graph.py [num_envs] [frames_per_wdw] [iterations]"""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))
import numpy as np
import torch

from diffphys_amd.dataloader import DataLoader
from diffphys_amd.phys_model import phys_model

nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 10
fpw = int(sys.argv[2]) if len(sys.argv) > 2 else 24
K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
spec = importlib.util.spec_from_file_location("pd_main", os.path.join(ROOT, "ppr-diffphys_amd", "main.py"))
pd_main = importlib.util.module_from_spec(spec); spec.loader.exec_module(pd_main)
opts = pd_main.get_opts(["--seqname", "mi-pace", "--urdf_template", "laikago", "--logroot", "/tmp/pprdp_prof/", "--logname", "g",
                         "--num_envs", str(nenv), "--frames_per_wdw", str(fpw)])
res = {}
for mode in ("eager-all-mlps", "eager", "graph"):
    torch.manual_seed(0); np.random.seed(0)
    model = phys_model(opts, DataLoader(opts)).cuda(); model.train()
    model.skip_zeroed_mlps = mode != "eager-all-mlps"   # False: torque_mlp / residual_f_mlp evaluated and multiplied by zero, as the reference does
    model.reinit_envs(nenv, frames_per_wdw=fpw)
    if mode == "graph":
        t0 = time.perf_counter()
        ok = model.capture_iteration(validate=True, verbose=True)
        torch.cuda.synchronize()
        print("capture_iteration: %s in %.2f s" % ("captured + validated" if ok else "REJECTED (eager fallback)", time.perf_counter() - t0), flush=True)
    np.random.seed(100)
    losses = []

    def one(it):
        model.set_progress(it)
        out = model.iteration()
        losses.append(out["total_loss"].detach().clone())
        model.update()

    for it in range(5):
        one(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(K):
        one(5 + it)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    res[mode] = (dt, torch.stack(losses).cpu().numpy(), {n: p.detach().cpu().numpy().copy() for n, p in model.named_parameters()})
    print("ITER %-14s %d envs x %d steps: %.2f ms per iteration (forward + backward + update)%s" % (
        mode, nenv, len(model.steps_idx), dt, "  [%d replays]" % model._graph["replays"] if getattr(model, "_graph", None) else ""), flush=True)
le, lg = res["eager-all-mlps"][1], res["eager"][1]
same = all(np.array_equal(res["eager-all-mlps"][2][n], res["eager"][2][n]) for n in res["eager"][2])
print("the two zeroed MLPs evaluated (as the reference) vs skipped: losses %s, parameters after %d iterations %s" % (
    "BIT-IDENTICAL" if np.array_equal(le, lg) else "max rel diff %.2e" % (np.abs(le - lg).max() / np.abs(le).max()), len(le), "BIT-IDENTICAL" if same else "different"))
le, lg = res["eager"][1], res["graph"][1]
print("losses of %d iterations, eager vs graph: %s  (first %.9e / %.9e, last %.9e / %.9e)" % (
    len(le), "BIT-IDENTICAL" if np.array_equal(le, lg) else "max rel diff %.2e" % (np.abs(le - lg).max() / np.abs(le).max()), le[0], lg[0], le[-1], lg[-1]))
same = all(np.array_equal(res["eager"][2][n], res["graph"][2][n]) for n in res["eager"][2])
print("parameters after %d iterations: %s" % (len(le), "BIT-IDENTICAL" if same else "different"))

# ---- where the captured iteration's time goes (the last model is the graph one)
if getattr(model, "_graph", None):
    import contextlib, io
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    R = 15   # (the schedule of this configuration has 101 steps: 25 + 2 x 15 fit)
    it0 = 30
    for block in range(2):
        rows, verdicts = [], []
        for it in range(R):
            model.set_progress(it0 + it)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev[0].record()
            out = model.iteration()
            ev[1].record()
            t1 = time.perf_counter()
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                model.update()
            verdicts.append("large grad" in buf.getvalue())
            ev[2].record()
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            rows.append([ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t0) * 1e3])
        it0 += R
        rows, verdicts = np.asarray(rows), np.asarray(verdicts)
        acc = np.median(rows[~verdicts], 0)   # (the all-clear iterations: a verdict costs its host path, the same either way)
        n_out = int(verdicts.sum())
        print("captured iteration: device time of iteration() %.2f ms, of update() %.2f ms; host time to enqueue iteration() %.2f ms, "
              "update() incl. its host transfer %.2f ms; wall %.2f ms  (medians over the all-clear iterations; %d of %d had another gradient-guard verdict)" % (
                  tuple(acc) + (n_out, R)), flush=True)
