#!/usr/bin/env python3
"""main.main() end to end, twice -- forward() + backward() replayed as the captured HIP graph (the default) and eagerly (--no_graph) --
with the same random streams: the checkpoints after N rounds must agree BIT FOR BIT.  Longer and wider than the 20-iteration test of
the GPU suite: evaluation passes and checkpoints in between, the gradient guard's per-parameter verdicts in a fifth of the iterations,
the LR schedule over its whole cycle.
    python scripts/gpu_main_graph_vs_eager.py [seqname] [num_rounds]"""
import contextlib
import importlib.util
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))
import numpy as np
import torch

seq = sys.argv[1] if len(sys.argv) > 1 else "mi-pace"
rounds = sys.argv[2] if len(sys.argv) > 2 else "25"
spec = importlib.util.spec_from_file_location("pd_main", os.path.join(ROOT, "ppr-diffphys_amd", "main.py"))
pd_main = importlib.util.module_from_spec(spec); spec.loader.exec_module(pd_main)
out = {}
for mode in ("graph", "eager"):
    torch.manual_seed(0); np.random.seed(0)
    argv = ["--seqname", seq, "--urdf_template", "laikago", "--logroot", "/tmp/pprdp_gve/", "--logname", mode, "--num_rounds", rounds]
    if mode == "eager":
        argv.append("--no_graph")
    buf = io.StringIO()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(buf):
        pd_main.main(argv)
    torch.cuda.synchronize()
    log = buf.getvalue()
    # (the save directory's name is main.py's business: find the newest ckpt_phys_latest.pth under the log root of this mode)
    cands = []
    for d, _, fs in os.walk("/tmp/pprdp_gve"):
        if "ckpt_phys_latest.pth" in fs and mode in d:
            cands.append(os.path.join(d, "ckpt_phys_latest.pth"))
    ck = torch.load(sorted(cands, key=os.path.getmtime)[-1], map_location="cpu")
    iters = [l for l in log.splitlines() if l.startswith("[iter")]
    out[mode] = (ck, iters, log.count("large grad"), [l for l in log.splitlines() if l.startswith("[eval")])
    ts = np.array([float(l[l.rfind("(") + 1:l.rfind(" s)")]) for l in iters])
    print("%-5s: %d iterations in %.1f s (median iteration %.1f ms, mean of iterations 20.. %.2f ms), %d gradient-guard flags, last: %s | %s" % (
        mode, len(iters), time.perf_counter() - t0, np.median(ts) * 1e3, ts[20:].mean() * 1e3, out[mode][2], iters[-1][:70], out[mode][3][-1]), flush=True)
a, b = out["graph"][0], out["eager"][0]
same = sorted(a) == sorted(b) and all(torch.equal(a[k], b[k]) for k in a)
strip = lambda l: l[: l.rfind("(")]   # (the printed iteration time differs, of course)
print("checkpoints after %d iterations, graph vs eager: %s;  every printed loss line equal: %s;  evaluation lines equal: %s" % (
    len(out["graph"][1]), "BIT-IDENTICAL (%d tensors)" % len(a) if same else "DIFFERENT",
    [strip(l) for l in out["graph"][1]] == [strip(l) for l in out["eager"][1]], out["graph"][3] == out["eager"][3]))
sys.exit(0 if same else 1)
