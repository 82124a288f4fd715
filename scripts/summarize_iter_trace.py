#!/usr/bin/env python3
"""profiles/<tag>_iteration_trace.md from a rocprofv3 --kernel-trace of scripts/gpu_iter_graph.py (gpurun_out/prof_<tag>/): the kernels of
ONE captured iteration of phys_model (the last one: between the last two rollout-forward launches), grouped.
    python scripts/summarize_iter_trace.py r05_iter"""
import collections
import csv
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05_iter"
f = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "prof_" + tag, "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_rollout_fwd" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
# the iteration starts with the MLPs, well before its rollout launch: cut at the largest idle gap in front of each rollout launch instead
def start_of(i):
    j = i
    while j > 0 and int(rows[j]["Start_Timestamp"]) - int(rows[j - 1]["End_Timestamp"]) < 150000 and i - j < 2000:
        j -= 1
    return j
s0, s1 = start_of(a), start_of(b)
seg = rows[s0:s1]
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])


def group(n):
    if any(k in n for k in ("k_rollout", "k_reduce_fk", "k_seeds_fk", "k_fk", "k_pose", "k_foot", "k_se3", "k_traj", "k_linear_wgrad", "k_colsum")):
        import re
        return "library: " + re.search(r"k_[a-z0-9_]+(<[^>]*>)?", n).group(0)[:60]
    if n.startswith("Cijk_"):
        return "GEMM (hipBLASLt / rocBLAS Tensile kernels)"
    if "multi_tensor_apply" in n or "FusedAdam" in n:
        return "torch multi-tensor (AdamW, foreach norms / scales)"
    if "reduce_kernel" in n:
        return "torch reductions"
    if "copyBuffer" in n or "fillBuffer" in n:
        return "runtime copies / fills"
    return "torch elementwise / cat / copy"


cnt, tot = collections.Counter(), collections.Counter()
for r in seg:
    g = group(r["Kernel_Name"])
    cnt[g] += 1
    tot[g] += dur(r)
span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
busy = sum(dur(r) for r in seg)
gaps = np.array([int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"]) for i in range(len(seg) - 1)])
out = ["# One iteration of `phys_model` on the reference's training window, captured (`%s`)" % tag, "",
       "`rocprofv3 --kernel-trace` of `scripts/gpu_iter_graph.py` (10 envs x 760 steps, 24 frames, mi-pace; forward + backward replayed as one HIP",
       "graph, update() eager), the last iteration of the run.  Times under the profiler (it stretches short kernels: unprofiled, the same",
       "iteration is what `scripts/gpu_iter_graph.py` prints).", "",
       "* kernels in the iteration: **%d**, span %.2f ms, kernel time %.2f ms, idle between kernels %.2f ms (median gap %.1f us)" % (
           len(seg), span / 1e6, busy / 1e6, np.clip(gaps, 0, None).sum() / 1e6, np.median(np.clip(gaps, 0, None)) / 1e3), "",
       "| group | kernels | time (us) | share |", "|---|---|---|---|"]
for g, t in tot.most_common():
    out.append("| %s | %d | %.1f | %.1f %% |" % (g, cnt[g], t / 1e3, 100.0 * t / busy))
path = os.path.join(ROOT, "profiles", tag + "_iteration_trace.md")
open(path, "w").write("\n".join(out) + "\n")
print("\n".join(out))
