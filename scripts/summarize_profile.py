#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (rocprofv3 csv output of scripts/profile_gpu.sh) into committed summaries:
profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.csv, profiles/<tag>_summary.md and profiles/pmc_summary.json
(HBM bytes per launch with the gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md section HBM)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
write_json = "--json" in sys.argv  # only the headline (Laikago 4096 x 100) profile feeds profiles/pmc_summary.json, which bench.py reads
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

# gpurun MERGES a call's output into gpurun_out/: a tag profiled twice leaves two runs' files side by side -- take the newest of each
newest = lambda pattern: sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]
stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
rows = [r for r in csv.DictReader(open(stats))]
with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows[:6]:
        w.writerow([r["Name"][:80], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])

acc = collections.defaultdict(list)
meta = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in newest(os.path.join(d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "rollout" not in r["Kernel_Name"]:
                continue
            k = "k_rollout_bwd" if "bwd" in r["Kernel_Name"] else "k_rollout_fwd"
            acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            meta[k] = dict(grid=r["Grid_Size"], wg=r["Workgroup_Size"], lds=r["LDS_Block_Size"], vgpr=r["VGPR_Count"],
                           agpr=r["Accum_VGPR_Count"], sgpr=r["SGPR_Count"], scratch=r["Scratch_Size"], name=r["Kernel_Name"])
with open(os.path.join(dst, tag + "_pmc.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "launches", "mean_per_launch"])
    for (k, c), v in sorted(acc.items()):
        w.writerow([k, c, len(v), "%.6g" % (sum(v) / len(v))])

def code_object_registers(demangled):
    """VGPRs / spills of a kernel from the code-object metadata of THIS tree's sources (scripts/dump_isa.sh compiles them to ISA):
    rocprofv3's VGPR_Count column is in allocation granules of a wave32 (it printed 80 / 128 for kernels that have 159 / 252), which
    reads as if four waves fitted a SIMD.  demangled: e.g. 'void k_rollout_bwd<16, 1, true, false, false>(...)'."""
    import re
    import subprocess
    m = re.search(r"(k_rollout_\w+)<([^>]*)>", demangled)
    if not m:
        return None
    args = [a.strip() for a in m.group(2).split(",")]
    mangled = "".join(("Lb1E" if a == "true" else "Lb0E" if a == "false" else "Li%sE" % a) for a in args)
    segw = args[0]
    try:
        out = subprocess.run(["bash", os.path.join(ROOT, "scripts", "dump_isa.sh"), segw], capture_output=True, text=True, timeout=900).stdout
    except Exception:
        return None
    for line in out.splitlines():
        if (m.group(1) + "I" + mangled + "E") in line.replace(" ", ""):
            f = line.split()
            g = lambda key: f[f.index(key) + 1] if key in f else "?"
            return dict(vgpr=g("vgpr"), vgpr_spill=g("spill"), sgpr=g("sgpr"), sgpr_spill=g("sspill"))
    return None


mean = lambda k, c: sum(acc[(k, c)]) / max(1, len(acc[(k, c)]))
avg_ns = {("k_rollout_bwd" if "bwd" in r["Name"] else "k_rollout_fwd"): float(r["AverageNs"]) for r in rows if "rollout" in r["Name"]}
summary = {}
cmd_file = os.path.join(src, "command.txt")
cmd = open(cmd_file).read().strip().split("bench.py", 1)[-1].strip() if os.path.exists(cmd_file) else "--steps 10 --warmup 2"
N_SIMD = 1024  # 256 CUs x 4
# A SIMD issues ONE non-packed fp32 wave64 instruction per 4.15-4.45 cycles whatever the number of waves or their instruction-level
# parallelism (scripts/micro/valu_chain.hip built with -fno-slp-vectorize, scripts/micro/pk_issue.hip; rounds 1-2 used 2.5, which was the
# rate of the v_pk_fma_f32 the SLP vectoriser had made of that microbenchmark, per FMA)
VALU_CYC = 4.2
lines = ["# rocprofv3 summary `%s` (bench.py %s, 1x MI355X)\n" % (tag, cmd),
         "Source: `scripts/profile_gpu.sh %s` on the GPU box; raw csv under `gpurun_out/prof_%s/` (scratch)." % (tag, tag), ""]
for k in ("k_rollout_fwd", "k_rollout_bwd"):
    if k not in avg_ns:
        continue
    fetch_kb, write_kb = mean(k, "FETCH_SIZE"), mean(k, "WRITE_SIZE")
    hbm = fetch_kb * 1024 * 2 + write_kb * 1024  # gfx950: FETCH_SIZE counts 128-B requests at 64 B
    wc = mean(k, "SQ_WAVE_CYCLES")
    gui = mean(k, "GRBM_GUI_ACTIVE")
    gui = gui / 8.0  # the counter is summed over the 8 XCDs (each has its own GRBM): per-XCD busy cycles = the launch duration
    valu_rate = mean(k, "SQ_INSTS_VALU") / max(1.0, gui * N_SIMD)  # wave-instructions issued per SIMD and GPU-busy cycle
    waves_per_simd = float(meta.get(k, {}).get("grid", 0)) / 64.0 / N_SIMD
    summary[k] = dict(avg_launch_ns=avg_ns[k], fetch_size_kb_raw=fetch_kb, write_size_kb=write_kb, hbm_bytes_per_launch=hbm,
                      valu_issue_per_simd_cycle=valu_rate, valu_busy=VALU_CYC * valu_rate, valu_insts_per_launch=mean(k, "SQ_INSTS_VALU"),
                      wait_share=mean(k, "SQ_WAIT_ANY") / max(1.0, wc), waves_per_simd=waves_per_simd, meta=meta.get(k, {}))
    lines += ["## %s  (%s)" % (k, meta.get(k, {}).get("name", "")),
              "* average launch %.1f us; grid %s x wg %s; registers from the code object: %s (rocprofv3's columns, in wave32 allocation granules: VGPR %s, SGPR %s, scratch %s)" % (
                  avg_ns[k] / 1e3, meta[k]["grid"], meta[k]["wg"],
                  (lambda r: "%s VGPRs (%s spilled), %s SGPRs (%s spilled)" % (r["vgpr"], r["vgpr_spill"], r["sgpr"], r["sgpr_spill"]) if r else "n/a")(code_object_registers(meta[k]["name"])),
                  meta[k]["vgpr"], meta[k]["sgpr"], meta[k]["scratch"]),
              "* HBM: FETCH_SIZE %.0f KB raw (x2 gfx950 correction -> %.1f MB), WRITE_SIZE %.0f KB (%.1f MB) => %.1f MB per launch, %.0f GB/s" % (
                  fetch_kb, fetch_kb * 2048 / 1e6, write_kb, write_kb * 1024 / 1e6, hbm / 1e6, hbm / avg_ns[k]),
              "* SQ: waves %.0f, VALU insts %.3g, SALU %.3g, LDS %.3g, VMEM rd %.3g wr %.3g" % (
                  mean(k, "SQ_WAVES"), mean(k, "SQ_INSTS_VALU"), mean(k, "SQ_INSTS_SALU"), mean(k, "SQ_INSTS_LDS"),
                  mean(k, "SQ_INSTS_VMEM_RD"), mean(k, "SQ_INSTS_VMEM_WR")),
              "* wave cycles (quad-cycles) %.3g: waiting (SQ_WAIT_ANY) %.0f%%, issuing (SQ_ACTIVE_INST_ANY) %.0f%%, issue-stalled (SQ_WAIT_INST_ANY) %.0f%%" % (
                  wc, 100 * mean(k, "SQ_WAIT_ANY") / wc, 100 * mean(k, "SQ_ACTIVE_INST_ANY") / wc, 100 * mean(k, "SQ_WAIT_INST_ANY") / wc),
              "* secondary bound: %.2f waves per SIMD launched; VALU issue %.3f wave-instructions per SIMD-cycle (x 4.2 cycles each = %.0f%% of the SIMDs' fp32 issue rate); GPU busy %.3g cycles" % (
                  waves_per_simd, valu_rate, 100 * VALU_CYC * valu_rate, gui),
              "* LDS: bank-conflict cycles %.3g of %.3g active (%.0f%%)" % (
                  mean(k, "SQ_LDS_BANK_CONFLICT"), mean(k, "SQ_LDS_IDX_ACTIVE"),
                  100 * mean(k, "SQ_LDS_BANK_CONFLICT") / max(1.0, mean(k, "SQ_LDS_IDX_ACTIVE"))), ""]
open(os.path.join(dst, tag + "_summary.md"), "w").write("\n".join(lines))
summary["tag"] = tag
try:  # the library the counters were taken on (scripts/profile_gpu.sh): bench.py compares it with the library it runs
    summary["source_hash"] = open(os.path.join(src, "source_hash.txt")).read().strip()
except Exception:
    summary["source_hash"] = None
json.dump(summary, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1)
if write_json:
    json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
print("\n".join(lines))
