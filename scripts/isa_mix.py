#!/usr/bin/env python3
"""Diagnostic: static instruction mix of the big loops of one kernel in an ISA dump (scripts/dump_isa.sh).
usage: isa_mix.py /tmp/isa/k16.s <kernel-name-substring> [min_loop_instructions]"""
import re, sys, collections
path, pat = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 300
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
labels, ins = {}, []
for l in lines[start:end + 1]:
    t = l.split(";")[0].strip()
    if not t or t.startswith("."): 
        m = re.match(r"^(\.LBB\S+):", l.strip())
        if m: labels[m.group(1)] = len(ins)
        continue
    m = re.match(r"^(\.LBB\S+):", t)
    if m: labels[m.group(1)] = len(ins); continue
    if re.match(r"^_Z\S*:", t): continue
    ins.append(t)
def cls(op):
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos", "v_exp", "v_log")): return "valu_trans"
    if op.startswith(("v_mov", "v_accvgpr")): return "v_mov"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "v_lane"
    if op.startswith("v_cndmask"): return "v_cndmask"
    if op.startswith("v_cmp"): return "v_cmp"
    if op.startswith("v_"): return "valu_other"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch")): return "s_branch"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_"): return "salu"
    return "other"
loops = []
for i, t in enumerate(ins):
    m = re.match(r"^s_cbranch_\w+\s+(\.LBB\S+)|^s_branch\s+(\.LBB\S+)", t)
    if m:
        tgt = labels.get(m.group(1) or m.group(2))
        if tgt is not None and tgt < i and i - tgt >= minlen: loops.append((tgt, i))
print("function: %d instructions" % len(ins))
for a, b in sorted(set(loops)):
    c = collections.Counter(cls(t.split()[0]) for t in ins[a:b + 1])
    n = b - a + 1
    dpp = sum(1 for t in ins[a:b + 1] if "dpp" in t.split()[0] or "row_" in t or "wave_sh" in t or "quad_perm" in t)
    print("loop [%d, %d]: %d instructions  (dpp %d)" % (a, b, n, dpp))
    for k, v in c.most_common(): print("    %-12s %5d  %5.1f%%" % (k, v, 100.0 * v / n))
    ops = collections.Counter(t.split()[0] for t in ins[a:b + 1])
    print("    top ops:", ", ".join("%s %d" % kv for kv in ops.most_common(14)))
