#!/usr/bin/env python3
"""Diagnostic: instructions of one loop of a kernel attributed to source lines (needs an ISA dump compiled with
-gline-tables-only).  usage: isa_lines.py dump.s <kernel-substring> <loop_start> <loop_end> (instruction indices as printed by
isa_mix.py on the SAME dump)"""
import re, sys, collections
path, pat, a, b = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
lines = open(path).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l)
cur, n, cnt = ("?", 0), 0, collections.Counter()
for l in lines[start:]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m: cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2))); continue
    t = l.split(";")[0].strip()
    if not t or t.startswith(".") or re.match(r"^(\.LBB\S+|_Z\S*):", t): continue
    if a <= n <= b: cnt[cur] += 1
    n += 1
    if t.startswith("s_endpgm"): break
src = {}
tot = sum(cnt.values())
print("total", tot)
for (f, ln), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:int(sys.argv[5]) if len(sys.argv) > 5 else 60]:
    if f not in src:
        try: src[f] = open("ppr-diffphys_amd/csrc/" + f).read().split("\n")
        except Exception: src[f] = []
    text = src[f][ln - 1].strip()[:110] if 0 < ln <= len(src[f]) else ""
    print("%4d %5.1f%%  %s:%d  %s" % (c, 100.0 * c / tot, f, ln, text))
