#!/usr/bin/env python3
"""Diagnostic: prints instructions [a, b] (isa_mix.py's numbering) of one kernel of an ISA dump, with labels.
usage: isa_loop.py /tmp/isa/k16.s <kernel-name-substring> a b"""
import re, sys
path, pat, a, b = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
k = -1
for l in lines[start:end + 1]:
    t = l.split(";")[0].strip()
    m = re.match(r"^(\.LBB\S+):", t or l.strip())
    if m:
        if a <= k + 1 <= b: print(m.group(1) + ":")
        continue
    if not t or t.startswith(".") or re.match(r"^_Z\S*:", t): continue
    k += 1
    if a <= k <= b: print("%5d  %s" % (k, t))
