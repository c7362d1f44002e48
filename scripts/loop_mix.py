#!/usr/bin/env python3
"""Instruction mix of the hottest basic block (most MFMAs) of a kernel in a -save-temps device assembly file:
   scripts/loop_mix.py file.s <substring of the mangled kernel name> [...]"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
for tag in sys.argv[2:]:
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and tag in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur = [], []
    for l in lines[start:end]:
        t = l.strip()
        if re.match(r"\.LBB\d+_\d+:", t):
            blocks.append(cur)
            cur = []
        elif t and not t.startswith((";", ".")):
            cur.append(t.split()[0])
    blocks.append(cur)
    best = max(blocks, key=lambda b: sum(op.startswith("v_mfma") for op in b))
    c = Counter()
    for op in best:
        if op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith("v_"): c["valu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith(("buffer_", "global_")): c["vmem"] += 1
        elif op.startswith("s_waitcnt"): c["wait"] += 1
        elif op.startswith("s_barrier"): c["barrier"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        else: c[op] += 1
    print(tag, dict(c), "total", len(best))
    print("   valu:", Counter(op for op in best if op.startswith("v_") and not op.startswith("v_mfma")).most_common(12))
    print("   lds :", Counter(op for op in best if op.startswith("ds_")).most_common(8))
