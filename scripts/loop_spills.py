#!/usr/bin/env python3
"""Where a kernel's scratch traffic sits: per basic block of a -save-temps device assembly, the MFMA count and the scratch loads /
stores -- spills inside the block with the MFMAs are paid every chunk, spills elsewhere once per tile or launch.
   scripts/loop_spills.py file.s <substring of the mangled kernel name> [...]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
for tag in sys.argv[2:]:
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and tag in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur, name = [], [], "entry"
    for l in lines[start:end]:
        t = l.strip()
        m = re.match(r"(\.LBB\d+_\d+):", t)
        if m:
            blocks.append((name, cur))
            cur, name = [], m.group(1)
        elif t and not t.startswith((";", ".")):
            cur.append(t)
    blocks.append((name, cur))
    print(tag)
    for name, b in blocks:
        mf = sum(x.startswith("v_mfma") for x in b)
        sl = sum(x.startswith("scratch_load") for x in b)
        ss = sum(x.startswith("scratch_store") for x in b)
        rl = sum(x.startswith(("v_readlane", "v_writelane")) for x in b)
        if mf or sl or ss:
            print(f"   {name:12s} insts {len(b):5d} mfma {mf:3d} scratch_load {sl:3d} scratch_store {ss:3d} lane-moves {rl:3d}")
