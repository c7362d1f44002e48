#!/usr/bin/env python3
"""VGPR / scratch use of every kernel in a hipcc -save-temps device assembly file:  scripts/kernel_regs.py file.s"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
    name, b = m.group(1), m.group(2)
    vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", b).group(1)
    ac = re.search(r"\.amdhsa_accum_offset (\d+)", b)
    sp = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", b).group(1)
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    d = d.replace("ipdm_conv::(anonymous namespace)::", "").replace("(ipdm_conv::ConvArgs, int)", "").replace("(ipdm_conv::ConvArgs)", "")
    print(f"{d[:100]:100s} vgpr {vg} accum_off {ac.group(1) if ac else '-'} scratch {sp}")
