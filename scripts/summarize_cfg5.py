#!/usr/bin/env python3
"""Reduce the raw rocprofv3 output of scripts/profile_cfg5.sh (under gpurun_out/) to the summaries kept in profiles/:
  profiles/<tag>_cfg5_kernel_stats.csv   top of the --kernel-trace --stats table of `scripts/bench_configs.py 5`
  profiles/<tag>_cfg5.json               the timings that run printed
  profiles/<tag>_upfirdn2d_pmc.csv       per upfirdn2d call shape: launches, kernel time (kernel trace), FETCH_SIZE / WRITE_SIZE
                                         (separate PMC passes joined by dispatch id), corrected HBM bytes (2*FETCH + WRITE,
                                         MI355X_MICROARCH.md: gfx950 FETCH_SIZE reports half of a wide coalesced read)
Usage: python scripts/summarize_cfg5.py r02"""
import collections
import csv
import glob
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = f"{root}/gpurun_out/prof_{tag}_cfg5"
os.makedirs(f"{root}/profiles", exist_ok=True)
with open(f"{src}/cfg5_kernel_stats.csv") as f, open(f"{root}/profiles/{tag}_cfg5_kernel_stats.csv", "w") as g:
    for i, line in enumerate(f):
        if i < 40:
            g.write(line)
shutil.copy(f"{src}/cfg5.json", f"{root}/profiles/{tag}_cfg5.json")


def rows(counter):
    f = glob.glob(f"{root}/gpurun_out/pmc_{tag}_cfg5_{counter}/**/*counter_collection.csv", recursive=True)[0]
    out = {}
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "upfirdn2d" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                out[int(r["Dispatch_Id"])] = r
    return out


fetch, write = rows("FETCH_SIZE"), rows("WRITE_SIZE")
assert sorted(fetch) == sorted(write), "the two PMC passes dispatched different kernel sequences"
groups = collections.defaultdict(list)
for d, r in fetch.items():
    name = re.search(r"(upfirdn2d_\w+(<[^>]*>)?)", r["Kernel_Name"]).group(1)
    w = float(write[d]["Counter_Value"])
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    groups[(name, int(r["Grid_Size"]), round(w))].append((float(r["Counter_Value"]), w, dur))
with open(f"{root}/profiles/{tag}_upfirdn2d_pmc.csv", "w") as fh:
    fh.write("kernel,grid_size,launches,mean_FETCH_SIZE_KiB_raw,mean_WRITE_SIZE_KiB_raw,hbm_MB_per_launch_corrected,"
             "mean_us_under_pmc,GBps_under_pmc\n")
    for (name, grid, _), v in sorted(groups.items(), key=lambda kv: -kv[1][0][1]):
        n = len(v)
        f_, w_, t_ = (sum(x[i] for x in v) / n for i in range(3))
        mb = (2 * f_ + w_) * 1024 / 1e6
        fh.write('"%s",%d,%d,%.1f,%.1f,%.2f,%.1f,%.0f\n' % (name, grid, n, f_, w_, mb, t_, mb / t_ * 1e3))
print(open(f"{root}/profiles/{tag}_upfirdn2d_pmc.csv").read())
