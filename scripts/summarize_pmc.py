#!/usr/bin/env python3
"""Reduce the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_pmc.sh (raw rocprofv3 output under gpurun_out/) to
profiles/<tag>_conv_pmc.csv and profiles/conv_hbm_traffic.json.  Usage: python scripts/summarize_pmc.py r02"""
import csv, glob, json, os, sys, collections
tag = sys.argv[1]
root = os.getcwd()
res = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{root}/gpurun_out/pmc_{tag}_{c}/**/*counter_collection.csv", recursive=True)[0]
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name") or row.get("Kernel Name")
            if not any(k in name for k in ("conv_mfma_kernel", "conv_wino", "conv_bx3_kernel")) \
                    or row.get("Counter_Name") != c:
                continue
            res[name][c].append(float(row["Counter_Value"]))
rows, tot_f, tot_w, tot_n = [], 0.0, 0.0, 0
for name, d in sorted(res.items()):
    n = len(d["FETCH_SIZE"])
    f = sum(d["FETCH_SIZE"]) / max(n, 1)
    w = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
    rows.append((name, n, f, w))
    tot_f += sum(d["FETCH_SIZE"]); tot_w += sum(d["WRITE_SIZE"]); tot_n += n
os.makedirs("profiles", exist_ok=True)
with open(f"profiles/{tag}_conv_pmc.csv", "w") as fh:
    fh.write("kernel,launches,mean_FETCH_SIZE_KiB_raw,mean_WRITE_SIZE_KiB_raw\n")
    for r in rows:
        fh.write('"%s",%d,%.1f,%.1f\n' % r)
# FETCH_SIZE / WRITE_SIZE are in KiB; gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads -> x2
per_launch = (2 * tot_f + tot_w) * 1024 / max(tot_n, 1)
import subprocess, datetime
try:
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except Exception:
    commit = None
json.dump({"hbm_bytes_per_launch": per_launch, "launches_profiled": tot_n,
           "conv_impl": os.environ.get("IPDM_CONV_IMPL", "hx2"), "measured_at_commit": commit,
           "measured_on": datetime.date.today().isoformat(), "tag": tag,
           "fetch_KiB_raw_total": tot_f, "write_KiB_raw_total": tot_w,
           "note": "mean over all convolution-kernel launches (conv_bx3_kernel, conv_wino1d_kernel, conv_wino_bx3_kernel; conv_mfma_kernel / conv_wino_kernel with IPDM_CONV_IMPL=f32) of bench.py; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                   "(gfx950 FETCH_SIZE halving corrected per MI355X_MICROARCH.md)"},
          open("profiles/conv_hbm_traffic.json", "w"), indent=1)
print(open("profiles/conv_hbm_traffic.json").read())
