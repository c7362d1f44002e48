#!/bin/bash
# SQ counters of one conv shape (scripts/bench_conv.py BENCH_ONLY=<idx>) -> gpurun_out/pmc_conv_<tag>.txt
set -e
TAG=${1:-wino}; IDX=${2:-0}
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_conv_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
export BENCH_ONLY=$IDX BENCH_NO_MIOPEN=1 BENCH_BX3=1 BENCH_WINO=1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT" -o p1 -- python3 scripts/bench_conv.py > "$OUT/log1.txt" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d "$OUT" -o p2 -- python3 scripts/bench_conv.py > "$OUT/log2.txt" 2>&1 || true
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "conv_" not in n or "pack" in n or "weight" in n: continue
        agg[n[:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for n, d in agg.items():
    print(n)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} mean {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
