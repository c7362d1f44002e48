#!/bin/bash
# HBM traffic of the conv kernels from PMC counters (separate passes, as MI355X_MICROARCH.md prescribes):
#   pass 1: FETCH_SIZE   pass 2: WRITE_SIZE      (kernel-trace only, no other trace domains)
# Writes profiles/<tag>_conv_pmc.csv (per conv instantiation: launches, mean FETCH_SIZE, mean WRITE_SIZE) and
# profiles/conv_hbm_traffic.json (bytes per launch, gfx950 correction applied: FETCH_SIZE x 2).
set -e
TAG=${1:-r01}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd "$ROOT"
for C in FETCH_SIZE WRITE_SIZE; do
  OUT=$ROOT/gpurun_out/pmc_${TAG}_$C
  rm -rf "$OUT"; mkdir -p "$OUT"
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT" -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-full-batch --no-graph > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
  echo "pass $C done: $(find "$OUT" -name '*counter_collection.csv' | head -1)"
done
echo "raw passes under gpurun_out/pmc_${TAG}_*; run scripts/summarize_pmc.py $TAG where profiles/ is kept"
