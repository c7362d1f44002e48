#!/bin/bash
# rocprofv3 evidence for the NCSN++ path (config 5): kernel trace + stats of `scripts/bench_configs.py 5`, then two PMC passes
# (FETCH_SIZE, WRITE_SIZE: separate runs, kernel-trace only, as MI355X_MICROARCH.md prescribes) reduced to the upfirdn2d
# kernels.  Raw output goes to gpurun_out/ (merged back by gpurun); scripts/summarize_cfg5.py <tag> then writes the profiles/ files.
# Usage: scripts/profile_cfg5.sh r02
set -e
TAG=${1:-r02}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd "$ROOT"
OUT=$ROOT/gpurun_out/prof_${TAG}_cfg5
rm -rf "$OUT"; mkdir -p "$OUT" "$ROOT/profiles"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o cfg5 -- python3 scripts/bench_configs.py 5 > "$OUT/cfg5.json" 2> "$OUT/cfg5.err" || { tail -20 "$OUT/cfg5.err"; exit 1; }
STATS=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
cut -c1-160 "$STATS" | head -16
for C in FETCH_SIZE WRITE_SIZE; do
  P=$ROOT/gpurun_out/pmc_${TAG}_cfg5_$C
  rm -rf "$P"; mkdir -p "$P"
  IPDM_CFG5_FORWARD_ONLY=1 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$P" -o pmc -- python3 scripts/bench_configs.py 5 > "$P/out.json" 2> "$P/err.txt" || { tail -20 "$P/err.txt"; exit 1; }
done
echo "raw output under gpurun_out/; run scripts/summarize_cfg5.py $TAG in the build container to write profiles/"
