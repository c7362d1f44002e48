#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command; writes raw output under gpurun_out/prof_<tag>/ and a
# summary (top kernels by total time) to profiles/<tag>_kernel_stats.csv.  Usage: scripts/profile_bench.sh r01
set -e
TAG=${1:-r01}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT" "$ROOT/profiles"
cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o "$TAG" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-alt --no-full-batch > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
STATS=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
echo "stats file: $STATS"
head -40 "$STATS" > "$OUT/${TAG}_kernel_stats_top.csv"

cat "$OUT/${TAG}_kernel_stats_top.csv" | cut -c1-200 | head -30
