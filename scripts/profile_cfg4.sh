#!/bin/bash
# rocprofv3 kernel trace + stats of one temporal score evaluation census of config 4 (scripts/census_cfg4.py); the stats file is
# copied to profiles/<tag>_cfg4_kernel_stats.csv by hand afterwards.  Usage: scripts/profile_cfg4.sh r04
set -e
TAG=${1:-r04}
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_${TAG}_cfg4; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o cfg4 -- python3 scripts/census_cfg4.py > "$OUT/census.txt" 2> "$OUT/err.txt" || { tail -20 "$OUT/err.txt"; exit 1; }
STATS=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
cut -c1-150 "$STATS" | head -12
