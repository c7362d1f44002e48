#!/bin/bash
# same-box A/B on the headline iteration: scripts/ab_bench.sh <rounds> <arm> ...   where an arm is
#   prod                       the in-tree library, default environment
#   <tag>                      _variants/libipdm_<tag>.so (scripts/build_variant.sh)
#   NAME=VAL[,NAME=VAL...]     the in-tree library with these environment variables
# (alternating runs; prints ms_per_step and the conv census time of every run)
ROUNDS=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    (
      if [[ "$v" == *=* ]]; then IFS=',' read -ra KV <<< "$v"; for kv in "${KV[@]}"; do export "$kv"; done
      elif [ "$v" != "prod" ]; then export IPDM_LIB=$ROOT/_variants/libipdm_$v.so; fi
      timeout -k 10 300 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-alt --no-full-batch 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), round(d['roofline']['conv_ms_per_step'],3))"
    )
  done
done
