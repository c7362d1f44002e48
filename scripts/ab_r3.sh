#!/bin/bash
# same-box comparison against the round-3 tree kept under _variants/r3tree (its own bench.py, package and library):
#   scripts/ab_r3.sh <rounds> [ENV=VAL,... for the current tree]
ROUNDS=$1; ENVS=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
run() { ( cd "$1"; shift; for kv in "$@"; do export "$kv"; done
  timeout -k 10 300 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-alt --no-full-batch 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['roofline']['conv_ms_per_step'],3))" ); }
for r in $(seq 1 $ROUNDS); do
  echo -n "r3   "; run "$ROOT/_variants/r3tree"
  echo -n "now  "; IFS=',' read -ra KV <<< "$ENVS"; run "$ROOT" "${KV[@]}"
done
