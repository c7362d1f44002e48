#!/bin/bash
# sample board power / shader clock (rocm-smi) while the headline iteration runs: scripts/power_probe.sh [bench args]
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
python bench.py --steps 400 --warmup 6 --no-cpu-baseline --no-alt --no-full-batch "$@" > gpurun_out/power_probe_bench.json 2>/dev/null &
BPID=$!
sleep 12
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "sclk|Power|mclk|Temperature \(Sensor (edge|junction|hotspot)" | tr -s ' ' | head -8
  echo ---
  sleep 1
done
wait $BPID
python -c "import json; d=json.load(open('gpurun_out/power_probe_bench.json')); print('ms_per_step', d['ms_per_step'])"
