#!/bin/bash
# run GPU steps one after another on the gpurun box: scripts/gpu_steps.sh "<seconds> <command...>" ...
# A failing step (tests red) does not stop the chain; a step that is KILLED (timeout / signal: exit >= 124) does -- no further
# GPU work is started behind a hung one.  Each step's output goes to gpurun_out/steps/<n>.log, its tail to stdout.
mkdir -p gpurun_out/steps
n=0
for s in "$@"; do
  n=$((n+1))
  secs=${s%% *}; cmd=${s#* }
  echo "== step $n (limit ${secs}s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > gpurun_out/steps/$n.log 2>&1
  rc=$?
  tail -n 15 gpurun_out/steps/$n.log
  echo "== step $n rc=$rc"
  if [ $rc -ge 124 ]; then echo "step $n was killed: stopping"; exit $rc; fi
done
exit 0
