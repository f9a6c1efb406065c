#!/bin/bash
# Run a list of GPU steps one after another on the gpurun box; each step under its own `timeout -k 10`, output to
# gpurun_out/<tag>_<n>.log.  A step that fails an assertion does not stop the list; a step that was KILLED (timeout /
# signal) does: nothing more is started on a GPU that may be wedged.
# usage: tools/gpu_steps.sh <tag> "<seconds> <command>" ["<seconds> <command>" ...]
set -u
tag=$1; shift
mkdir -p gpurun_out
n=0
for step in "$@"; do
  n=$((n+1))
  secs=${step%% *}; cmd=${step#* }
  log=gpurun_out/${tag}_${n}.log
  echo "[$tag $n] $cmd" | tee "$log"
  timeout -k 10 "$secs" bash -c "$cmd" >> "$log" 2>&1
  rc=$?
  echo "[$tag $n] rc=$rc" | tee -a "$log"
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "[$tag] step $n was killed: stopping"; exit $rc; fi
done
exit 0
