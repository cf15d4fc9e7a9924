#!/bin/bash
# Run GPU steps one after another on the gpurun box; a step that fails an assertion does not stop the sequence, a step that
# is killed at its time limit does (no further GPU work after a hang).
# usage: bash tools/gpu_seq.sh "<seconds> <logname> <command...>" ...
mkdir -p gpurun_out
for spec in "$@"; do
  set -- $spec
  lim=$1; log=$2; shift 2
  echo "== $log: $*" | tee -a gpurun_out/seq.log
  timeout -k 10 "$lim" bash -c "$*" > "gpurun_out/$log" 2>&1
  rc=$?
  echo "   rc=$rc" | tee -a gpurun_out/seq.log
  tail -n 3 "gpurun_out/$log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ] || [ $rc -eq 139 ]; then echo "step killed at its limit or crashed (GPU fault): stopping" | tee -a gpurun_out/seq.log; exit 1; fi
done
exit 0
