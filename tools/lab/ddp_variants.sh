#!/bin/bash
# The 2-rank rehearsal under diagnostic toggles; stops at the first run that was killed (timeout / signal) rather than failed.
out=gpurun_out/${1:-ddpvar}; mkdir -p $out
run() { name=$1; shift
  env CAPE_REHEARSAL_VERBOSE=1 "$@" timeout -k 10 200 python tools/ddp_rehearsal.py > $out/$name.log 2>&1; rc=$?
  echo "$name rc=$rc"; if [ $rc -ge 124 ]; then echo "killed: stopping"; exit 1; fi; }
run default
run nosplitk CAPE_NO_SPLIT_K=1
run noside CAPE_SIDE_STREAM=0
run nopacked CAPE_GEMM_PACKED=0
exit 0
