#!/bin/bash
# One A/B session on one box: tools/ab_session.sh <outdir> <config> <reps> name1 name2 ...  ("tree" = the in-tree build);
# every build twice in turn, then once with the in-kernel phase clocks (VGICP_DEBUG_STAMPS=1).
OUT=gpurun_out/$1; CFG=$2; REPS=$3; shift 3; mkdir -p $OUT
for round in 1 2; do
for name in "$@"; do
  if [ "$name" = tree ]; then unset VGICP_LIB_PATH; else export VGICP_LIB_PATH=$PWD/eskf_lio_amd/lib_ab/$name/libvgicp_hip.so; fi
  echo "== $name ($round)" | tee -a $OUT/ab_$CFG.log
  timeout 300 python3 tools/probe.py $CFG $REPS 2>&1 | grep -E "eager|rror" | tee -a $OUT/ab_$CFG.log
done; done
for name in "$@"; do
  if [ "$name" = tree ]; then unset VGICP_LIB_PATH; else export VGICP_LIB_PATH=$PWD/eskf_lio_amd/lib_ab/$name/libvgicp_hip.so; fi
  echo "== $name (stamps)" | tee -a $OUT/ab_$CFG.log
  VGICP_DEBUG_STAMPS=1 timeout 300 python3 tools/probe.py $CFG $((REPS / 2)) 2>&1 | grep -E "stamps\] (persistent|inside)" | tee -a $OUT/ab_$CFG.log
done
unset VGICP_LIB_PATH
