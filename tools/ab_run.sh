#!/bin/bash
# Times builds made with tools/ab_build.sh against each other in ONE session on ONE box (probe.py, two passes).
# usage: ab_run.sh <outdir under gpurun_out> <C2|C5> <reps> name1 name2 ...   ("tree" = the in-tree build)
OUT=gpurun_out/$1; CFG=$2; REPS=$3; shift 3; mkdir -p $OUT
for round in 1 2; do
for name in "$@"; do
  if [ "$name" = tree ]; then unset VGICP_LIB_PATH; else export VGICP_LIB_PATH=$PWD/eskf_lio_amd/lib_ab/$name/libvgicp_hip.so; fi
  echo "== $name ($round)"; timeout 300 python3 tools/probe.py $CFG $REPS 2>&1 | grep -E "eager|error|Error" | tee -a $OUT/ab_$CFG.log
done; done
