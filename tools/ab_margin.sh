#!/bin/bash
# Developer aid: the persistent launch's round at C2 across neighbour-prefetch margins (run-time switch) and builds
# (tools/ab_build.sh), in one session; then the in-kernel phase clocks.
# usage (repo root, GPU box): tools/ab_margin.sh "<builds>" "<margins>"     ("tree" = the in-tree build)
B=${1:-tree}; M=${2:-0.03}
for round in 1 2; do
for b in $B; do for m in $M; do
  if [ "$b" = tree ]; then unset VGICP_LIB_PATH; else export VGICP_LIB_PATH=$PWD/eskf_lio_amd/lib_ab/$b/libvgicp_hip.so; fi
  echo "== $b margin $m ($round)"; VGICP_PREFETCH_MARGIN=$m timeout 300 python3 tools/probe.py C2 100 2>&1 | grep -E "eager"
done; done; done
for b in $B; do for m in $M; do
  if [ "$b" = tree ]; then unset VGICP_LIB_PATH; else export VGICP_LIB_PATH=$PWD/eskf_lio_amd/lib_ab/$b/libvgicp_hip.so; fi
  echo "== $b margin $m (stamps)"; VGICP_DEBUG_STAMPS=1 VGICP_PREFETCH_MARGIN=$m timeout 300 python3 tools/probe.py C2 50 2>&1 | grep -E "stamps\] persistent" | cut -c1-330
done; done
