#!/bin/bash
# Times ONE kernel (name pattern) of the 60 000-point frame chain across builds made with tools/ab_build.sh, in one
# session on one box (rocprofv3 kernel statistics of tools/probe_frame.py, two passes).
# usage: ab_kernel.sh <kernel-name-pattern> name1 name2 ...   ("tree" = the in-tree build)
PAT=$1; shift
ROOT=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
for name in "$@"; do
  if [ "$name" = tree ]; then unset VGICP_LIB_PATH; else export VGICP_LIB_PATH=$ROOT/eskf_lio_amd/lib_ab/$name/libvgicp_hip.so; fi
  rm -rf /tmp/abk; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -o p -- python3 $ROOT/tools/probe_frame.py 30 60000 > /tmp/abk.out 2>&1
  python3 - "$name" "$PAT" <<'PY'
import csv, glob, sys
for f in glob.glob("/tmp/abk/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Name"]:
            print(f"{sys.argv[1]}: {sys.argv[2]} {float(r['AverageNs']) / 1e3:.2f} us (min {float(r['MinNs']) / 1e3:.2f}) over {r['Calls']} calls", flush=True)
PY
  grep -o "\"ms_per_frame\": [0-9.]*" /tmp/abk.out | head -1
done; done
