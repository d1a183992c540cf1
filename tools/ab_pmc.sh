#!/bin/bash
# VALU / SALU / LDS instruction counts of the search kernel across builds (tools/ab_build.sh), counters only.
# usage: ab_pmc.sh name1 name2 ...   ("tree" = the in-tree build)
ROOT=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for name in "$@"; do
  if [ "$name" = tree ]; then unset VGICP_LIB_PATH; else export VGICP_LIB_PATH=$ROOT/eskf_lio_amd/lib_ab/$name/libvgicp_hip.so; fi
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAVES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
    rm -rf /tmp/abpmc; rocprofv3 --pmc $C --output-format csv -d /tmp/abpmc -o p -- python3 $ROOT/tools/probe_frame.py 12 60000 > /dev/null 2>&1
    find /tmp/abpmc -name "*counter_collection.csv" | while read f; do python3 - "$f" "$name" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if "knn_search" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in acc.items():
    print(sys.argv[2], k, "dispatches", n, "mean per dispatch", round(v / n), flush=True)
PY
    done
  done
done
