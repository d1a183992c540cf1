#!/bin/bash
# Times the neighbour-search kernel of builds made with tools/ab_build.sh against each other in ONE session on ONE
# box (rocprofv3 kernel statistics of tools/probe_preprocess.py, two passes).
# usage: ab_prep.sh name1 name2 ...   ("tree" = the in-tree build)
ROOT=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
for name in "$@"; do
  if [ "$name" = tree ]; then unset VGICP_LIB_PATH; else export VGICP_LIB_PATH=$ROOT/eskf_lio_amd/lib_ab/$name/libvgicp_hip.so; fi
  rm -rf /tmp/abp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp -o p -- python3 $ROOT/tools/probe_preprocess.py 100000 0.3 > /dev/null 2>&1
  python3 - "$name" <<'PY'
import csv, glob, sys
for f in glob.glob("/tmp/abp/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    calls = max(int(r["Calls"]) for r in rows if "knn_search" in r["Name"])
    total = sum(float(r["TotalDurationNs"]) for r in rows) / calls / 1e3
    for r in rows:
        if "knn_search" in r["Name"]:
            print(f"{sys.argv[1]}: knn_search {float(r['AverageNs']) / 1e3:.1f} us (min {float(r['MinNs']) / 1e3:.1f}) over {r['Calls']} calls; "
                  f"all kernels of one preparation {total:.1f} us", flush=True)
PY
done; done
