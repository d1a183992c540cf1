#!/bin/bash
# Per-kernel times of the hand-written sort alone (eskf_lio_amd/lib/sort_check time <n> <kind>), rocprofv3 kernel statistics.
# usage: ab_sort.sh "<binaries under eskf_lio_amd/lib>" n1 n2 ...
ROOT=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
BINS=$1; shift
for bin in $BINS; do for n in "$@"; do for kind in 0 1; do
  rm -rf /tmp/abs; echo -n "$bin: "; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abs -o p -- $ROOT/eskf_lio_amd/lib/$bin time $n $kind 2>/dev/null | grep "per sort"
  python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/abs/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sortk" in r["Name"]:
            print(f"   {r['Name'][:60]:60s} {float(r['AverageNs']) / 1e3:7.2f} us (min {float(r['MinNs']) / 1e3:.2f}, max {float(r['MaxNs']) / 1e3:.2f}) x {r['Calls']}")
PY
done; done; done
