#!/bin/bash
# One kernel of the frame chain (name pattern) against the number of copy threads (VGICP_UPLOAD_THREADS), rocprofv3 kernel
# statistics of tools/probe_frame.py, interleaved in one session.   usage: ab_threads_kernel.sh <pattern> t1 t2 ...
PAT=$1; shift
ROOT=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for round in 1 2; do for t in "$@"; do
  export VGICP_UPLOAD_THREADS=$t
  rm -rf /tmp/abk; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -o p -- python3 $ROOT/tools/probe_frame.py 30 60000 > /tmp/abk.out 2>&1
  python3 - "$t" "$PAT" <<'PY'
import csv, glob, sys
for f in glob.glob("/tmp/abk/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Name"]:
            print(f"threads {sys.argv[1]}: {sys.argv[2]} {float(r['AverageNs']) / 1e3:.2f} us (min {float(r['MinNs']) / 1e3:.2f}) over {r['Calls']} calls", flush=True)
PY
  grep -o "\"ms_per_frame\": [0-9.]*" /tmp/abk.out | head -1
done; done
