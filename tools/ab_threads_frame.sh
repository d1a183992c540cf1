#!/bin/bash
# The frame chain (tools/probe_frame.py) against the number of copy threads, interleaved in one session.
ROOT=$GRAFT_REPO_ROOT
for round in 1 2 3; do for t in 3 4 6; do
  echo -n "threads $t: "; VGICP_UPLOAD_THREADS=$t python3 $ROOT/tools/probe_frame.py 30 60000 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:round(v,4) for k,v in d.items() if 'ms_per_frame' in k and 'host_auth' not in k})"
done; done
