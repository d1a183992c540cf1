#!/bin/bash
# Times the frame chain (tools/probe_frame.py) of builds made with tools/ab_build.sh against each other in ONE session.
# usage: ab_frame.sh name1 name2 ...   ("tree" = the in-tree build)
ROOT=$GRAFT_REPO_ROOT
for round in 1 2 3; do
for name in "$@"; do
  if [ "$name" = tree ]; then unset VGICP_LIB_PATH; else export VGICP_LIB_PATH=$ROOT/eskf_lio_amd/lib_ab/$name/libvgicp_hip.so; fi
  python3 $ROOT/tools/probe_frame.py 30 60000 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name: %.4f ms per frame, drop-in %.4f, stages %s' % (d['ms_per_frame'], d.get('dropin_ms_per_frame', -1), {k: round(v,1) for k,v in d['stage_us'].items() if isinstance(v,float)}))"
done; done
