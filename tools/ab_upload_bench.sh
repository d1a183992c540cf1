#!/bin/bash
# Developer aid: the headline loop of bench.py (fresh clouds freed inside the timed loop) across upload switches, one session.
# usage (repo root, GPU box): tools/ab_upload_bench.sh
for round in 1 2; do
for cfg in "0 2" "1 2" "1 3" "1 4" "0 3"; do
  set -- $cfg
  echo "== compact=$1 threads=$2 ($round)"
  VGICP_UPLOAD_COMPACT=$1 VGICP_UPLOAD_THREADS=$2 timeout 600 python3 bench.py --steps 300 --no-cpu-baseline --no-c5 --no-frame-chain 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); u=d['config']['upload']
        print('ms_per_step %.4f  upload_ms(host) %.4f  reused %.4f  p99 %.3f' % (d['ms_per_step'], u['upload_ms'], u['ms_per_step_reused'], u['step_ms_p99']))"
done; done
