#!/bin/bash
# rocprofv3 kernel trace of the resident frame chain (tools/replay.py --resident): GPU-busy time per frame against
# the stage timers the replay prints.   usage (GPU box): bash tools/frame_prof.sh [frames] [points]
ROOT=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
F=${1:-40}; P=${2:-60000}
rm -rf /tmp/fp; rocprofv3 --kernel-trace --output-format csv -d /tmp/fp -o f -- python3 $ROOT/tools/replay.py --synthetic $F --points $P --device-map --resident --out /tmp/fp/traj.tum 2>&1 | grep "elapsed time"
python3 - $F <<'PY'
import csv, glob, sys, collections
frames = int(sys.argv[1])
rows = []
for f in glob.glob("/tmp/fp/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
busy = collections.Counter(); calls = collections.Counter()
for r in rows:
    name = r["Kernel_Name"].replace("vgicp::(anonymous namespace)::", "").split("(")[0][:60]
    busy[name] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); calls[name] += 1
total = sum(busy.values())
print(f"GPU busy {total / frames / 1e3:.1f} us per frame over {len(rows) / frames:.1f} launches per frame")
for name, t in busy.most_common(12):
    print(f"  {t / frames / 1e3:7.1f} us/frame  {calls[name] / frames:5.1f} launches/frame  {name}")
PY
