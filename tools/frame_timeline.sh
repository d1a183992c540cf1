#!/bin/bash
# Device timeline of ONE frame of the chain (tools/probe_frame.py): start and end of every kernel and copy of a
# steady-state frame relative to the frame's first device activity (rocprofv3 kernel + memory-copy trace).
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ftl; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ftl -o t -- python3 $GRAFT_REPO_ROOT/tools/probe_frame.py 12 60000 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob("/tmp/ftl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].split("::")[-1][:40]))
for f in glob.glob("/tmp/ftl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
# frames: split at sweep_prologue_kernel launches; print the 8th frame of the first run
starts = [i for i, e in enumerate(ev) if "sweep_prologue" in e[2]]
k = starts[8]
# back up to the copies that precede the prologue of this frame
j = k
while j > 0 and ev[j - 1][2].startswith("COPY") and ev[k][0] - ev[j - 1][1] < 100000: j -= 1
t0 = ev[j][0]
end = starts[9] if len(starts) > 9 else len(ev)
for s, e, name in ev[j:end]:
    print(f"{(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f} us  ({(e - s) / 1e3:6.1f})  {name}")
PY
