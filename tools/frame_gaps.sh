#!/bin/bash
# Gaps between consecutive kernels of the frame chain (rocprofv3 kernel trace of tools/probe_frame.py): where the device idles
# between the launches of one frame.   usage (GPU box): bash tools/frame_gaps.sh [frames] [points]
ROOT=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fg; rocprofv3 --kernel-trace --output-format csv -d /tmp/fg -o f -- python3 $ROOT/tools/probe_frame.py ${1:-30} ${2:-60000} > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob("/tmp/fg/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("vgicp::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    return ("rocprim:" + n.split("wrapped_")[-1][:28]) if "rocprim" in n.lower() else n[:40]
gap = collections.defaultdict(list); dur = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if g < 200:   # within one frame
        gap[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append(g)
for r in rows:
    dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for (a, b), g in sorted(gap.items(), key=lambda kv: -sum(kv[1])):
    if len(g) >= 20:
        m = sum(g) / len(g); tot += m * len(g)
        print(f"{m:7.2f} us x {len(g):4d}  {a} -> {b}")
print("sum of the listed gaps per frame: %.1f us" % (tot / max(1, len(dur.get("knn_search_kernel", [1])))))
PY
