#!/bin/bash
# Instruction mix of the registration kernel (counters only, separate passes): is a round bound by VALU issue or by memory?
# usage (through gpurun): bash tools/kernel_mix_pmc.sh <tag> [bench args, e.g. --config C5 --steps 10]
set -u
TAG=${1:-mix}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-c5 --no-frame-chain $*"
for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d "$OUT/$T" -o p -- python3 bench.py $ARGS > /dev/null 2> "$OUT/$T.err"
done
find "$OUT" -name "*counter_collection.csv" | while read f; do python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if "persistent_kernel" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in acc.items():
    print(k, "dispatches", n, "mean per dispatch %.0f" % (v / n))
PY
done
