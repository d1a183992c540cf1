#!/bin/bash
# PMC instruction mix of the scan preparation's search kernel (counters only, separate passes).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prep_pmc
rm -rf $OUT; mkdir -p $OUT
for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $OUT/$T -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_preprocess.py 100000 0.3 > /dev/null 2>&1
done
find $OUT -name "*counter_collection.csv" | while read f; do python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if "knn_search" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in acc.items():
    print(k, "dispatches", n, "mean per dispatch", v / n)
PY
done
