#!/bin/bash
# rocprofv3 kernel statistics for the scan preparation (tools/probe_preprocess.py) on the GPU box.
cd /tmp && export TMPDIR=/tmp
N=${1:-100000}
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prep_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prep_prof -o prep -- python3 $GRAFT_REPO_ROOT/tools/probe_preprocess.py $N 0.3
find $GRAFT_REPO_ROOT/gpurun_out/prep_prof -name "*kernel_stats.csv" | head -1 | xargs head -15 | cut -c1-200
