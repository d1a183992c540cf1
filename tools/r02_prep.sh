#!/bin/bash
set -u
OUT=gpurun_out/${1:-r02p}; mkdir -p $OUT
timeout 900 python3 -m pytest tests -m gpu -q -k "preprocess or frame or replay or smoke or chain" > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest.log"; grep -E "passed|failed|^FAILED|^E  " "$OUT/pytest.log" | head -30
echo "== tile"; VGICP_DEBUG_PREP=1 timeout 300 python3 tools/probe_preprocess.py 100000 0.3 2>&1 | grep -E "prep\]" | tail -4 | tee $OUT/probe_tile.log
echo "== wave"; VGICP_PREP_SEARCH=wave timeout 300 python3 tools/probe_preprocess.py 100000 0.3 2>&1 | grep -E "\[prep\] device" | tee $OUT/probe_wave.log
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prep_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o prep -- python3 $GRAFT_REPO_ROOT/tools/probe_preprocess.py 100000 0.3 > /dev/null 2>&1
find $GRAFT_REPO_ROOT/$OUT/prof -name "*kernel_stats.csv" | head -1 | xargs head -14 | cut -c1-160
