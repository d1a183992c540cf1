#!/bin/bash
set -u
OUT=gpurun_out/${1:-r02h}; mkdir -p $OUT
timeout 1500 python3 -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest.log"; grep -E "passed|failed|^FAILED|^E  " "$OUT/pytest.log" | head -30
timeout 300 python3 tools/probe.py C2 30 2>&1 | tee $OUT/probe_c2.log
VGICP_DEBUG_STAMPS=1 timeout 300 python3 tools/probe.py C2 30 2>&1 | grep stamps | grep -v "body launches" | tee -a $OUT/probe_c2.log
timeout 300 python3 tools/probe.py C5 10 2>&1 | tee $OUT/probe_c5.log
