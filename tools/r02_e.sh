#!/bin/bash
set -u
OUT=gpurun_out/${1:-r02e}; mkdir -p $OUT
timeout 1500 python3 -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest.log"; tail -6 "$OUT/pytest.log"
for pm in 0 0.01 0.03 0.06 0.12 0.5; do
  echo "== prefetch margin $pm"
  VGICP_PREFETCH_MARGIN=$pm VGICP_DEBUG_STAMPS=1 timeout 300 python3 tools/probe.py C2 30 2>&1 | grep -v "body launches" | tee -a $OUT/probe_pm.log
done
