#!/bin/bash
# round-2 first GPU pass: parity suite, then stamped probes of the persistent launch at C2 and C5
set -u
OUT=gpurun_out/${1:-r02a}
mkdir -p "$OUT"
timeout 900 python3 -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest.log"
tail -5 "$OUT/pytest.log"
VGICP_DEBUG_STAMPS=1 timeout 300 python3 tools/probe.py C2 30 > "$OUT/probe_c2.log" 2>&1
cat "$OUT/probe_c2.log"
VGICP_DEBUG_STAMPS=1 timeout 600 python3 tools/probe.py C5 10 > "$OUT/probe_c5.log" 2>&1
cat "$OUT/probe_c5.log"
