#!/bin/bash
set -u
OUT=gpurun_out/${1:-r02b}
mkdir -p "$OUT"
timeout 1500 python3 -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest.log"
tail -15 "$OUT/pytest.log"
timeout 600 python3 tools/probe_upload.py C2 30 > "$OUT/upload_c2.log" 2>&1
cat "$OUT/upload_c2.log"
nproc; lscpu | grep -E "Model name|Socket|Thread|NUMA node\(s\)"
