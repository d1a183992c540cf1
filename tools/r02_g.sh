#!/bin/bash
set -u
OUT=gpurun_out/${1:-r02g}; mkdir -p $OUT
timeout 1500 python3 -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest.log"; grep -E "passed|failed|^FAILED|^E  " "$OUT/pytest.log" | head -30
BENCH_FORCE_COMM=1 timeout 300 python3 bench.py --steps 20 --no-cpu-baseline > $OUT/bench_comm1.json 2> $OUT/bench_comm1.err; tail -c 1200 $OUT/bench_comm1.json; tail -3 $OUT/bench_comm1.err
VGICP_DEBUG_STAMPS=1 timeout 300 python3 tools/probe.py C2 30 2>&1 | grep -v "body launches" | tee $OUT/probe_c2.log
