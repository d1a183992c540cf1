#!/bin/bash
set -u
OUT=gpurun_out/${1:-r02c}
mkdir -p "$OUT"
timeout 1500 python3 -m pytest tests -m gpu -q -k "persistent or degenerate or solve_step" > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest.log"
tail -8 "$OUT/pytest.log"
timeout 600 python3 bench.py > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"
tail -c 3000 "$OUT/bench_full.json"
bash tools/profile_gpu.sh $(basename $OUT)_c2
bash tools/profile_gpu.sh $(basename $OUT)_c5 --config C5 --steps 10
