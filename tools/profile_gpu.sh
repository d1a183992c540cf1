#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   1. --kernel-trace --stats            per-kernel durations
#   2. --pmc FETCH_SIZE                  read-side memory traffic      } separate passes, counters only
#   3. --pmc WRITE_SIZE                  write-side memory traffic     } (MI355X_MICROARCH.md §HBM, §PMC slots)
# Output: gpurun_out/<tag>/ ; tools/summarize_profile.py turns it into profiles/<tag>_*.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--steps 30 --warmup 3 --no-cpu-baseline --no-c5 --no-frame-chain $*"
python3 bench.py $ARGS > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py $ARGS > "$OUT/bench_pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py $ARGS > "$OUT/bench_pmc_write.json" 2> "$OUT/pmc_write.err"
find "$OUT" -name "*.csv" | head -20
tail -c 600 "$OUT/bench_plain.json"
