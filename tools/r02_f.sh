#!/bin/bash
set -u
T=${1:-r02f}
mkdir -p gpurun_out/$T
timeout 600 python3 bench.py > gpurun_out/$T/bench_full.json 2> gpurun_out/$T/bench_full.err
tail -c 1500 gpurun_out/$T/bench_full.json
bash tools/profile_gpu.sh ${T}_c2
bash tools/profile_gpu.sh ${T}_c5 --config C5 --steps 10
timeout 300 python3 tools/probe_upload.py C2 30 | tee gpurun_out/$T/upload.log
