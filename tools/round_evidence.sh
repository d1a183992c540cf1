#!/bin/bash
# One gpurun call for the evidence of a round: the default bench line, the three rocprofv3 passes at C2 and C5
# (tools/profile_gpu.sh -> gpurun_out/<tag>_c2, <tag>_c5; reduce with tools/summarize_profile.py) and the upload probe.
# usage: gpurun -- bash tools/round_evidence.sh <tag>
set -u
T=${1:-r02f}
mkdir -p gpurun_out/$T
timeout 600 python3 bench.py > gpurun_out/$T/bench_full.json 2> gpurun_out/$T/bench_full.err
tail -c 1500 gpurun_out/$T/bench_full.json
bash tools/profile_gpu.sh ${T}_c2
bash tools/profile_gpu.sh ${T}_c5 --config C5 --steps 10
timeout 300 python3 tools/probe_upload.py C2 30 | tee gpurun_out/$T/upload.log
