#!/bin/bash
# Round 5 evidence in one gpurun call: bash tools/round5_evidence.sh <tag>
set -u
T=${1:-r12}
mkdir -p gpurun_out/$T
timeout 900 python3 bench.py > gpurun_out/$T/bench_full.json 2> gpurun_out/$T/bench_full.err
tail -c 400 gpurun_out/$T/bench_full.json
bash tools/profile_gpu.sh ${T}_c2
bash tools/profile_gpu.sh ${T}_c5 --config C5 --steps 10
# frame chain: per-kernel stats
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fk && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fk -o f -- python3 $GRAFT_REPO_ROOT/tools/probe_frame.py 30 60000 > /dev/null 2>&1; cp $(find /tmp/fk -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/$T/frame_kernel_stats.csv )
bash tools/prep_pmc.sh > gpurun_out/$T/prep_pmc.txt 2>&1
timeout 200 python3 tools/soak_upload.py 45 > gpurun_out/$T/soak_upload.txt 2>&1
VGICP_UPLOAD_THREADS=4 timeout 200 python3 tools/soak_upload.py 30 >> gpurun_out/$T/soak_upload.txt 2>&1
timeout 300 python3 tools/soak_exchange.py 60 > gpurun_out/$T/soak_exchange.txt 2>&1
python3 tools/probe_munmap.py > gpurun_out/$T/munmap.txt 2>&1
python3 -m pytest tests/test_replay.py -m gpu -q -k street -s 2>&1 | grep -v amdgpu | tail -12 > gpurun_out/$T/street.txt
tail -3 gpurun_out/$T/soak_upload.txt gpurun_out/$T/soak_exchange.txt gpurun_out/$T/street.txt
