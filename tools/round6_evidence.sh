#!/bin/bash
# Round 6 evidence in one gpurun call: bash tools/round6_evidence.sh <tag>
# (reduce with: python tools/summarize_profile.py <tag>_c2 ; python tools/summarize_profile.py <tag>_c5 1000000)
set -u
T=${1:-r15}
mkdir -p gpurun_out/$T
timeout 900 python3 bench.py > gpurun_out/$T/bench_full.json 2> gpurun_out/$T/bench_full.err
tail -c 400 gpurun_out/$T/bench_full.json
bash tools/profile_gpu.sh ${T}_c2
bash tools/profile_gpu.sh ${T}_c5 --config C5 --steps 10
# frame chain: per-kernel stats
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fk && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fk -o f -- python3 $GRAFT_REPO_ROOT/tools/probe_frame.py 30 60000 > /dev/null 2>&1; cp $(find /tmp/fk -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/$T/frame_kernel_stats.csv )
bash tools/prep_pmc.sh > gpurun_out/$T/prep_pmc.txt 2>&1
# the C2 round's phase clocks, the drop-in classes' host time, the upload's switches
VGICP_DEBUG_STAMPS=1 timeout 300 python3 tools/probe.py C2 50 2>&1 | grep -E "eager|stamps" > gpurun_out/$T/c2_stamps.txt
VGICP_DEBUG_STAMPS=1 timeout 300 python3 tools/probe.py C5 10 2>&1 | grep -E "eager|stamps" >> gpurun_out/$T/c2_stamps.txt
timeout 300 python3 tools/probe_eager.py 30 60000 2>&1 | grep -v "ICP not" > gpurun_out/$T/dropin_host_time.txt
for h in 0 2; do echo "hash helper threads: $h (back to back, then with the shadow grid's worker drained between frames)" >> gpurun_out/$T/dropin_host_time.txt
  timeout 300 python3 tools/probe_eager.py 30 60000 - $h 2>&1 | grep "eager    full" >> gpurun_out/$T/dropin_host_time.txt
  timeout 300 python3 tools/probe_eager.py 30 60000 drain $h 2>&1 | grep "eager    full" >> gpurun_out/$T/dropin_host_time.txt; done
# the hand-written sort alone: against std::stable_sort, then per-kernel times
( timeout 300 eskf_lio_amd/lib/sort_check; bash tools/ab_sort.sh sort_check 60000 100000 1000000 ) > gpurun_out/$T/sort_standalone.txt 2>&1
timeout 600 bash tools/ab_upload_bench.sh > gpurun_out/$T/upload_switches.txt 2>&1
# soaks
timeout 200 python3 tools/soak_upload.py 45 > gpurun_out/$T/soak_upload.txt 2>&1
timeout 300 python3 tools/soak_exchange.py 60 > gpurun_out/$T/soak_exchange.txt 2>&1
timeout 300 python3 tools/soak_preprocess.py 90 11 > gpurun_out/$T/soak_preprocess.txt 2>&1
timeout 200 python3 tools/soak_dropin.py 60 2>&1 | grep -v 'ICP not' | tail -n 3 > gpurun_out/$T/soak_dropin.txt
python3 -m pytest tests/test_replay.py -m gpu -q -k street -s 2>&1 | grep -v amdgpu | tail -12 > gpurun_out/$T/street.txt
tail -n 3 gpurun_out/$T/soak_upload.txt gpurun_out/$T/soak_exchange.txt gpurun_out/$T/soak_preprocess.txt gpurun_out/$T/street.txt
