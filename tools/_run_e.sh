cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r11
timeout 2300 python -m pytest tests/ -x -q -m gpu > gpurun_out/r11_gpu_suite.log 2>&1; echo "gpu suite rc=$?"; grep -E "passed|failed" gpurun_out/r11_gpu_suite.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash tools/profile_gpu.sh r11_c2 > /dev/null 2>&1
bash tools/profile_gpu.sh r11_c5 --config C5 --steps 10 > /dev/null 2>&1
timeout 300 python3 tools/probe_upload.py C2 30 > gpurun_out/r11/upload.log 2>&1
for N in 2 4 8; do BENCH_SHARE_DEVICE=1 timeout 600 python3 bench.py --gpus $N --steps 50 --warmup 5 --no-cpu-baseline --no-c5 --no-frame-chain 2>/dev/null | tail -1 > gpurun_out/r11/inproc_$N.json; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fr; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fr -o f -- python3 $GRAFT_REPO_ROOT/tools/probe_frame.py 30 60000 > $GRAFT_REPO_ROOT/gpurun_out/r11/frame_probe.json 2>/dev/null
cp $(find /tmp/fr -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r11/frame_kernel_stats.csv
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do ./examples/frame_chain 10 60000 2>&1 | tail -5; done > gpurun_out/r11/frame_chain_example.txt
