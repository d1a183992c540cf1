#!/bin/bash
# Developer aid: build the current csrc tree into eskf_lio_amd/lib_ab/<name>/libvgicp_hip.so (extra hipcc flags
# after the name), so that two variants can be timed in ONE gpurun session on ONE box with
# VGICP_LIB_PATH=eskf_lio_amd/lib_ab/<name>/libvgicp_hip.so (box-to-box variance is ~5 %).
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/eskf_lio_amd/lib_ab/$NAME
mkdir -p "$OUT" /tmp/ab_$NAME
cd "$ROOT/eskf_lio_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-unused-parameter $*"
# the kernels file with the Makefile's scheduler strategy (KERNEL_SCHED="" tools/ab_build.sh ... builds without it)
KS=${KERNEL_SCHED--mllvm -amdgpu-sched-strategy=max-memory-clause}
/opt/rocm/bin/hipcc $FLAGS $KS -c -o /tmp/ab_$NAME/k.o vgicp_kernels.hip &
/opt/rocm/bin/hipcc $FLAGS -Wno-unused-function -c -o /tmp/ab_$NAME/m.o vgicp_mapupdate.hip &
/opt/rocm/bin/hipcc $FLAGS -ffp-contract=off -Wno-unused-function -c -o /tmp/ab_$NAME/p.o vgicp_preprocess.hip &
/opt/rocm/bin/hipcc $FLAGS -c -o /tmp/ab_$NAME/c.o vgicp_capi.hip &
/opt/rocm/bin/hipcc $FLAGS -c -o /tmp/ab_$NAME/u.o vgicp_multi.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libvgicp_hip.so" /tmp/ab_$NAME/{k,m,p,c,u}.o -ldl
echo "built $OUT/libvgicp_hip.so"
