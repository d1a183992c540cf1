// Micro-probe: latency of a flag ping-pong between two workgroups of the SAME XCD (workgroup-scope = sc0
// accesses served by the XCD's L2) and between two workgroups on DIFFERENT XCDs (agent-scope = sc1), plus the
// XCC id each workgroup reports. Build: hipcc -O3 --offload-arch=gfx950 -o scope_pingpong scope_pingpong.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef __attribute__((address_space(1))) unsigned int gu32;

template <int SCOPE>
__device__ __forceinline__ unsigned load_flag(unsigned* p) {
  return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, SCOPE);
}
template <int SCOPE>
__device__ __forceinline__ void store_flag(unsigned* p, unsigned v) {
  __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, SCOPE);
}

// MIXED (scope -1): the writer stores PLAIN (the line stays in its XCD's L2) and drains (s_waitcnt vmcnt(0)), the
// reader polls with sc1 loads (L1 bypassed, L2 served): a candidate fast path between workgroups of one XCD.
constexpr int kMixed = -1;
template <>
__device__ __forceinline__ unsigned load_flag<kMixed>(unsigned* p) {
  return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <>
__device__ __forceinline__ void store_flag<kMixed>(unsigned* p, unsigned v) {
  *(volatile unsigned*)p = v;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// workgroups a and b bounce a counter `rounds` times; everybody else leaves at once
template <int SCOPE>
__global__ void pingpong(unsigned* flags, int a, int b, int rounds, unsigned long long* ticks, unsigned* xcc) {
  if (threadIdx.x != 0) return;
  unsigned id = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  xcc[blockIdx.x] = id & 0xF;
  if ((int)blockIdx.x != a && (int)blockIdx.x != b) return;
  const bool first = (int)blockIdx.x == a;
  unsigned* to_a = flags;        // written by b, read by a
  unsigned* to_b = flags + 64;   // written by a, read by b (its own 256-byte line)
  const unsigned long long t0 = wall_clock64();
  bool lost = false;
  for (int r = 1; r <= rounds && !lost; ++r) {
    unsigned spins = 0;
    if (first) {
      store_flag<SCOPE>(to_b, (unsigned)r);
      while (load_flag<SCOPE>(to_a) != (unsigned)r) {
        if (++spins > 200000u) { lost = true; break; }   // never hang the GPU: a stale cache would spin forever
        __builtin_amdgcn_s_sleep(1);
      }
    } else {
      while (load_flag<SCOPE>(to_b) != (unsigned)r) {
        if (++spins > 200000u) { lost = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      store_flag<SCOPE>(to_a, (unsigned)r);
    }
  }
  if (lost) atomicAdd(&xcc[63], 0x100u);   // marker: the exchange did not complete at this scope
  if (first) *ticks = wall_clock64() - t0;
}

int main() {
  unsigned* flags;
  unsigned long long* ticks;
  unsigned* xcc;
  const int grid = 64;
  (void)hipMalloc(&flags, 4096);
  (void)hipMalloc(&ticks, 8);
  (void)hipMalloc(&xcc, grid * 4);
  std::vector<unsigned> hx(grid);
  const int rounds = 2000;
  auto run = [&](int scope, int a, int b, const char* what) {
    (void)hipMemset(flags, 0, 4096);
    (void)hipMemset(ticks, 0, 8);
    (void)hipMemset(xcc, 0, grid * 4);
    if (scope == -1) hipLaunchKernelGGL(pingpong<kMixed>, dim3(grid), dim3(64), 0, 0, flags, a, b, rounds, ticks, xcc);
    else if (scope == 0) hipLaunchKernelGGL(pingpong<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(grid), dim3(64), 0, 0, flags, a, b, rounds, ticks, xcc);
    else hipLaunchKernelGGL(pingpong<__HIP_MEMORY_SCOPE_AGENT>, dim3(grid), dim3(64), 0, 0, flags, a, b, rounds, ticks, xcc);
    if (hipDeviceSynchronize() != hipSuccess) { std::printf("%s: launch failed\n", what); return; }
    unsigned long long t = 0;
    (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hx.data(), xcc, grid * 4, hipMemcpyDeviceToHost);
    std::printf("%s: workgroups %d (xcc %u) <-> %d (xcc %u): %.3f us per round trip%s\n", what, a, hx[a] & 0xF,
                b, hx[b] & 0xF, t * 0.01 / rounds, (hx[63] & 0x100u) ? "  [DID NOT COMPLETE: values not visible at this scope]" : "");
  };
  run(1, 0, 1, "agent scope (sc1), neighbours in dispatch order");
  run(1, 0, 8, "agent scope (sc1), 8 apart");
  run(0, 0, 8, "workgroup scope (sc0), 8 apart");
  run(0, 0, 16, "workgroup scope (sc0), 16 apart");
  run(-1, 0, 8, "plain store + drain / sc1 load, 8 apart (same XCD)");
  run(-1, 0, 16, "plain store + drain / sc1 load, 16 apart (same XCD)");
  run(-1, 0, 1, "plain store + drain / sc1 load, neighbours (different XCDs)");
  std::printf("xcc ids of workgroups 0..15:");
  for (int i = 0; i < 16; ++i) std::printf(" %u", hx[i] & 0xF);
  std::printf("\n");
  return 0;
}
