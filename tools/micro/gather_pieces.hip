// Micro-probe: what a 16-byte piece of a scattered 128-byte record costs a compute unit on gfx950.
//
// The accumulate phase of the registration at C5 (1M points, 10M voxels) reads, per matched point, the 96-byte
// payload of one voxel record as SIX 16-byte loads of one lane (global_load_dwordx4), every lane of a wave at a
// different 128-byte line of a multi-GB table.  DESIGN.md §4 has two measurements that disagree about what such a
// piece costs (six extra pieces per match from the same line: +26 us per round; five pieces instead of six: nothing).
// This probe takes the kernel away: 8 waves per CU on every CU (one 512-thread workgroup each, as the persistent
// launch), every active lane loads PIECES x 16 bytes of a random line per step, STEPS steps, the steps of a wave
// dependent through the sum only (as the kernel's accumulators).  Reported: ns per wave-step, and cycles per
// lane-piece per CU at the clock measured by s_memtime.
//   build: hipcc -O3 --offload-arch=gfx950 -o gather_pieces gather_pieces.hip ; run: ./gather_pieces [table MB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

template <int PIECES>
__global__ __launch_bounds__(512) void gather(const char* __restrict__ table, const uint32_t* __restrict__ lines,
                                              int steps, unsigned long long active_mask, double* __restrict__ out) {
  const uint32_t tid = blockIdx.x * 512u + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  const bool active = (active_mask >> lane) & 1ull;
  double acc = 0.0;
  for (int s = 0; s < steps; ++s) {
    const uint32_t line = lines[(size_t)s * gridDim.x * 512u + tid];  // coalesced: the address does not depend on acc
    if (active) {
      const d2* p = reinterpret_cast<const d2*>(table + (size_t)line * 128u);
      d2 v[PIECES];
#pragma unroll
      for (int k = 0; k < PIECES; ++k) v[k] = p[k];
#pragma unroll
      for (int k = 0; k < PIECES; ++k) acc += v[k].x + v[k].y;
    }
  }
  out[tid] = acc;
}

template <int PIECES>
float run(const char* table, const uint32_t* lines, int steps, unsigned long long mask, double* out, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(gather<PIECES>, dim3(grid), dim3(512), 0, 0, table, lines, steps, mask, out);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(gather<PIECES>, dim3(grid), dim3(512), 0, 0, table, lines, steps, mask, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best;
}

int main(int argc, char** argv) {
  const size_t table_mb = argc > 1 ? (size_t)atol(argv[1]) : 5120;
  const size_t table_bytes = table_mb << 20, n_lines = table_bytes / 128;
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int grid = cus, steps = 64;
  char* table; uint32_t* lines; double* out;
  hipMalloc(&table, table_bytes); hipMemset(table, 0, table_bytes);
  const size_t n_idx = (size_t)steps * grid * 512;
  std::vector<uint32_t> h(n_idx);
  unsigned long long x = 88172645463325252ull;
  for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)(x % n_lines); }
  hipMalloc(&lines, n_idx * 4); hipMemcpy(lines, h.data(), n_idx * 4, hipMemcpyHostToDevice);
  hipMalloc(&out, (size_t)grid * 512 * 8);
  printf("table %zu MB, %d CUs x 8 waves, %d steps per wave; every active lane reads PIECES x 16 B of its own random 128-B line per step\n",
         table_mb, cus, steps);
  const unsigned long long all = ~0ull, half = 0x5555555555555555ull;
  for (int pass = 0; pass < 2; ++pass) {
    const unsigned long long mask = pass == 0 ? all : half;
    const int lanes = pass == 0 ? 64 : 32;
    float ms[8];
    ms[0] = run<1>(table, lines, steps, mask, out, grid); ms[1] = run<2>(table, lines, steps, mask, out, grid);
    ms[2] = run<3>(table, lines, steps, mask, out, grid); ms[3] = run<4>(table, lines, steps, mask, out, grid);
    ms[4] = run<5>(table, lines, steps, mask, out, grid); ms[5] = run<6>(table, lines, steps, mask, out, grid);
    ms[6] = run<7>(table, lines, steps, mask, out, grid); ms[7] = run<8>(table, lines, steps, mask, out, grid);
    for (int p = 1; p <= 8; ++p) {
      const double per_step_ns = ms[p - 1] * 1e6 / steps;                       // all 8 waves of a CU run concurrently
      const double lines_total = (double)steps * grid * 8 * lanes;
      printf("%2d active lanes, %d pieces: %8.1f ns per wave-step, %6.2f ns per line per CU, %5.2f TB/s of 128-B lines, %6.2f ns per lane-piece per CU\n",
             lanes, p, per_step_ns, ms[p - 1] * 1e6 / (steps * 8.0 * lanes), lines_total * 128.0 / (ms[p - 1] * 1e-3) / 1e12,
             ms[p - 1] * 1e6 / (steps * 8.0 * lanes * p));
    }
  }
  return 0;
}
