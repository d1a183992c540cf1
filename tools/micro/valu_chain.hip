// Micro-probe: what one wave per SIMD gets out of dependent / independent fp64 VALU chains, v_readlane
// broadcasts and LDS read-modify-writes on gfx950 (cycles per operation = time * clock / operations).
// build: hipcc -O3 --offload-arch=gfx950 -o valu_chain valu_chain.hip ; run: ./valu_chain
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(64) void dep_fma(double* out, int n, double a, double b) {
  double x = threadIdx.x;
  for (int i = 0; i < n; ++i) x = fma(x, a, b);
  out[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ __launch_bounds__(64) void indep_fma(double* out, int n, double a, double b) {
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < n; i += 8) {
    x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
    x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ __launch_bounds__(64) void readlane_dist(double* out, int n, const double* pts) {
  const int lane = threadIdx.x;
  const double px = pts[3 * lane], py = pts[3 * lane + 1], pz = pts[3 * lane + 2];
  const double qx = px * 0.5, qy = py * 0.5, qz = pz * 0.5;
  double acc = 0.0;
  for (int i = 0; i < n; ++i) {
    const int c = i & 63;
    const double x = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(px), c), __builtin_amdgcn_readlane(__double2loint(px), c));
    const double y = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(py), c), __builtin_amdgcn_readlane(__double2loint(py), c));
    const double z = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(pz), c), __builtin_amdgcn_readlane(__double2loint(pz), c));
    const double dx = x - qx, dy = y - qy, dz = z - qz;
    acc += dx * dx + dy * dy + dz * dz;
  }
  out[blockIdx.x * 64 + lane] = acc;
}
__global__ __launch_bounds__(64) void lds_hist(double* out, int n, double s) {
  __shared__ unsigned hist[32][64];
  const int lane = threadIdx.x;
  for (int b = 0; b < 32; ++b) hist[b][lane] = 0;
  double v = lane * 0.37;
  for (int i = 0; i < n; ++i) {
    v = v * s + 0.123;
    int b = (int)v & 31;
    hist[b][lane] += 1u;
  }
  unsigned t = 0;
  for (int b = 0; b < 32; ++b) t += hist[b][lane];
  out[blockIdx.x * 64 + lane] = t + v;
}

template <typename F>
float timed(F&& launch) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch(); hipDeviceSynchronize();
  hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
  double *out, *pts;
  hipMalloc(&out, 4096 * 64 * 8); hipMalloc(&pts, 64 * 3 * 8);
  double h[192]; for (int i = 0; i < 192; ++i) h[i] = i * 0.01; hipMemcpy(pts, h, sizeof h, hipMemcpyHostToDevice);
  const int n = 100000;
  for (int grid : {1, 768, 1024}) {
    float t1 = timed([&] { hipLaunchKernelGGL(dep_fma, dim3(grid), dim3(64), 0, 0, out, n, 1.0000001, 1e-9); });
    float t2 = timed([&] { hipLaunchKernelGGL(indep_fma, dim3(grid), dim3(64), 0, 0, out, n, 1.0000001, 1e-9); });
    float t3 = timed([&] { hipLaunchKernelGGL(readlane_dist, dim3(grid), dim3(64), 0, 0, out, n, pts); });
    float t4 = timed([&] { hipLaunchKernelGGL(lds_hist, dim3(grid), dim3(64), 0, 0, out, n, 1.0001); });
    printf("grid %4d: dependent fma %.1f ns/op | 8 independent fma %.2f ns/op | readlane+dist2 %.1f ns/candidate | lds hist step %.1f ns\n",
           grid, t1 * 1e6 / n, t2 * 1e6 / n, t3 * 1e6 / n, t4 * 1e6 / n);
  }
  return 0;
}
