// Micro-probe: accuracy of v_rcp_f64 and of one / two Newton steps on top of it, in ulps against the correctly
// rounded 1.0 / d.  Build: hipcc -O3 --offload-arch=gfx950 -o rcp_accuracy rcp_accuracy.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

__global__ void probe(const double* d, double* y0, double* y1, double* y2, double* y3, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = d[i];
  double y = __builtin_amdgcn_rcp(x);
  y0[i] = y;
  double e = fma(-x, y, 1.0);
  y = fma(y, e, y);
  y1[i] = y;
  e = fma(-x, y, 1.0);
  y2[i] = fma(y, e, y);
  // ONE third-order step on the bare instruction: e = 1 - x y0, y = y0 + y0 (e + e^2) -- 3 dependent operations instead of 4
  const double z = y0[i];
  const double f = fma(-x, z, 1.0);
  y3[i] = fma(z, fma(f, f, f), z);
}

static long long bits(double v) { long long b; std::memcpy(&b, &v, 8); return b; }

int main() {
  const int n = 1 << 22;
  std::vector<double> h(n), r0(n), r1(n), r2(n), r3(n);
  std::mt19937_64 rng(5);
  std::uniform_real_distribution<double> mant(1.0, 2.0);
  std::uniform_int_distribution<int> ex(-40, 40);
  for (int i = 0; i < n; ++i) h[i] = std::ldexp(mant(rng), ex(rng));
  double *d, *a, *b, *c, *c3;
  (void)hipMalloc(&d, n * 8); (void)hipMalloc(&a, n * 8); (void)hipMalloc(&b, n * 8); (void)hipMalloc(&c, n * 8); (void)hipMalloc(&c3, n * 8);
  (void)hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, d, a, b, c, c3, n);
  (void)hipMemcpy(r0.data(), a, n * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(r1.data(), b, n * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(r2.data(), c, n * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(r3.data(), c3, n * 8, hipMemcpyDeviceToHost);
  long long m3 = 0, exact3 = 0;
  long long m0 = 0, m1 = 0, m2 = 0, exact1 = 0, exact2 = 0;
  for (int i = 0; i < n; ++i) {
    const long long t = bits(1.0 / h[i]);
    const long long e0 = std::llabs(bits(r0[i]) - t), e1 = std::llabs(bits(r1[i]) - t), e2 = std::llabs(bits(r2[i]) - t);
    m0 = e0 > m0 ? e0 : m0; m1 = e1 > m1 ? e1 : m1; m2 = e2 > m2 ? e2 : m2;
    exact1 += e1 == 0; exact2 += e2 == 0;
    const long long e3 = std::llabs(bits(r3[i]) - t); m3 = e3 > m3 ? e3 : m3; exact3 += e3 == 0;
  }
  std::printf("v_rcp_f64: max %lld ulp | + 1 Newton step: max %lld ulp, exact %.4f | + 2 steps: max %lld ulp, exact %.4f (%d values)\n",
              m0, m1, exact1 / (double)n, m2, exact2 / (double)n, n);
  std::printf("one third-order step (3 dependent operations): max %lld ulp, exact %.4f\n", m3, exact3 / (double)n);
  return 0;
}
