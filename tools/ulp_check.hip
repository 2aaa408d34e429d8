// dev tool: are the engine's range-restricted exact sqrt / division bit-identical to IEEE sqrtf and '/'
// (and to the host) on gfx950?  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ulp_check.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
__device__ float fast_sqrt(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  const float sd = __int_as_float(__float_as_int(s) - 1), su = __int_as_float(__float_as_int(s) + 1);
  const float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
  float t = (0.0f >= rd) ? sd : s;
  return (0.0f < ru) ? su : t;
}
__device__ float fast_div(float a, float b) {
  float r = __builtin_amdgcn_rcpf(b);
  r = __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
  float q = a * r;
  q = __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
  return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
}
__global__ void k(const float *x, const float *y, float *o, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  o[4 * i + 0] = fast_sqrt(x[i]);
  o[4 * i + 1] = sqrtf(x[i]);
  o[4 * i + 2] = fast_div(y[i], x[i]);
  o[4 * i + 3] = y[i] / x[i];
}
int main() {
  size_t n = 1 << 24;
  std::vector<float> x(n), y(n), o(4 * n);
  srand(1);
  for (size_t i = 0; i < n; ++i) {
    double u = rand() / (double)RAND_MAX, v = rand() / (double)RAND_MAX;
    // q / R ranges of the PSF: log-uniform 1e-12..1e3 for the divisor / sqrt argument, 1e-12..1 numerator
    x[i] = (float)pow(10.0, -12.0 + 15.0 * u);
    y[i] = (float)pow(10.0, -12.0 + 12.0 * v);
    // R == 0 only ever meets |sin R| == 0 (0/0 = NaN on both paths); a/0 with a > 0 cannot occur
    if (i % 1024 == 0) { x[i] = 0.0f; y[i] = 0.0f; }
    if (i % 4096 == 1) y[i] = 0.0f;
  }
  float *dx, *dy, *dout;
  (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&dy, n * 4); (void)hipMalloc(&dout, 4 * n * 4);
  (void)hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dy, y.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx, dy, dout, n);
  (void)hipMemcpy(o.data(), dout, 4 * n * 4, hipMemcpyDeviceToHost);
  long bs = 0, bd = 0, hs = 0, hd = 0;
  for (size_t i = 0; i < n; ++i) {
    float hsq = sqrtf(x[i]), hdv = y[i] / x[i];
    if (memcmp(&o[4 * i], &o[4 * i + 1], 4)) { if (bs++ < 3) printf("sqrt x=%.9g fast=%.9g ieee=%.9g\n", x[i], o[4 * i], o[4 * i + 1]); }
    if (memcmp(&o[4 * i + 2], &o[4 * i + 3], 4) && !(o[4 * i + 2] != o[4 * i + 2] && o[4 * i + 3] != o[4 * i + 3])) { if (bd++ < 3) printf("div %.9g/%.9g fast=%.9g ieee=%.9g\n", y[i], x[i], o[4 * i + 2], o[4 * i + 3]); }
    if (memcmp(&o[4 * i], &hsq, 4)) hs++;
    if (memcmp(&o[4 * i + 2], &hdv, 4) && !(hdv != hdv && o[4 * i + 2] != o[4 * i + 2])) hd++;
  }
  printf("n=%zu  fast_sqrt!=sqrtf(dev): %ld  fast_div!=div(dev): %ld  fast_sqrt!=host: %ld  fast_div!=host: %ld\n", n, bs, bd, hs, hd);
  return (bs || bd || hs || hd) ? 1 : 0;
}
