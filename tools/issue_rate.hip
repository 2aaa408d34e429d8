// dev tool / evidence for bench.py's `valu_issue_frac`: how many wave64 VALU instructions one SIMD of an MI355X issues per
// second, for plain v_fma_f32, packed v_pk_fma_f32 and the quarter-class v_rsq_f32, at 1 .. 8 wavefronts per SIMD.
// Every lane runs ILP independent chains of N dependent instructions (inline asm: nothing is folded or reordered away).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/issue_rate tools/issue_rate.hip      Run: tools/issue_rate > profiles/r04_issue_rate.txt
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ILP = 8, ITER = 2048;
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(64) void k_issue(float *out, float a, float b) {
  float r[ILP];
  f2 p[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) { r[i] = a + i + threadIdx.x; p[i] = (f2){a + i, b + threadIdx.x}; }
  const f2 a2 = {a, a}, b2 = {b, b};
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
      if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(a2), "v"(b2));
      if (KIND == 2) asm volatile("v_rsq_f32 %0, %0" : "+v"(r[i]));
      if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(a2));
      if (KIND == 4) asm volatile("v_mov_b32 %0, %0" : "+v"(r[i]));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += r[i] + p[i].x + p[i].y;
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int KIND>
double run(int waves_per_simd, int cus, float *d_out) {
  // one wavefront per workgroup; the dispatcher spreads them over the SIMDs: cus * 4 SIMDs * waves_per_simd workgroups
  const int wgs = cus * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_issue<KIND>, dim3(wgs), dim3(64), 0, 0, d_out, 1.0001f, 0.5f);   // warm-up
  CHK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_issue<KIND>, dim3(wgs), dim3(64), 0, 0, d_out, 1.0001f, 0.5f);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double insts = (double)wgs * ITER * ILP;            // wave-instructions
  return insts / (best * 1e-3) / (cus * 4.0);               // per second and SIMD
}

int main() {
  hipDeviceProp_t p;
  CHK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  float *d_out;
  CHK(hipMalloc(&d_out, (size_t)cus * 4 * 8 * 64 * sizeof(float)));
  printf("# %s, %d CUs, clock %d MHz: wave64 VALU instructions issued per second and SIMD (x 1e9), %d independent chains per lane\n", p.name, cus,
         p.clockRate / 1000, ILP);
  printf("# waves/SIMD   v_fma_f32   v_pk_fma_f32   v_pk_mul_f32   v_rsq_f32   v_mov_b32\n");
  for (int w = 1; w <= 8; ++w)
    printf("%12d %11.3f %14.3f %14.3f %11.3f %11.3f\n", w, run<0>(w, cus, d_out) / 1e9, run<1>(w, cus, d_out) / 1e9, run<3>(w, cus, d_out) / 1e9,
           run<2>(w, cus, d_out) / 1e9, run<4>(w, cus, d_out) / 1e9);
  printf("# bench.py divides SQ_INSTS_VALU per launch by launch time x 1024 SIMDs x 0.6e9: the v_fma_f32 / v_pk_fma_f32 rows say what the\n"
         "# denominator should be (a wave64 instruction per 4 cycles at 2.4 GHz = 0.6e9; per 2 cycles = 1.2e9)\n");
  return 0;
}
