// svr_hip.hip -- MI355X (gfx950) slice-to-volume super-resolution engine behind the C-ABI of
// include/svr_hip.h.  Written for CDNA4 only: 64-wide wavefronts; the PSF evaluated row per lane, two taps per lane-op
// (v_pk_fma_f32); the scatter with one wavefront owning four planes of a tile's LDS box (back_wave_kernel), the gather
// with a workgroup per tile over (pixel, plane) units (fwd_unit_kernel); optionally the evaluated taps kept in HBM and
// streamed (the coefficient table, k_coeff_build); hardware float atomics only at the flush.
//
// Reference behaviour being replaced (citations: RC.cu = source/reconstructionGPU2/
// reconstruction_cuda2.cu, RC.cuh = include/reconstruction_cuda2.cuh, RVH =
// include/recon_volumeHelper.cuh, RG.cc = irtkReconstructionGPU.cc).  See DESIGN.md for the
// canonical float32 PSF sequence and the list of reproduced / not reproduced quirks.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -munsafe-fp-atomics ... (build.py)
#include <hip/hip_runtime.h>
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/svr_hip.h"

// csrc/svr_sort.hip: radix sort / prefix sum of the vendor's primitives library for the work lists of svr_cell.inc
int svr_sort_keys_u64(void *tmp, size_t *tmp_bytes, const uint64_t *keys_in, uint64_t *keys_out, size_t n, int end_bit, hipStream_t stream);
int svr_inclusive_sum_u32(void *tmp, size_t *tmp_bytes, const uint32_t *in, uint32_t *out, size_t n, hipStream_t stream);

// ------------------------------------------------------------------------------------------
// constants of the reference configuration section (RC.cuh:54-75)
// ------------------------------------------------------------------------------------------
#define SVR_STEP 0.0001f        // __step
#define PSF_SUPPORT 16          // MAX_PSF_SUPPORT with USE_INFINITE_PSF_SUPPORT 1
#define PSF_CENTRE 7            // (MAX_PSF_SUPPORT - 1) / 2   (RC.cu:219)
// `abs(oldPSF - psfval) < PSF_EPSILON` compares a float against the double 0.00001
// (RC.cuh:72, RC.cu:238); for a float f that is exactly  f <= 0.00001f  because 0.00001f is
// the largest float below 1e-5.
#define PSF_EPS_F 0.00001f
#define LDS_ROW 20              // floats per PSF row in LDS (16 + pad: conflict-free b128 stores)
#define WAVES_PER_BLOCK 4
#define CHUNK_PIX 2048          // slice-grid pixels per block in the EM kernels

namespace {

// ------------------------------------------------------------------------------------------
// host float helpers with the literal operation order of RVH:134-159 (no FMA contraction)
// ------------------------------------------------------------------------------------------
void matmul4(const float *A, const float *B, float *C) {
  float t[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      t[i * 4 + j] = A[i * 4 + 0] * B[0 * 4 + j] + A[i * 4 + 1] * B[1 * 4 + j] +
                     A[i * 4 + 2] * B[2 * 4 + j] + A[i * 4 + 3] * B[3 * 4 + j];
  memcpy(C, t, sizeof(t));
}
void matvec3_host(const float *M, const float v[3], float out[3]) {
  out[0] = M[0] * v[0] + M[1] * v[1] + M[2] * v[2] + M[3];
  out[1] = M[4] * v[0] + M[5] * v[1] + M[6] * v[2] + M[7];
  out[2] = M[8] * v[0] + M[9] * v[1] + M[10] * v[2] + M[11];
}

// per-slice constants, 64 floats, read through the scalar cache (slice index is wave-uniform)
struct SliceConst {
  float I2W[12];   // rows 0..2 of sliceI2W
  float T[12];     // slice transformation
  float A[12];     // combInvTrans = sliceW2I * Tinv * reconI2W  (RC.cu:223)
  float Lp[9];     // linear part of A scaled to calcPSF's argument space
  float dim[3];    // slice voxel dims (dx, dy, thickness)
  float kx, ky, inv2s2;
  float dd, w;     // Gaussian recurrence along a row: (z' step)^2, and exp(-2 dd inv2s2) or -1 (see gauss_pairs)
  int own;         // volume axis whose planes the scatter's slots own: 1 = y, 2 = z (the one closer to the slice normal)
  float invD;      // 1 / determinant of the in-plane map over (x, lane axis), or 0 if degenerate (unit_is_dead)
  float pad[9];
};
static_assert(sizeof(SliceConst) == 256, "SliceConst layout");

struct VolGeom {
  int vx, vy, vz;
  float W2I[12];   // rows 0..2 of reconstructedW2I
  float c0[3];     // d_PSFI2W * ((PSFsize-1)/2)  (RC.cu:172)
  int pvr;         // patch-to-volume constants
};

// ------------------------------------------------------------------------------------------
// canonical PSF (bit-identical to oracle/svr_oracle.c psf_canon: IEEE fma/mul/add/rint + integer ops)
// ------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ float canon_exp_neg(float a) {
  const float LOG2E = 1.442695040888963407359924681001892137426645954152985934135449406931f;
  const float L2U = 0.693145751953125f;
  const float L2L = 1.428606765330187045e-06f;
  float d = -a;
  float q = __builtin_rintf(d * LOG2E);
  float s = __builtin_fmaf(q, -L2U, d);
  s = __builtin_fmaf(q, -L2L, s);
  float u = 0.000198527617612853646278381f;
  u = __builtin_fmaf(u, s, 0.00139304355252534151077271f);
  u = __builtin_fmaf(u, s, 0.00833336077630519866943359f);
  u = __builtin_fmaf(u, s, 0.0416664853692054748535156f);
  u = __builtin_fmaf(u, s, 0.166666671633720397949219f);
  u = __builtin_fmaf(u, s, 0.5f);
  u = __builtin_fmaf(s * s, u, s) + 1.0f;
  float r = ldexpf(u, (int)q);
  return (a > 87.0f) ? 0.0f : r;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ void matvec3(const float *M, float x, float y, float z, float &a,
                                        float &b, float &c) {
  a = M[0] * x + M[1] * y + M[2] * z + M[3];
  b = M[4] * x + M[5] * y + M[6] * z + M[7];
  c = M[8] * x + M[9] * y + M[10] * z + M[11];
}

enum { MODE_GAUSS = 0, MODE_FWD = 1, MODE_BACK = 2, MODE_BIAS = 3 };

struct PsfArgs {
  const SliceConst *sc;
  VolGeom vg;
  int sx, sy;               // slice grid
  const uint32_t *list;     // pixel indices to process
  uint32_t n;
  const float *slices;
  const float *mask;
  float *psf_sums;
  const float *scales;        // per slice
  const float *bias;          // per-pixel log bias field, NULL when _disableBiasC (RC.cu:200-203)
  const unsigned char *flag;  // optional per-pixel activity flag replacing `v_PSF_sums != 0` (Gaussian pass 2)
  unsigned char *flag_out;    // Gaussian pass 1: 1 where the pixel's sume passed the threshold
  const unsigned char *spx;   // PVR superpixel masks [ns][64*64] of '0'/'1' or NULL (ImagePatch2D.cuh:51)
  // gauss (MODE_BIAS reuses recon/volw for the bias volume / its accumulated weights)
  float *recon, *volw;
  int *voxcount;
  // forward
  const float *vol;
  const float2 *volm;       // SVR gather: {V m, m} per voxel, packed by k_pack_volm right before the pass (NULL: vol + mask)
  float *simslices, *simweights;
  unsigned char *siminside;
  unsigned char *slice_inside;  // [slice]: some pixel of the slice has siminside == 1 (k_slice_inside; the cell gather's finish sets it itself: siminside only ever
                                // goes from 0 to 1 between two Gaussian passes, so the per-slice flag needs no pass of its own)
  // back
  const float *weights, *slice_weights;
  float *addon, *cmap;
  // coefficient table (COEFF instantiations): the evaluated taps of every live unit of every PSF pixel, kept in HBM
  const float4 *coeff;        // [pixel id][16 units][4 tap quads][16 rows] float4
  const uint32_t *coeff_id;   // slice-grid index -> pixel id
};

// per-pixel state shared by all modes
struct PixelState {
  int cxi, cyi, czi;   // rounded centre voxel psfxyz (RC.cu:225-226)
  float bx, by, bz;    // scaled residual at the centre (canonical form)
};

__device__ __forceinline__ int clampi(float f) {
  // |centre| beyond 2^20 is out of any volume; clamp so the int arithmetic cannot overflow
  f = fminf(fmaxf(f, -1048576.0f), 1048576.0f);
  return (int)f;   // NaN -> 0 after the clamp's fmin/fmax semantics
}

__device__ __forceinline__ PixelState pixel_setup(const SliceConst &S, const VolGeom &vg, int px,
                                                  int py) {
  PixelState P;
  float wx, wy, wz, tx, ty, tz, vx_, vy_, vz_;
  // d_reconstructedW2I * (slicesTransformation * (sliceI2W * slicePos))  RC.cu:225
  matvec3(S.I2W, (float)px, (float)py, 0.0f, wx, wy, wz);
  matvec3(S.T, wx, wy, wz, tx, ty, tz);
  matvec3(vg.W2I, tx, ty, tz, vx_, vy_, vz_);
  float cx = roundf(vx_), cy = roundf(vy_), cz = roundf(vz_);
  P.cxi = clampi(cx); P.cyi = clampi(cy); P.czi = clampi(cz);
  double d0 = (double)S.A[0] * cx + (double)S.A[1] * cy + (double)S.A[2] * cz + (double)S.A[3] - (double)(float)px;
  double d1 = (double)S.A[4] * cx + (double)S.A[5] * cy + (double)S.A[6] * cz + (double)S.A[7] - (double)(float)py;
  double d2 = (double)S.A[8] * cx + (double)S.A[9] * cy + (double)S.A[10] * cz + (double)S.A[11] - 0.0;
  P.bx = (float)((d0 * S.dim[0] - vg.c0[0]) * S.kx);
  P.by = (float)((d1 * S.dim[1] - vg.c0[1]) * S.ky);
  P.bz = vg.pvr ? (float)(d2 * S.dim[2] / 2.5 - vg.c0[2]) : (float)(d2 * S.dim[2] - vg.c0[2]);
  return P;
}

// hot per-slice constants held in registers (SGPRs: the slice index is wave-uniform)
struct RowConst {
  float Lp[9];
  float inv2s2;
  float dd, w;
  float invD;
};
__device__ __forceinline__ float uniform_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ RowConst load_row_const(const SliceConst &S) {
  RowConst R;
#pragma unroll
  for (int i = 0; i < 9; ++i) R.Lp[i] = S.Lp[i];
  // (through readfirstlane: the four values stay in scalar registers; read as plain struct fields the vectoriser fuses
  // them into one 16-byte load of a stack copy of the struct -- 32 bytes of scratch per lane for nothing)
  R.inv2s2 = uniform_f(S.inv2s2);
  R.dd = uniform_f(S.dd);
  R.w = uniform_f(S.w);
  R.invD = uniform_f(S.invD);
  return R;
}

// One (y,z) row of a pixel's footprint: walks the 16 x-taps sequentially with the
// epsilon-skip (RC.cu:233-239).  out[x] = psf, or -1 if skipped.  fy, fz = row offsets in
// [-7, 8] relative to the centre voxel.
// The PSF of the row's taps is evaluated STAGE BY STAGE over chunks of EVAL_CHUNK taps (arrays in
// registers): every stage is EVAL_CHUNK independent instructions, which is what lets one wave
// issue back to back -- evaluated tap after tap the dependent chains (sqrt, division, two
// polynomials) left the SIMDs latency-bound (forward time scaled 1:1 with occupancy).  The
// arithmetic per tap is exactly the oracle's psf_canon, so the values stay bit-identical.
// N = PSF support (16 for SVR, 12 for PVR), CENTRE = (N-1)/2; PVR selects the patch-to-volume
// constants (sinc_pi Taylor branch, strict float epsilon; R2/include/pointSpreadFunction.cuh:45-70,
// R2/include/reconConfig.cuh:138).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 bc2(float c) { return (f2)(c); }

// H pairs of consecutive x-taps of one (y,z) row, first tap at lattice offset fx0: val2[i] = psf of taps
// (fx0 + 2i, fx0 + 2i + 1).  rowx/rowy/rowz = the row's part of the scaled lattice coordinates.
// psf of 2H taps given their scaled lattice coordinates (x', y', z') two per lane-op
// exp(-a) of two values: canon_exp_neg per component
__device__ __forceinline__ f2 exp_neg2(f2 a) {
  const f2 t = -a * bc2(1.442695040888963407359924681001892137426645954152985934135449406931f);
  const f2 k = (f2){__builtin_rintf(t.x), __builtin_rintf(t.y)};
  f2 r = fma2(k, bc2(-0.693145751953125f), -a);
  r = fma2(k, bc2(-1.428606765330187045e-06f), r);
  f2 s = fma2(bc2(0.000198527617612853646278381f), r, bc2(0.00139304355252534151077271f));
  s = fma2(s, r, bc2(0.00833336077630519866943359f));
  s = fma2(s, r, bc2(0.0416664853692054748535156f));
  s = fma2(s, r, bc2(0.166666671633720397949219f));
  s = fma2(s, r, bc2(0.5f));
  s = fma2(r * r, s, r) + bc2(1.0f);
  s = (f2){ldexpf(s.x, (int)k.x), ldexpf(s.y, (int)k.y)};
  return (f2){(a.x > 87.0f) ? 0.0f : s.x, (a.y > 87.0f) ? 0.0f : s.y};
}

// The Gaussian factor exp(-z'^2 inv2s2) of a row's taps, canonical definition (oracle: canon_gauss_row).  z' is
// linear along the row, so the factor of neighbouring taps differs by a ratio that itself changes by the constant
// w = exp(-2 dd inv2s2): from the two central taps (lattice offsets 0 and 1, where |z'| is smallest for every row
// that matters) the factors are stepped outwards,
//     g(-j) = g(-j+1) * rl,  rl *= w      g(1+j) = g(j) * rr,  rr *= w,
// two multiplications per tap instead of an exponential.  g[j] = {factor of tap -j, factor of tap 1+j}.
// A row whose central factors are too small to start from (a > GAUSS_AMAX: thin slices seen obliquely, far from
// the slice plane), and every row of a slice whose first ratio could overflow (S.w < 0, set by the host), takes
// the exponential per tap instead; which rows do is part of the definition, and the oracle applies the same test.
#define GAUSS_AMAX 60.0f
// the rare path of gauss_pairs, out of line so that it costs the common path no registers
__device__ __attribute__((noinline)) f2 gauss_direct2(float dz, float rowz, float inv2s2, float j, bool keep, f2 g) {
  const f2 z = fma2(bc2(dz), (f2){-j, 1.0f + j}, bc2(rowz));
  const f2 e = exp_neg2((z * z) * bc2(inv2s2));
  return keep ? g : e;
}
struct GaussStep {        // state of the recurrence between two chunks of pairs
  f2 g, r;
  bool ok;
};
__device__ __forceinline__ GaussStep gauss_start(const RowConst S, float rowz) {
  const f2 zc = fma2(bc2(S.Lp[6]), (f2){0.0f, 1.0f}, bc2(rowz));
  const f2 ac = (zc * zc) * bc2(S.inv2s2);
  const f2 t = fma2(((f2){-2.0f, 2.0f}) * zc, bc2(S.Lp[6]), bc2(S.dd)) * bc2(S.inv2s2);
  GaussStep st;
  st.r = exp_neg2(t);
  st.g = exp_neg2(ac);
  st.ok = ac.x <= GAUSS_AMAX && ac.y <= GAUSS_AMAX && S.w >= 0.0f;
  return st;
}
// factors of pairs j0 .. j0 + H - 1
template <int H>
__device__ __forceinline__ void gauss_pairs(const RowConst S, float rowz, GaussStep &st, int j0, f2 g[H]) {
#pragma unroll
  for (int i = 0; i < H; ++i) {
    if (j0 + i > 0) {
      st.g = st.g * st.r;
      st.r = st.r * bc2(S.w);
    }
    g[i] = st.g;
  }
#ifndef SVR_NO_GAUSS_FALLBACK
  if (!__all(st.ok)) {
#pragma unroll
    for (int i = 0; i < H; ++i)
      if (j0 + i > 0) g[i] = gauss_direct2(S.Lp[6], rowz, S.inv2s2, (float)(j0 + i), st.ok, g[i]);
  }
#endif
}
// the same for the first tap (lattice offset -CENTRE) of two rows at once: x = row A, y = row B
template <int CENTRE>
__device__ __forceinline__ f2 gauss_first_tap2(const RowConst S, f2 rowz) {
  const f2 zl = fma2(bc2(S.Lp[6]), bc2(0.0f), rowz), zr = fma2(bc2(S.Lp[6]), bc2(1.0f), rowz);
  const f2 al = (zl * zl) * bc2(S.inv2s2), ar = (zr * zr) * bc2(S.inv2s2);
  const f2 t = fma2(bc2(-2.0f) * zl, bc2(S.Lp[6]), bc2(S.dd)) * bc2(S.inv2s2);
  f2 r = exp_neg2(t), g = exp_neg2(al);
#pragma unroll
  for (int j = 1; j <= CENTRE; ++j) {
    g = g * r;
    r = r * bc2(S.w);
  }
  const bool okx = al.x <= GAUSS_AMAX && ar.x <= GAUSS_AMAX && S.w >= 0.0f;
  const bool oky = al.y <= GAUSS_AMAX && ar.y <= GAUSS_AMAX && S.w >= 0.0f;
  if (!__all(okx && oky)) {
    const f2 z = fma2(bc2(S.Lp[6]), bc2((float)(-CENTRE)), rowz);
    const f2 e = exp_neg2((z * z) * bc2(S.inv2s2));
    g.x = okx ? g.x : e.x;
    g.y = oky ? g.y : e.y;
  }
  return g;
}

// in-plane factor sinc^2 of 2H taps given their scaled in-plane lattice coordinates (x', y'), two per lane-op.
// The canonical sequence (oracle: canon_rsqrt / canon_sinc) is built from fma / mul / add / rint and one integer
// shift-subtract only -- no quarter-rate transcendental, no correctly rounded sqrt or division to emulate:
//   q = x'^2 + y'^2;  y ~ 1/sqrt(q) (bit-trick start + 2 Newton-form steps with tuned constants, round 3);  r = q y;  f = r - rint(r) (exact);
//   sin(pi r)/(pi r) = +-(f y) P(f^2), P = degree-4 fit of sinc on [0, 1/4];  the square removes the sign.
// q == 0 gives NaN like sin(0)/0 in the reference (RC.cu:129); PVR takes sinc_pi's Taylor branch there instead.
#define RSQRT_MAGIC 0x5f376686u
#define RSQRT_K1 1.5009000301361084f
#define RSQRT_K2 1.5000005960464478f
template <int H, bool PVR>
__device__ __forceinline__ void eval_pairs_xyz(const RowConst S, const f2 xs[H], const f2 ys[H], f2 val2[H]) {
#define EACH for (int i = 0; i < H; ++i)
    f2 q[H], y[H], h[H], r[H], f[H], p[H];
#pragma unroll
    EACH q[i] = fma2(ys[i], ys[i], xs[i] * xs[i]);
#pragma unroll
    EACH y[i] = (f2){__uint_as_float(RSQRT_MAGIC - (__float_as_uint(q[i].x) >> 1)),
                     __uint_as_float(RSQRT_MAGIC - (__float_as_uint(q[i].y) >> 1))};
#pragma unroll
    EACH h[i] = bc2(0.5f) * q[i];
#pragma unroll
    for (int it = 0; it < 2; ++it) {                       // two steps with centred errors (RSQRT_K1, RSQRT_K2): 7.1e-7
#pragma unroll
      EACH r[i] = h[i] * y[i];
#pragma unroll
      EACH r[i] = fma2(-r[i], y[i], bc2(it == 0 ? RSQRT_K1 : RSQRT_K2));
#pragma unroll
      EACH y[i] = y[i] * r[i];
    }
#pragma unroll
    EACH r[i] = q[i] * y[i];
#pragma unroll
    EACH f[i] = r[i] - (f2){__builtin_rintf(r[i].x), __builtin_rintf(r[i].y)};
#pragma unroll
    EACH h[i] = f[i] * f[i];
#pragma unroll
    EACH p[i] = fma2(bc2(0.024719201028347015f), h[i], bc2(-0.1904420256614685f));
#pragma unroll
    EACH p[i] = fma2(p[i], h[i], bc2(0.8117148876190186f));
#pragma unroll
    EACH p[i] = fma2(p[i], h[i], bc2(-1.6449332237243652f));
#pragma unroll
    EACH p[i] = fma2(p[i], h[i], bc2(1.0f));
#pragma unroll
    EACH p[i] = (f[i] * y[i]) * p[i];
    if (PVR) {
      // sinc_pi: Taylor branch below eps^(1/4) instead of the NaN at 0 (pointSpreadFunction.cuh:45-70).  Only a tap within
      // 0.006 voxels of the PSF centre takes it: the branch (two IEEE divisions per tap) is entered when some lane has one
      // (round 3: it used to run for every tap and was a third of the patch-based kernels' instructions)
      // (the test is on q, a superset: pi sqrt(q) < 1.858e-2 needs q < 3.5e-5; q == 0 makes r NaN, which a minimum of r would skip)
      float qmin = FLT_MAX;
#pragma unroll
      EACH qmin = __builtin_fminf(__builtin_fminf(qmin, q[i].x), q[i].y);
      if (__any(!(qmin >= 4.0e-5f))) {
#pragma unroll
      EACH {
        for (int c = 0; c < 2; ++c) {
          const float x = 3.14159265359f * (c ? r[i].y : r[i].x), x2 = x * x;
          float t = 1.0f;
          if (x >= 1.1920929e-07f) {
            t -= x2 / 6.0f;
            if (x >= 3.4526698300e-04f) t += (x2 * x2) / 120.0f;
          }
          if (c) p[i].y = (x >= 1.8581361323e-02f) ? p[i].y : t;
          else p[i].x = (x >= 1.8581361323e-02f) ? p[i].x : t;
        }
      }
      }
    } else {
      // the R = 0 tap: only a pixel that sits exactly on a voxel centre of an aligned slice has one.  One v_min3_f32 per pair and a
      // branch that is never taken on real data (the empty asm keeps it one)
      float qmin = FLT_MAX;
#pragma unroll
      EACH qmin = __builtin_fminf(__builtin_fminf(qmin, q[i].x), q[i].y);
      if (__builtin_expect(__any(qmin == 0.0f), 0)) {
        asm volatile("" ::: "memory");
#pragma unroll
        EACH {
          p[i].x = (q[i].x == 0.0f) ? __builtin_nanf("") : p[i].x;
          p[i].y = (q[i].y == 0.0f) ? __builtin_nanf("") : p[i].y;
        }
      }
    }
#pragma unroll
    EACH val2[i] = p[i] * p[i];                         // si * si; the caller multiplies by the Gaussian factor
#undef EACH
}

// ZERO: skipped taps come out as -0.0f instead of -1: adding them changes nothing, and a processed tap whose value is
// exactly +0 (an underflowed Gaussian factor, sin(pi r) at an integer r) can still be told from a skipped one by its bits
// HP: pairs evaluated side by side (0 = the default below); the cell kernels ask for a whole row at once (no pixel tables, no
// tile state: they have the registers, and the longer independent stages are worth 3-4 % of their time)
// (the two halves of eval_row_t: the values of a row's taps, and the epsilon-walk over them -- a kernel that has loads to issue in
// between calls them one after the other)
template <int N, bool PVR, int HP = 0>
__device__ __forceinline__ void eval_row_vals(const RowConst S, float bx, float by, float bz, float fy, float fz, float val[N]);
template <int N, bool PVR, bool ZERO>
__device__ __forceinline__ void eps_walk(const float val[N], float out[N]) {
  float old = FLT_MAX;
#pragma unroll
  for (int x = 0; x < N; ++x) {
    const float v = val[x];
    // SVR: |d| < 0.00001 (double) == |d| <= 0.00001f; PVR: |d| < 0.00001f.  NaN compares false -> processed
    const bool skip = PVR ? (__builtin_fabsf(old - v) < PSF_EPS_F) : (__builtin_fabsf(old - v) <= PSF_EPS_F);
    out[x] = skip ? (ZERO ? -0.0f : -1.0f) : v;
    old = skip ? old : v;
  }
}
template <int N, bool PVR, bool ZERO = false, int HP = 0>
__device__ __forceinline__ void eval_row_t(const RowConst S, float bx, float by, float bz, float fy,
                                           float fz, float out[N]) {
  float val[N];
  eval_row_vals<N, PVR, HP>(S, bx, by, bz, fy, fz, val);
  eps_walk<N, PVR, ZERO>(val, out);
}
template <int N, bool PVR, int HP>
__device__ __forceinline__ void eval_row_vals(const RowConst S, float bx, float by, float bz, float fy, float fz, float val[N]) {
  // Taps are evaluated two per lane in float2 registers: every fma / mul / add stage of eval_pairs_xyz
  // compiles to v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (half the issue slots for the same cycles);
  // sqrt, rcp, rint and the selects stay per component.  Pair j holds the taps at lattice offsets -j and 1 + j
  // (the order of the Gaussian recurrence, gauss_pairs); the in-plane part is evaluated in two chunks of N/4 pairs.
  constexpr int NP = N / 2;
#ifndef SVR_EVAL_H16
#define SVR_EVAL_H16 4
#endif
  constexpr int H = HP ? HP : (N == 16 ? SVR_EVAL_H16 : NP / 2);   // pairs evaluated side by side (instruction-level parallelism against registers)
  constexpr int CENTRE = (N - 1) / 2;
  static_assert(NP % H == 0 && CENTRE == NP - 1, "pairs (-j, 1 + j) around the central taps");
  const float rowx = __builtin_fmaf(S.Lp[1], fy, __builtin_fmaf(S.Lp[2], fz, bx));
  const float rowy = __builtin_fmaf(S.Lp[4], fy, __builtin_fmaf(S.Lp[5], fz, by));
  const float rowz = __builtin_fmaf(S.Lp[7], fy, __builtin_fmaf(S.Lp[8], fz, bz));
  GaussStep st = gauss_start(S, rowz);
#pragma unroll
  for (int c0 = 0; c0 < NP; c0 += H) {
    f2 xs[H], ys[H], v2[H], g[H];
#pragma unroll
    for (int i = 0; i < H; ++i) {
      const f2 fx = (f2){(float)(-(c0 + i)), (float)(1 + c0 + i)};
      xs[i] = fma2(bc2(S.Lp[0]), fx, bc2(rowx));
      ys[i] = fma2(bc2(S.Lp[3]), fx, bc2(rowy));
    }
    eval_pairs_xyz<H, PVR>(S, xs, ys, v2);
    gauss_pairs<H>(S, rowz, st, c0, g);
#pragma unroll
    for (int i = 0; i < H; ++i) {
      const f2 v = v2[i] * g[i];                        // (si * si) * gz
      val[CENTRE - (c0 + i)] = v.x;
      val[CENTRE + 1 + c0 + i] = v.y;
    }
  }
}
// Can every tap of the (fy, fz) row be proven to lie below the epsilon of the skip test?  Then the
// reference processes only the row's first tap (oldPSF starts at FLT_MAX) and skips the other 15:
// |v0 - v| <= max(v0, v) < 1e-5.  Upper bound of the canonical value over the row:
//   exp(-a_min) * min(1, 1 / (pi^2 q_min)),
// a_min from the end taps (z' is monotone in x, and so is its rounded fma), q_min from the vertex of the
// quadratic x'^2 + y'^2 shrunk by 1 % (covers the roundings of x', y', q), both with the polynomial errors
// of the canonical sin / exp (< 1e-6) far inside the 2 % margin of the threshold.  A row that may contain
// the R = 0 tap (NaN, RC.cu:129) is never declared dead.
__host__ __device__ __forceinline__ bool row_is_dead(const RowConst S, float bx, float by, float bz, float fy, float fz) {
  const float rowx = __builtin_fmaf(S.Lp[1], fy, __builtin_fmaf(S.Lp[2], fz, bx));
  const float rowy = __builtin_fmaf(S.Lp[4], fy, __builtin_fmaf(S.Lp[5], fz, by));
  const float rowz = __builtin_fmaf(S.Lp[7], fy, __builtin_fmaf(S.Lp[8], fz, bz));
  const float z0 = __builtin_fmaf(S.Lp[6], (float)(-PSF_CENTRE), rowz);
  const float z1 = __builtin_fmaf(S.Lp[6], (float)(PSF_SUPPORT - 1 - PSF_CENTRE), rowz);
  const float zmin = (z0 > 0.0f) == (z1 > 0.0f) && z0 != 0.0f && z1 != 0.0f ? fminf(fabsf(z0), fabsf(z1)) : 0.0f;
  const float amin = (zmin * zmin) * S.inv2s2;
  const float qa = S.Lp[0] * S.Lp[0] + S.Lp[3] * S.Lp[3];
  const float qb = 2.0f * (S.Lp[0] * rowx + S.Lp[3] * rowy);
  const float qc = rowx * rowx + rowy * rowy;
  float fx = qa > 1e-12f ? -qb / (2.0f * qa) : 0.0f;
  fx = fminf(fmaxf(fx, (float)(-PSF_CENTRE)), (float)(PSF_SUPPORT - 1 - PSF_CENTRE));
  const float qe0 = (qa * (float)(PSF_CENTRE * PSF_CENTRE) - qb * (float)PSF_CENTRE) + qc;
  const float e1 = (float)(PSF_SUPPORT - 1 - PSF_CENTRE);
  const float qe1 = (qa * (e1 * e1) + qb * e1) + qc;
  const float qv = (qa * (fx * fx) + qb * fx) + qc;
  const float qmin = fmaxf(fminf(qv, fminf(qe0, qe1)), 0.0f) * 0.99f - 1e-5f;
  if (!(qmin > 1e-6f)) return false;   // may contain the R = 0 tap: keep the row live
  // ln(1e5 / 0.98) = 11.533
#ifndef SVR_DEAD_THR
#define SVR_DEAD_THR 11.56f
#endif
  return amin + logf(fmaxf(1.0f, 9.8696044f * qmin)) > SVR_DEAD_THR;
}

__device__ __forceinline__ void eval_row_at(const RowConst S, float bx, float by, float bz, float fy,
                                            float fz, float out[16]) {
  eval_row_t<16, false>(S, bx, by, bz, fy, fz, out);
}
// Phase 1 of the wave-per-pixel kernels: lane = one (y,z) row of quarter q (4 z-planes x 16 y)
__device__ __forceinline__ void eval_row(const RowConst S, const PixelState &P, int lane, int q,
                                         float out[16]) {
  eval_row_at(S, P.bx, P.by, P.bz, (float)((lane & 15) - PSF_CENTRE), (float)(4 * q + (lane >> 4) - PSF_CENTRE), out);
}

__device__ __forceinline__ uint32_t sat0(int i) { return (uint32_t)max(i, 0); }  // float->uint saturation

template <int MODE>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void psf_kernel(PsfArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[WAVES_PER_BLOCK][64 * LDS_ROW];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const uint32_t pi = blockIdx.x * WAVES_PER_BLOCK + wave;
  if (pi >= a.n) return;
  const uint32_t idx = __builtin_amdgcn_readfirstlane(a.list[pi]);
  const uint32_t n2 = (uint32_t)(a.sx * a.sy);
  const uint32_t sl = idx / n2;
  const uint32_t rem = idx - sl * n2;
  const int py = (int)(rem / (uint32_t)a.sx);
  const int px = (int)(rem - (uint32_t)py * (uint32_t)a.sx);
  const SliceConst &S = a.sc[sl];
  const VolGeom &vg = a.vg;
  const PixelState P = pixel_setup(S, vg, px, py);
  const RowConst RC = load_row_const(S);
  float *my = lds[wave];

  float s = a.slices[idx];
  float sume;
  if (MODE == MODE_GAUSS) {
    // pass 1 (RC.cu:228-258): sume over processed, in-bounds taps (truncating cast, no mask)
    double acc = 0.0;
    for (int q = 0; q < 4; ++q) {
      float out[16];
      eval_row(RC, P, lane, q, out);
      const uint32_t az = sat0(P.czi + 4 * q + (lane >> 4) - PSF_CENTRE);
      const uint32_t ay = sat0(P.cyi + (lane & 15) - PSF_CENTRE);
      const bool rowin = az < (uint32_t)vg.vz && ay < (uint32_t)vg.vy;
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const uint32_t ax = sat0(P.cxi + x - PSF_CENTRE);
        const bool use = rowin && ax < (uint32_t)vg.vx && !(out[x] < 0.0f);
        acc += use ? (double)out[x] : 0.0;
      }
    }
    sume = (float)wave_sum(acc);
    if (!(sume > 0.5f)) return;          // also drops NaN (RC.cu:251-258)
    if (lane == 0) a.psf_sums[idx] = sume;
    s = a.bias ? s * expf(-a.bias[idx]) * a.scales[sl] : s * a.scales[sl];   // RC.cu:200-203
  } else {
    sume = a.psf_sums[idx];
  }
  if (MODE == MODE_BIAS) {
    // normalizeBiasKernel3D_tex RC.cu:544-550: the scattered value is the pixel's log bias
    s = a.bias[idx];
    const float scale = a.scales[sl];
    if (scale > 0) s -= logf(scale);
  }

  float f0 = 0.0f, f1 = 0.0f;   // FWD: sim, weight partials.  BACK: (w*sw*e)/sume, (w*sw)/sume
  if (MODE == MODE_BACK) {
    float w = a.weights[idx];
    float ss = a.simslices[idx];
    float e = a.bias ? s * expf(-a.bias[idx]) * a.scales[sl] : s * a.scales[sl];   // RC.cu:439-442
    e = (ss > 0.0f) ? (e - ss) : 0.0f;   // RC.cu:444-447
    float ws = w * a.slice_weights[sl];
    f1 = ws / sume;
    f0 = f1 * e;
  }
  const float rsume = 1.0f / sume;
  bool hit = false;
  const int x2 = lane & 15, r4 = lane >> 4;
  const uint32_t ax = sat0(P.cxi + x2 - PSF_CENTRE);
  const bool xin = ax < (uint32_t)vg.vx;

  for (int q = 0; q < 4; ++q) {
    {
      float out[16];
      eval_row(RC, P, lane, q, out);
      float4 *dst = reinterpret_cast<float4 *>(my + lane * LDS_ROW);
      dst[0] = make_float4(out[0], out[1], out[2], out[3]);
      dst[1] = make_float4(out[4], out[5], out[6], out[7]);
      dst[2] = make_float4(out[8], out[9], out[10], out[11]);
      dst[3] = make_float4(out[12], out[13], out[14], out[15]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // Phase 2: lane = (x, 4 consecutive y rows): coalesced 4 x 64 B volume accesses
    for (int j = 0; j < 16; ++j) {
      const int row = 4 * j + r4;
      const float val = my[row * LDS_ROW + x2];
      const uint32_t az = sat0(P.czi + 4 * q + (row >> 4) - PSF_CENTRE);
      const uint32_t ay = sat0(P.cyi + (row & 15) - PSF_CENTRE);
      const bool ok = xin && ay < (uint32_t)vg.vy && az < (uint32_t)vg.vz && !(val < 0.0f);
      if (ok) {
        const uint32_t vi = ax + ay * (uint32_t)vg.vx + az * (uint32_t)(vg.vx * vg.vy);
        if (a.mask[vi] != 0.0f) {
          if (MODE == MODE_GAUSS || MODE == MODE_BIAS) {
            float p = val / sume;                       // RC.cu:278, 591
            unsafeAtomicAdd(a.volw + vi, p);
            unsafeAtomicAdd(a.recon + vi, p * s);
          } else if (MODE == MODE_FWD) {
            float p = val * rsume;
            f0 += p * a.vol[vi];
            f1 += p;
          } else {
            unsafeAtomicAdd(a.addon + vi, val * f0);
            unsafeAtomicAdd(a.cmap + vi, val * f1);
          }
          hit = true;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  if (MODE == MODE_GAUSS) {
    if (__ballot(hit) != 0ull && lane == 0) a.voxcount[idx] = 1;   // RC.cu:291-294
  } else if (MODE == MODE_FWD) {
    float sim = wave_sum(f0), w = wave_sum(f1);
    bool inside = __ballot(hit) != 0ull;
    if (lane == 0 && w > 0.0f) {                                    // RC.cu:398-403
      a.simslices[idx] = sim / w;
      a.simweights[idx] = w;
      a.siminside[idx] = inside ? 1 : 0;
    }
  }
}


// ------------------------------------------------------------------------------------------
// LDS-tiled back-projection (SuperresolutionKernel3D_tex RC.cu:408-522)
// ------------------------------------------------------------------------------------------
// One workgroup owns a TW x TH tile of slice pixels.  The union of their 16^3 footprints (after
// the float->uint saturation) is a small box of the volume; {addon, cmap} for that box live in
// LDS, every wave scatters its pixels' taps with ds_add_f32 straight from the row-per-lane
// layout (no transposition needed: LDS has no coalescing rule), and the box is flushed to HBM
// once with the mask test applied per voxel instead of per tap.  This removes ~95 % of the
// device-scope float atomics of the direct kernel, which are what bound it (profiles/r01_a).
#define TILE_WAVES 16

struct TileArgs {
  const uint32_t *tiles;   // tile ids: (slice * tiles_y + ty) * tiles_x + tx
  uint32_t ntiles;
  int tiles_x, tiles_y;
  int cap;                 // voxels of LDS accumulator available
  int dbg;                 // dev experiments only (0 = production)
  int tw, th;              // tile size in pixels, tw * th <= 64
  int gauss;               // scatter kernels: 1 = Gaussian-reconstruction pass 2 (recon|volw instead of addon|cmap)
};

// activity of a pixel for the tile kernels: s != -1 and (flag ? flag : v_PSF_sums != 0); both NULL: s != -1
__device__ __forceinline__ bool pixel_active(const float *slices, const float *psf_sums, const unsigned char *flag,
                                             size_t idx) {
  if (slices[idx] == -1.0f) return false;
  if (flag) return flag[idx] != 0;
  return psf_sums ? psf_sums[idx] != 0.0f : true;
}

__global__ void k_build_tiles(const float *slices, const float *psf_sums, const unsigned char *flag, int sx, int sy,
                              int ns, int tiles_x, int tiles_y, int TILE_W, int TILE_H, uint32_t *tiles,
                              uint32_t *counter, const unsigned char *slice_sel = nullptr, int want = 0, uint32_t t0 = 0u,
                              uint32_t t_end = 0xFFFFFFFFu) {
  const int lane = threadIdx.x & 63;
  const uint32_t t = t0 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // (t0 .. t_end: this dispatch's piece of the candidates)
  const uint32_t total = min((uint32_t)tiles_x * tiles_y * ns, t_end);
  const uint32_t tt = t < total ? t : 0u;                  // (every wavefront reaches the barriers below)
  const int sl = tt / (tiles_x * tiles_y);
  const bool skip = slice_sel && slice_sel[sl] != want;
  const int r = tt - sl * tiles_x * tiles_y;
  const int ty = r / tiles_x, tx = r - ty * tiles_x;
  const int px = tx * TILE_W + (lane % TILE_W), py = ty * TILE_H + (lane / TILE_W);
  bool act = false;
  if (lane < TILE_W * TILE_H && px < sx && py < sy) {
    size_t idx = (size_t)px + (size_t)py * sx + (size_t)sl * sx * sy;
    act = pixel_active(slices, psf_sums, flag, idx);
  }
  // one atomic on the global counter per workgroup, not per tile (168 k tiles on one address took 1.2 ms on P4)
  __shared__ uint32_t sh_n, sh_base, sh_t[16];
  if (threadIdx.x == 0) sh_n = 0;
  __syncthreads();
  const bool keep = t < total && !skip && __ballot(act) != 0ull;
  if (keep && lane == 0) sh_t[atomicAdd(&sh_n, 1u)] = t;
  __syncthreads();
  if (threadIdx.x == 0 && sh_n) sh_base = atomicAdd(counter, sh_n);
  __syncthreads();
  if (threadIdx.x < sh_n) tiles[sh_base + threadIdx.x] = sh_t[threadIdx.x];
}

// XCD-aware tile index.  Workgroups are handed to the eight XCDs round-robin by their index and every XCD has its own L2.
// The tile lists are (roughly) in raster order, so with the plain index an XCD sees every eighth tile and shares no box
// with its own previous tile; with this permutation every XCD takes runs of SVR_XCD_RUN neighbouring tiles, whose boxes of
// the volume overlap, while the eight XCDs still advance through the list together (measured on the gather of P4:
// 4.60 -> 4.47 ms with runs of 16; a coarser split -- whole slices per XCD -- lost to load imbalance).
#ifndef SVR_XCD_RUN
#define SVR_XCD_RUN 16
#endif
__device__ __forceinline__ uint32_t xcd_run_index(uint32_t b, uint32_t n) {
  if (SVR_XCD_RUN <= 0) return b;
  constexpr uint32_t R = SVR_XCD_RUN > 0 ? SVR_XCD_RUN : 1, G = 8u * R;
  const uint32_t base = b / G * G;
  if (base + G > n) return b;                             // the last, incomplete group keeps its order
  const uint32_t q = b - base;
  return base + (q & 7u) * R + (q >> 3);
}


struct PixelRec {      // per tile pixel, in LDS
  int cx, cy, cz;
  float bx, by, bz;
  float f0, f1;
};

#define SLOT_MAXP 48      // planes of a box the plane tables are sized for
struct RowWalk {       // i = y * P + x walked in steps of `stride` without a division per element
  int x, y, tx, ty, P;
  __device__ __forceinline__ void init(int i0, int stride, int P_) {
    P = P_;
    y = i0 / P; x = i0 - y * P;
    ty = stride / P; tx = stride - ty * P;
  }
  __device__ __forceinline__ void step() {
    x += tx; y += ty;
    if (x >= P) { x -= P; ++y; }
  }
};

// Is every tap of the unit (all x, all offsets of the lane axis, fixed offset `fo` of volume axis F in {1, 2}) provably
// below the epsilon of the skip test, and free of the R = 0 tap?  |z'| is bounded from below over the unit's lattice
// square (z' is affine), exp(-a_min) with sinc^2 <= 1 bounds the value; the R = 0 tap (NaN, RC.cu:129 -- it would be
// processed, and so would every tap after it in its row) is excluded by evaluating q at the lattice points around the
// real solution of x' = y' = 0.  A degenerate in-plane map (rows along the slice normal) never takes the shortcut.
__device__ __forceinline__ bool unit_is_dead(const RowConst S, float bx, float by, float bz, int F, float fo) {
  const bool f1 = F == 1;                               // (selects, not indexing: an index would put RowConst into scratch)
  const float LxF = f1 ? S.Lp[1] : S.Lp[2], LxG = f1 ? S.Lp[2] : S.Lp[1];
  const float LyF = f1 ? S.Lp[4] : S.Lp[5], LyG = f1 ? S.Lp[5] : S.Lp[4];
  const float LzF = f1 ? S.Lp[7] : S.Lp[8], LzG = f1 ? S.Lp[8] : S.Lp[7];
  const float lo_ = (float)(-PSF_CENTRE), hi_ = (float)(PSF_SUPPORT - 1 - PSF_CENTRE);
  const float c = __builtin_fmaf(LzF, fo, bz);
  const float zlo = c + fminf(S.Lp[6] * lo_, S.Lp[6] * hi_) + fminf(LzG * lo_, LzG * hi_);
  const float zhi = c + fmaxf(S.Lp[6] * lo_, S.Lp[6] * hi_) + fmaxf(LzG * lo_, LzG * hi_);
  if (!(zlo > 0.0f || zhi < 0.0f)) return false;
  const float zmin = fminf(fabsf(zlo), fabsf(zhi)) * 0.999f - 1e-4f;      // covers the roundings of the fma chains of z'
  if (!(zmin > 0.0f && (zmin * zmin) * S.inv2s2 > SVR_DEAD_THR)) return false;
  if (S.invD == 0.0f) return false;                     // degenerate in-plane map: no shortcut
  const float cx = __builtin_fmaf(LxF, fo, bx), cy = __builtin_fmaf(LyF, fo, by);
  const float ox = (cy * LxG - cx * LyG) * S.invD, og = (cx * S.Lp[3] - cy * S.Lp[0]) * S.invD;
  if (!(fabsf(ox) < 64.0f && fabsf(og) < 64.0f)) return true;             // the zero of (x', y') is far outside the lattice
  const float fx0 = floorf(ox), fg0 = floorf(og);
  bool zero = false;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float fx = fx0 + (float)(j & 1), fg = fg0 + (float)(j >> 1);
    const float fy = f1 ? fo : fg, fz = f1 ? fg : fo;
    const float xs = __builtin_fmaf(S.Lp[0], fx, __builtin_fmaf(S.Lp[1], fy, __builtin_fmaf(S.Lp[2], fz, bx)));
    const float ys = __builtin_fmaf(S.Lp[3], fx, __builtin_fmaf(S.Lp[4], fy, __builtin_fmaf(S.Lp[5], fz, by)));
    zero = zero || __builtin_fmaf(ys, ys, xs * xs) == 0.0f;
  }
  return !zero;
}

// 16 bytes of the coefficient table: read once per pass and never again before the next one, so the load is marked
// non-temporal (it should not push the volume, the mask and the slice data out of the L2 / the memory-side cache)
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_stream(const float4 *p) {
  const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

// 16 bytes per lane from global memory straight into the LDS (LDS-DMA, gfx950: global_load_lds_dwordx4): lane l's 16 bytes land at
// lds_dst + 16 l (lds_dst: wave-uniform LDS byte address).  No VGPR is the destination, so nothing the compiler can copy, spill or wait for:
// the data is in flight for as long as the kernel likes, and the kernel counts its completion itself (glds_wait) before it reads the LDS.
// M0 holds the destination and is compiler-reserved: saved and restored inside the statement.  (hipcc does not count this load in its
// own s_waitcnt bookkeeping; an uncounted older operation only makes the compiler's counted waits stricter, never looser: vmcnt is in order.)
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// wait until at most N vector-memory operations of this wavefront are outstanding (N an immediate), then the LDS may be read
template <int N>
__device__ __forceinline__ void glds_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ uint32_t lds_addr_of(const void *p) { return (uint32_t)(uintptr_t)p; }   // (a generic pointer into the LDS: aperture in the high half, LDS byte address in the low)

#ifndef SVR_COEFF_STORE_NT
#define SVR_COEFF_STORE_NT 1
#endif
__device__ __forceinline__ void store_stream(float4 *p, float a, float b, float c, float d) {
#if SVR_COEFF_STORE_NT
  __builtin_nontemporal_store((nt_f4){a, b, c, d}, reinterpret_cast<nt_f4 *>(p));
#else
  *p = make_float4(a, b, c, d);
#endif
}
// The coefficient table (svr_ctx::d_coeff): one wavefront per PSF pixel, four of its NS units per pass -- slot = unit,
// lane = row, exactly the decomposition of the scatter and the gather.  A unit's NS x NS taps (skipped ones as -0.0f) go
// out as NS/4 x 16 float4: [tap quad q][row y], so that the 16 lanes of a slot write and later read 256 contiguous bytes
// per instruction (SVR: 16 units of 1 KiB per pixel; PVR, support 12: 12 units of 768 bytes).  Dead units (unit_is_dead) are not stored: the kernels evaluate their first taps themselves.
template <int NS, bool PVR>
__global__ __launch_bounds__(256) void k_coeff_build(PsfArgs a, float4 *coeff, uint32_t *coeff_id, uint32_t base) {
  constexpr int NC = (NS - 1) / 2, QUADS = NS / 4;
  static_assert(NS % 4 == 0 && NS <= 16, "tap quads, one row per lane of a 16-lane slot");
  const uint32_t wl = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (wl >= a.n) return;
  const int lane = threadIdx.x & 63;
  const uint32_t idx = a.list[wl];
  const uint32_t w = base + wl;                          // the pixel's id in the table (the list goes out in pieces)
  const uint32_t n2 = (uint32_t)(a.sx * a.sy);
  const uint32_t sl = idx / n2, rem = idx - sl * n2;
  const int py = (int)(rem / (uint32_t)a.sx), px = (int)(rem - (uint32_t)py * (uint32_t)a.sx);
  const SliceConst &S = a.sc[sl];
  const PixelState P = pixel_setup(S, a.vg, px, py);
  const RowConst RC = load_row_const(S);
  const bool swap = S.own == 1;
  const int F = swap ? 1 : 2;
  if (lane == 0) coeff_id[idx] = w;
  const int slot = lane >> 4, y = lane & 15;
  const float fyl = (float)(y - NC);
  for (int g = 0; 4 * g < NS; ++g) {
    const int u = 4 * g + slot;
    const float fu = (float)(u - NC);
    const bool dead = !PVR && unit_is_dead(RC, P.bx, P.by, P.bz, F, fu);   // (the bound is derived for the SVR constants)
    if (__all(dead)) continue;
    float out[NS];
    eval_row_t<NS, PVR, true>(RC, P.bx, P.by, P.bz, swap ? fu : fyl, swap ? fyl : fu, out);
    if (!dead && u < NS && y < NS) {
      float4 *dst = coeff + ((size_t)w * NS + u) * (QUADS * 16) + y;
#pragma unroll
      for (int q = 0; q < QUADS; ++q) store_stream(dst + q * 16, out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
    }
  }
}



// getReconValueFromTexture (R2/reconVolume.cu:170-187): linear filter at the un-offset coordinate =
// 0.125 * sum over {p-1,p}^3 with zero border
__device__ __forceinline__ float pvr_tex(const float *vol, const VolGeom &vg, int X, int Y, int Z) {
  float v = 0.0f;
#pragma unroll
  for (int dz = -1; dz <= 0; ++dz)
#pragma unroll
    for (int dy = -1; dy <= 0; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 0; ++dx) {
        const int x = X + dx, y = Y + dy, z = Z + dz;
        const float t = (x >= 0 && y >= 0 && z >= 0) ? vol[(size_t)x + (size_t)y * vg.vx + (size_t)z * vg.vx * vg.vy] : 0.0f;
        v += 0.125f * t;
      }
  return v;
}

// Sum over the 16 lanes of a slot, result in the slot's first lane: four DPP row shifts folded into the additions
// (v_add_f32_dpp) instead of four dependent ds_bpermute round trips through the LDS crossbar.  Lanes shifted in from
// beyond the row read 0 (bound_ctrl), so the tree is ((v0 + v8) + (v4 + v12)) + ... -- fixed, the same for every unit.
template <int N> __device__ __forceinline__ int row_shl_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x100 + N, 0xF, 0xF, true);
}
template <int N> __device__ __forceinline__ float row_shl_f(float v) {
  return __builtin_bit_cast(float, row_shl_i<N>(__builtin_bit_cast(int, v)));
}
template <int N> __device__ __forceinline__ double row_shl_d(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)row_shl_i<N>((int)(unsigned)b), hi = (unsigned)row_shl_i<N>((int)(unsigned)(b >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float row_sum16(float v) {
  v += row_shl_f<8>(v); v += row_shl_f<4>(v); v += row_shl_f<2>(v); v += row_shl_f<1>(v);
  return v;
}
__device__ __forceinline__ double row_sum16(double v) {
  v += row_shl_d<8>(v); v += row_shl_d<4>(v); v += row_shl_d<2>(v); v += row_shl_d<1>(v);
  return v;
}
#include "svr_tile.inc"    // the tile kernels of rounds 1-2: back_tiled_kernel, back_slot_kernel, back_wave_kernel (back_mode 1 / 3 / 4), fwd_unit_kernel (fwd_mode 1)

// ------------------------------------------------------------------------------------------
// Patch-to-volume (PVR) PSF kernels -- first correct version
// ------------------------------------------------------------------------------------------
// patchBasedPSFReconstructionKernel (R2/patchBasedPSFReconstruction_gpu.cu:41-145),
// patchBasedSimulatePatchesKernel (R2/patchBasedSimulatePatches_gpu.cu:41-125),
// patchBasedSuperresolution_gpuKernel (R2/patchBasedSuperresolution_gpu.cu:34-111).
// Patches are the "slices" of the padded grid [nPatches][pY][pX]; support 12^3, sigma_z = dim.z,
// through-plane offset / 2.5, sinc_pi, pixel kept if sume > 1e-5 or NaN, optional superpixel mask.
// One wavefront per patch pixel; lane = one of the 144 (y,z) rows (3 passes), 12 x-taps each, the
// volume is accessed straight from the row-per-lane layout (not yet tiled like the SVR scatter).
#define PVR_N 12
#define PVR_CENTRE 5

#define MODE_GAUSS2 4      // pass 2 of the Gaussian reconstruction only (v_PSF_sums from the tiled pass 1)
template <int MODE>
__device__ __forceinline__ void pvr_pixel(const PsfArgs &a, uint32_t idx, int lane) {
  const uint32_t n2 = (uint32_t)(a.sx * a.sy);
  const uint32_t sl = idx / n2;
  const uint32_t rem = idx - sl * n2;
  const int py = (int)(rem / (uint32_t)a.sx);
  const int px = (int)(rem - (uint32_t)py * (uint32_t)a.sx);
  const SliceConst &S = a.sc[sl];
  const VolGeom &vg = a.vg;
  const PixelState P = pixel_setup(S, vg, px, py);
  const RowConst RC = load_row_const(S);
  float s = a.slices[idx] * a.scales[sl];                 // s * patch.scale
  float sume;
  if (MODE == MODE_GAUSS) {
    double acc = 0.0;
    const bool spx_ok = !a.spx || a.spx[(size_t)sl * 4096 + px + 64 * py] == '1';
    if (spx_ok) {
      for (int p = 0; p < 3; ++p) {
        const int r = p * 64 + lane;
        const bool valid = r < PVR_N * PVR_N;
        const int z = r / PVR_N, y = r - z * PVR_N;
        float out[PVR_N];
        eval_row_t<PVR_N, true>(RC, P.bx, P.by, P.bz, (float)(y - PVR_CENTRE), (float)(z - PVR_CENTRE), out);
        const uint32_t az = sat0(P.czi + z - PVR_CENTRE), ay = sat0(P.cyi + y - PVR_CENTRE);
        const bool rowin = valid && az < (uint32_t)vg.vz && ay < (uint32_t)vg.vy;
#pragma unroll
        for (int x = 0; x < PVR_N; ++x) {
          const uint32_t ax = sat0(P.cxi + x - PVR_CENTRE);
          acc += (rowin && ax < (uint32_t)vg.vx && !(out[x] < 0.0f)) ? (double)out[x] : 0.0;
        }
      }
    }
    sume = (float)wave_sum(acc);
    if (!((sume > 0.00001f) || (sume != sume))) return;   // patchBasedPSFReconstruction_gpu.cu:110
    if (lane == 0) a.psf_sums[idx] = sume;
  } else {
    sume = a.psf_sums[idx];
  }
  float f0 = 0.0f, f1 = 0.0f;
  if (MODE == MODE_BACK) {
    const float ss = a.simslices[idx];
    const float e = (ss > 0.0f) ? (s - ss) : 0.0f;         // patchBasedSuperresolution_gpu.cu:64-67
    f1 = a.weights[idx] * a.slice_weights[sl];             // w * patch_weight
    f0 = f1 * e;
  }
  bool hit = false;
  for (int p = 0; p < 3; ++p) {
    const int r = p * 64 + lane;
    const bool valid = r < PVR_N * PVR_N;
    const int z = r / PVR_N, y = r - z * PVR_N;
    float out[PVR_N];
    eval_row_t<PVR_N, true>(RC, P.bx, P.by, P.bz, (float)(y - PVR_CENTRE), (float)(z - PVR_CENTRE), out);
    const uint32_t az = sat0(P.czi + z - PVR_CENTRE), ay = sat0(P.cyi + y - PVR_CENTRE);
    const bool rowin = valid && az < (uint32_t)vg.vz && ay < (uint32_t)vg.vy;
    if (!rowin) continue;
#pragma unroll
    for (int x = 0; x < PVR_N; ++x) {
      const uint32_t ax = sat0(P.cxi + x - PVR_CENTRE);
      if (ax < (uint32_t)vg.vx && !(out[x] < 0.0f)) {
        const uint32_t vi = ax + ay * (uint32_t)vg.vx + az * (uint32_t)(vg.vx * vg.vy);
        if (a.mask[vi] != 0.0f) {
          const float pv = out[x] / sume;
          if (MODE == MODE_GAUSS || MODE == MODE_GAUSS2) {
            unsafeAtomicAdd(a.volw + vi, pv);
            unsafeAtomicAdd(a.recon + vi, s * pv);
          } else if (MODE == MODE_FWD) {
            f0 += pv * pvr_tex(a.vol, vg, (int)ax, (int)ay, (int)az);
            f1 += pv;
          } else {
            unsafeAtomicAdd(a.addon + vi, pv * f0);
            unsafeAtomicAdd(a.cmap + vi, pv * f1);
          }
          hit = true;
        }
      }
    }
  }
  if (MODE == MODE_GAUSS) {
    if (__ballot(hit) != 0ull && lane == 0) a.voxcount[idx] = 1;
  } else if (MODE == MODE_FWD) {
    const float sim = wave_sum(f0), w = wave_sum(f1);
    const bool inside = __ballot(hit) != 0ull;
    if (lane == 0 && w > 0.0f) {
      a.simslices[idx] = sim / w;
      a.simweights[idx] = w;
      a.siminside[idx] = inside ? 1 : 0;
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void pvr_kernel(PsfArgs a) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const uint32_t pi = blockIdx.x * WAVES_PER_BLOCK + wave;
  if (pi >= a.n) return;
  pvr_pixel<MODE>(a, __builtin_amdgcn_readfirstlane(a.list[pi]), lane);
}

// fallback of the tiled PVR scatter: the pixels of the listed tiles with device atomics (MODE_BACK / MODE_GAUSS2)
template <int MODE>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void pvr_tiles_kernel(PsfArgs a, TileArgs ta) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const uint32_t t = ta.tiles[blockIdx.x];
  const int per_slice = ta.tiles_x * ta.tiles_y;
  const uint32_t sl = t / per_slice;
  const int r = t - sl * per_slice;
  const int ty = r / ta.tiles_x, tx = r - ty * ta.tiles_x;
  for (int k = wave; k < ta.tw * ta.th; k += WAVES_PER_BLOCK) {
    const int px = tx * ta.tw + k % ta.tw, py = ty * ta.th + k / ta.tw;
    if (px >= a.sx || py >= a.sy) continue;
    const uint32_t idx = (uint32_t)px + (uint32_t)py * a.sx + sl * (uint32_t)(a.sx * a.sy);
    if (pixel_active(a.slices, a.psf_sums, a.flag, idx)) pvr_pixel<MODE>(a, idx, lane);
  }
}

// test probe: one wave evaluates one pixel; out[x + 16*y + 256*z] = psf or -1 when skipped
__global__ __launch_bounds__(64) void k_probe_pixel(PsfArgs a, uint32_t idx, float *out, int *centre) {
  const int lane = threadIdx.x & 63;
  const uint32_t n2 = (uint32_t)(a.sx * a.sy);
  const uint32_t sl = idx / n2;
  const uint32_t rem = idx - sl * n2;
  const int py = (int)(rem / (uint32_t)a.sx);
  const int px = (int)(rem - (uint32_t)py * (uint32_t)a.sx);
  const SliceConst &S = a.sc[sl];
  const PixelState P = pixel_setup(S, a.vg, px, py);
  const RowConst RC = load_row_const(S);
  if (a.vg.pvr) {
    for (int p = 0; p < 3; ++p) {
      const int r = p * 64 + lane;
      if (r >= PVR_N * PVR_N) continue;
      const int z = r / PVR_N, y = r - z * PVR_N;
      float v[PVR_N];
      eval_row_t<PVR_N, true>(RC, P.bx, P.by, P.bz, (float)(y - PVR_CENTRE), (float)(z - PVR_CENTRE), v);
#pragma unroll
      for (int x = 0; x < PVR_N; ++x) out[x + 16 * y + 256 * z] = v[x];
    }
  } else {
    for (int q = 0; q < 4; ++q) {
      float v[16];
      eval_row(RC, P, lane, q, v);
      const int z = 4 * q + (lane >> 4), y = lane & 15;
#pragma unroll
      for (int x = 0; x < 16; ++x) out[x + 16 * y + 256 * z] = v[x];
    }
  }
  if (lane == 0) { centre[0] = P.cxi; centre[1] = P.cyi; centre[2] = P.czi; }
}

#include "svr_small.inc"   // list compaction, reductions, the EM / scale / bias / volume kernels, the NCC cost of the IRTK registration

}  // namespace

// ==========================================================================================
// context
// ==========================================================================================
struct RegState;   // GPU slice-to-volume registration state (svr_reg.inc)
namespace { struct CellState; struct SlabPlan; struct SliceEm; void slice_em_free(SliceEm *); void slice_em_invalidate(SliceEm *); }  // sorted pixels, runs, items and staging of the scatter without atomics (svr_cell.inc)

struct svr_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  bool disable_bias = true, debug_gpu = false;

  // volume
  uint32_t vx = 0, vy = 0, vz = 0;
  float vdim[3] = {1, 1, 1};
  size_t nv = 0;
  float *d_recon_volw = nullptr;   // recon | volw
  float *d_addon_cmap = nullptr;   // addon | cmap
  float *d_mask = nullptr, *d_snap = nullptr, *d_recon_new = nullptr;
  float *recon_cur = nullptr;   // the volume: d_recon_volw or d_recon_new (the fused volume update writes the other one, svr_regul.inc)
  int reg_tile = -1;            // k_regul_fused's tile: -1 = by the volume's extent, 0 = 64 x 8, 1 = 32 x 16, 2 = 32 x 8
  int reg_mode = 1;             // volume update: 1 = k_regul_fused (one kernel, LDS planes, float32 weights), 0 = k_reg_prep + k_regularize (fp64 weights)
  bool prep_pending = false;    // reg_mode 1, non-adaptive: Prep's own changes of addon / cmap are applied when somebody reads the buffers
  bool cmap_from_scatter = false;   // cmap is what the scatter wrote (zero outside the mask): the update skips tiles outside the mask's box
  float2 *d_volm = nullptr;   // {V m, m}, refreshed before every SVR forward projection
  bool have_mask = false;
  // bounding box of mask != 0 (inclusive), from the host copy handed to svr_set_mask: the scatter only ever writes mask voxels,
  // so a sharded run need not all-reduce the rest of a volume pair (svr_pair_pack / svr_pair_unpack)
  int mbox_lo[3] = {0, 0, 0}, mbox_hi[3] = {-1, -1, -1};
  bool mbox_valid = false;
  float *d_pair_pack = nullptr;
  size_t pair_pack_cap = 0;

  // slice grid
  uint32_t sx = 0, sy = 0, ns = 0;
  size_t np = 0;
  float *d_slices = nullptr, *d_weights = nullptr, *d_simslices = nullptr, *d_simweights = nullptr,
        *d_psf_sums = nullptr;
  unsigned char *d_siminside = nullptr;
  int *d_voxcount = nullptr;
  bool have_slices = false;
  bool sem_weights_current = false;     // d_slice_weights holds the device-side EM's weights (svr_slice_em_run) and nobody has written it since
  float *d_scales = nullptr, *d_slice_weights = nullptr, *d_scales_host_copy = nullptr,
        *d_tmp_ns = nullptr;
  unsigned char *d_slice_inside = nullptr;
  std::vector<float> h_slice_weights;             // RC.cu:1346.  h_scales (RC.cu:1345) lives on the device: d_scales_host_copy
  bool have_scales = false;
  // per-slice vectors go up through pinned slots without a stream synchronisation (upload_ns); what each device vector holds
  // is mirrored on the host, and a vector that is already there is not sent again (the SR loop sends the slice weights twice
  // and the previous scale vector once per iteration: RC.cu:2123, 3238)
  static constexpr int UP_SLOTS = 4;
  float *h_up = nullptr;                          // [UP_SLOTS][ns], pinned
  hipEvent_t up_ev[UP_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
  bool up_busy[UP_SLOTS] = {false, false, false, false};
  int up_next = 0;
  std::vector<float> mir_scales, mir_slice_weights, mir_scales_copy;
  // small results come down through one pinned arena: the copies are queued on the stream, ONE synchronisation, then they
  // are handed to the caller's memory (down_queue / down_flush)
  unsigned char *h_down = nullptr;
  size_t down_cap = 0, down_used = 0;
  struct DownItem { void *dst; size_t off, bytes; };
  std::vector<DownItem> down_items;
  float *d_em = nullptr;                          // {sigma, mix, m} of the fused M-step + E-step (svr_mstep_estep)

  // geometry
  std::vector<float> slice_dims;   // ns*3
  std::vector<uint32_t> stack_sizes;   // updateStackSizes (RC.cuh:210)
  std::vector<float> mI2W, mW2I, mT, mTinv;
  float reconI2W[16], reconW2I[16];
  float psf_c0[3] = {0, 0, 0};
  float quality_factor = 1.0f;
  bool have_dims = false, have_mats = false, have_psf = false, sc_dirty = true;
  SliceConst *d_sc = nullptr;

  // work lists
  uint32_t *d_active = nullptr, *d_psf_list = nullptr, *d_counter = nullptr, *d_tiles = nullptr,
           *d_tiles_fb = nullptr;
  uint32_t n_active = 0, n_psf = 0, n_tiles = 0;
  int tiles_x = 0, tiles_y = 0, tile_w = 4, tile_h = 4;
  int reg_blind = 4;        // GPU slice-to-volume registration: line-search steps per host round trip, the active count on the device (0: a round trip per step)
  int tune_tiles = 0;       // tiles a tuner's trial launch runs on (0: TUNE_TILES); a short job -- the command lines -- asks for fewer
  int reg_red_threads = 0;  // workgroup size of the registration's per-image reductions: 0 = by image size, 256, 1024
  int reg_batch = 1;        // GPU slice-to-volume registration: the twelve evaluations of a gradient as one launch sequence (0: one by one)
  int pvr_reg_levels = 3, pvr_reg_steps = 4, pvr_reg_iterations = 20;   // PatchBased2D3DRegistration_gpu2 schedule (tests shorten it)
  bool psf_list_valid = false;
  unsigned char *d_gauss_flag = nullptr;   // pixels whose sume passed in the current Gaussian pass
  uint32_t *d_tiles_tmp = nullptr;         // tile list of the Gaussian passes
  uint32_t *d_tiles_sample = nullptr;      // every stride-th tile, for the trial launches of the tuners
  size_t tiles_sample_cap = 0;
  size_t tiles_tmp_cap = 0;                // its capacity in tiles (the tile shapes can change between calls)
  int gauss_mode = 1;                      // 1 = unit-based pass 1 (fwd_unit_kernel<GAUSS1>) + the LDS scatter, 0 = psf_kernel<MODE_GAUSS>
  uint32_t *d_tiles_fwd = nullptr;   // tiles of fwd_tw x fwd_th pixels for fwd_unit_kernel
  uint32_t n_tiles_fwd = 0;
  bool tiles_back_valid = false, tiles_fwd_valid = false;   // the tile lists belong to the current PSF list (ensure_tiles_back / ensure_tiles_fwd)
  int fwd_tw = 4, fwd_th = 4, fwd_tiles_x = 0, fwd_tiles_y = 0;
  int fwd_unit_cap = 9300;  // box voxels (float2) of fwd_unit_kernel: 72.7 KiB + up to 6.6 KiB static = 64 LDS granules of 1280 B -> 2 workgroups of 8 waves per CU (9400 would push the GAUSS1 table instantiation to 65)
  int fwd_mode = 2;         // 2 = the gather over (cell, plane) items (fwd_cell_kernel, svr_cell.inc; SVR on the fly -- patch-based runs and the
                            // coefficient table take 1), >= 1 = unit-based gather per slice tile (fwd_unit_kernel), 0 = wave-per-pixel kernel (psf_kernel<MODE_FWD>)
  // The forward tile shape that suits a problem depends on how many voxels a pixel spans: at 2 voxels per pixel (0.5 mm
  // reconstructions of 1 mm pixels) the box of a 4x4 tile no longer fits the LDS and every tap falls back to global loads.  The first forward pass of a problem times the candidate shapes on the real data
  // and keeps the fastest; the results do not depend on the shape (per-pixel sums in a fixed order).
  bool fwd_tune_pending = false, fwd_tile_user = false;
  int fwd_autotune = 0;     // 1: tile shapes by timed trial launches (rounds 1-3); 0: from the geometry (tile_shape_rule): every run, every rank the same
  // the same for the scatter's tile (4x4 pixels; 4x2 and 2x2 once a pixel spans more than ~2.4 voxels), timed on the first
  // back-projection after new slice geometry.  The scatter's sums are float atomics in run-dependent order with any shape.
  bool back_tune_pending = false, tile_user = false, in_tune = false;
  int pvr = 0;              // 1: patch-to-volume constants and kernels (svr_set_option "pvr")
  int pvr_mode = 1;         // PVR kernels: 1 = the unit-based gather / wave-owned scatter with support 12, 0 = wave-per-pixel (pvr_kernel)
  unsigned char *d_spx = nullptr;
  int back_mode = 5;        // 5 = cell-owned planes, staged and combined in a fixed order: no atomics (svr_cell.inc; the default for SVR on the fly --
                            // patch-based runs and the coefficient table take 4 unless the mode was set explicitly),
                            // 4 = wave-owned LDS planes (back_wave_kernel), 3 = the workgroup kernel for every tile (back_slot_kernel<8>),
                            // 1 = LDS tiles with ds_add_f32 (back_tiled_kernel), 0 = direct device-scope atomics per tap (psf_kernel<MODE_BACK>)
  int tile_cap = 0;         // voxels of LDS accumulator per workgroup
  int n_cu = 0;             // compute units of the device (the grids of the persistent cell kernels)
  int dbg_back = 0;
  int dbg_fwd_lds = 0;      // dev experiment: extra dynamic LDS on the forward launch (limits occupancy)
  uint32_t n_tiles_fb = 0;
  uint32_t n_tiles_fb8 = 0;   // tiles of the last scatter that the wave-owned kernel handed to the workgroup kernel (box larger than wave_cap)
  // PSF launches that were asked for on the cell path (back_mode 5 / fwd_mode 2: no atomics, bit-identical from run to run) and left it
  // because the cell lists cannot hold the geometry (svr_cell.inc cell_prepare: `usable`): [0] scatters (-> back_mode 4: float atomics,
  // run-dependent last bits), [1] gathers, [2] pass 1 of the Gaussian reconstruction (-> the tile kernels: same bits, slower);
  // [3] tiles of tiled scatters re-run by a workgroup kernel.  svr_fallbacks() reads them; the first of each kind says so on stderr.
  uint64_t fallbacks[4] = {0, 0, 0, 0};
  void note_fallback(int kind, const char *what) {
    if (!fallbacks[kind]++) fprintf(stderr, "svr: %s -- counted in svr_fallbacks()[%d]\n", what, kind);
  }
  uint32_t *d_tiles_fb2 = nullptr;
  bool wave_cap_user = false;
  bool back_mode_user = false;   // svr_set_option("back_mode") was called: no automatic choice between 5 and 4
  bool fwd_mode_user = false;    // likewise "fwd_mode"
  bool sr_no_wait = false;       // inside svr_superresolution: its two halves do not wait for the device
  // The coefficient table: what irtkReconstruction::CoeffInit keeps as _volcoeffs on the CPU path (RG.cc:2305-2673) --
  // every PSF pixel's evaluated taps, 16 KiB per pixel, written once per slice geometry by k_coeff_build and streamed by
  // the COEFF instantiations of the scatter and the gather instead of being re-evaluated in every SR iteration.  The
  // reference's GPU path recomputes them (a 2015 GPU had 6-12 GB); 288 GB of HBM hold them for every workload of
  // BASELINE.json (P4: 19 GB, S8: 174 GB).  Values are bit-identical to the on-the-fly evaluation by construction.
  float4 *d_coeff = nullptr;
  uint32_t *d_coeff_id = nullptr;
  uint32_t *d_coeff_order = nullptr;   // the active pixels in the order of their places in the table (coeff_order)
  size_t coeff_order_cap = 0;
  size_t coeff_cap = 0;        // pixels the allocation holds
  bool coeff_valid = false;
  int coeff_mode = 1;          // option "coeff_table": 0 = evaluate on the fly, 1 = stream the table (falls back to 0 if it does not fit).  Round 6: ON by
                               // default for slice-to-volume runs (written by the first gather of the SR iterations, coeff_lazy: P4 +7 %, S8 +5 % with one
                               // table per four SR iterations INSIDE the timed region), off for the patch-based path (no gain there: DESIGN 5.1b)
  bool legacy_ok = false;      // option "legacy_kernels": back_mode 0 / 1 / 3 and fwd_mode 0 may be asked for (tests)
  bool coeff_user = false;     // svr_set_option("coeff_table") was called: the "pvr" option leaves the mode alone
  int coeff_lazy = 1;          // option "coeff_lazy" (round 6): 1 = the table is written by the first gather of the SR iterations after a new slice geometry
                               // (fwd_cell_kernel<.., 3>: the evaluation that pass needs anyway) instead of by k_coeff_build; passes before it evaluate
  bool coeff_full = false;     // the table holds every pixel with s != -1 (k_coeff_build), not only the PSF pixels of the gather that wrote it
  const uint32_t *coeff_list = nullptr;   // the active pixels in table order (d_coeff_order, or d_active), while coeff_ids_valid
  bool coeff_ids_valid = false;   // d_coeff_order / d_coeff_id match the current cell lists (cell_invalidate resets): a table thrown away without a new
                                  // geometry (svr_set_option coeff_invalidate) keeps its pixels' places
  int wave_groups = 1, wave_cap = 2096;      // back_wave_kernel: wavefronts per tile, box voxels of a wavefront's four planes (14 LDS granules of 1280 B with the static part: 9 wavefronts per CU)

  // reductions
  double *d_partial = nullptr, *d_per_slice = nullptr, *d_out = nullptr;
  double *d_em_all = nullptr;       // [world][8]: every rank's M-step sums (svr_mstep_partial / svr_mstep_estep_ranks)
  size_t em_all_cap = 0;
  int chunks = 0;

  // bias correction (allocated only when bias correction is enabled)
  float *d_bias = nullptr, *d_wb = nullptr, *d_wr = nullptr, *d_buffer = nullptr;       // slice grid
  float *d_bias_vol = nullptr, *d_volume_weights = nullptr, *d_maskC = nullptr, *d_mbuf = nullptr;   // volume
  float mask_sigma_bias = 12.0f;
  bool maskC_valid = false;

  // registration cost (NCC)
  short *d_reg_targets = nullptr, *d_reg_source = nullptr;
  short *d_pyr_full[2] = {nullptr, nullptr}, *d_pyr_a = nullptr, *d_pyr_b = nullptr;   // device pyramid of the registration (svr_pyr_*)
  size_t pyr_full_cap[2] = {0, 0}, pyr_work_cap = 0;
  void *d_pyr_meta = nullptr;
  size_t pyr_meta_cap = 0;
  int *d_ncc_idx = nullptr;              // grow-only scratch of svr_ncc_evaluate
  double *d_ncc_m = nullptr;
  long long *d_ncc_s = nullptr;
  size_t ncc_cap = 0;
  int reg_tx = 0, reg_ty = 0, reg_n = 0;
  uint32_t reg_vx = 0, reg_vy = 0, reg_vz = 0;

  // GPU slice-to-volume registration (svr_reg.inc)
  RegState *reg = nullptr;

  // the scatter without atomics (back_mode 5, svr_cell.inc): cell size in voxels (x, lane axis), wavefronts per item
  SliceEm *sem = nullptr;         // the slice-level EM on the device (svr_em.inc)
  SlabPlan *slab = nullptr;       // sharded runs: the index lists and slab boundaries of reduce-scatter -> slab update -> all-gather (svr_slab.inc)
  bool vol_clean[2] = {false, false};   // d_recon_volw / d_recon_new known to be zero outside the dilated mask (svr_slab.inc)
  CellState *cell = nullptr;      // the scatter's cell lists
  CellState *cell_g = nullptr;    // the gather's, when it works on another cell size (cell_prepare_gather)
  int *d_cellc = nullptr;         // centre voxel + class of every active pixel (k_cell_centres), shared by the two
  size_t cellc_cap = 0;
  uint32_t cellc_n = 0;
  bool cellc_valid = false;
  int cellc_range[16] = {0};
  int cell_w = 0, cell_h = 0, cell_gw = 0, cell_gh = 0;   // 0: by the pixel density (cell_auto_size); cell_gw / cell_gh: the gather's own
  int cell_order = 1;       // > 0: items in order of falling work, in classes of 2^(cell_order - 1) pixels (0: (cell, plane) order)
  int cell_combine = 2;     // 2: one item_of load per wavefront and class, early out where nothing is staged (k_cell_combine_wave); 1: a voxel's slabs in two batches (k_cell_combine_fast); 0: the general form.  Same bits
  int cell_balance = 0;     // an item heavier than the launch's work / (1024 x cell_balance) is cut into parts (0: never)
  int cell_split = 1, cell_qx = 1, cell_band = 3;   // cell_qx: cells of a quad along x (1, 2 or 4; the other factor along the lane axis)

  // timers
  bool timers = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;        // tile tuning
  // kernel timers: event pairs are recorded on the stream and read back later (svr_timer_get / reset / every TIMER_BATCH
  // pairs) -- timing a pass does not synchronise the stream
  struct TimedSpan { int which; hipEvent_t a, b; };
  std::vector<hipEvent_t> ev_free;
  std::vector<TimedSpan> ev_pending;
  hipEvent_t ev_open = nullptr;                   // svr_timer_begin .. svr_timer_end
  double t_ms[SVR_T_COUNT] = {0};
  long t_n[SVR_T_COUNT] = {0};

  float *recon() { return recon_cur; }
  float *volw() { return d_recon_volw + nv; }
  float *addon() { return d_addon_cmap; }
  float *cmap() { return d_addon_cmap + nv; }
};

namespace {

int fail(svr_ctx *c, int code, const std::string &msg) {
  if (c) c->err = msg;
  return code;
}
// Every public entry point selects the context's device first: the command lines run one host thread per device and start
// fresh threads per phase (a new thread's current device is 0), so hipMalloc / hipMemGetInfo / event calls inside an entry
// point would otherwise land on GPU 0 until the first call that happened to set the device.
#define SVR_ENTER(c)                                                                        \
  do {                                                                                      \
    if (c) (void)hipSetDevice((c)->device);                                                 \
  } while (0)
#define HIPCHK(expr)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(ctx, (int)e_, std::string(#expr) + ": " + hipGetErrorString(e_));       \
  } while (0)
#define KCHK(name)                                                                          \
  do {                                                                                      \
    hipError_t e_ = hipGetLastError();                                                      \
    if (e_ != hipSuccess)                                                                   \
      return fail(ctx, (int)e_, std::string(name) + ": " + hipGetErrorString(e_));        \
  } while (0)
#define NEED(cond, what)                                                                    \
  do {                                                                                      \
    if (!(cond)) return fail(ctx, SVR_E_STATE, std::string(__func__) + ": " + what);      \
  } while (0)

// The scatter variant in effect: the cell-owned scatter (5) unless the caller named another (4 = wave-owned planes per
// slice tile with the atomic flush, the default until round 3; patch-based runs took it until the cell kernels got five
// slots of 12 lanes for support 12 and cell sizes that follow the pixel density: PVR4 11.3 -> 9.4 ms, PVR8spx 36.6 -> 27.3 ms).
inline int back_mode_eff(const svr_ctx *ctx) { return ctx->back_mode; }
void reg_free(RegState *r);
void cell_free(CellState *c);
void cell_invalidate(svr_ctx *ctx);
void cell_pids_invalidate(svr_ctx *ctx);
void cell_gf_invalidate(svr_ctx *ctx);

template <class T>
void free_dev(T *&p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

constexpr size_t TIMER_BATCH = 256;
inline hipEvent_t timer_event(svr_ctx *c) {
  if (!c->ev_free.empty()) { hipEvent_t e = c->ev_free.back(); c->ev_free.pop_back(); return e; }
  hipEvent_t e = nullptr;
  // timing only: nobody reads device memory on the strength of these events, so they need not write the caches back to system scope when
  // they are recorded (hipEventDisableSystemFence: "for events that are only being used to measure timing") -- the default's fence is
  // what made every recorded event a 5.8 us hole in the stream (profiles/r05_step_timeline_p4.txt)
  if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) { e = nullptr; (void)hipEventCreate(&e); }
  return e;
}
// the recorded pairs -> t_ms / t_n (waits for the last one)
inline void timers_resolve(svr_ctx *c) {
  for (const auto &sp : c->ev_pending) {
    float ms = 0;
    if (sp.a && sp.b && hipEventSynchronize(sp.b) == hipSuccess && hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
      const int w = sp.which & 255, also = (sp.which >> 8) - 1;      // (ScopedTimer::also: the same pair counted under a second name)
      c->t_ms[w] += ms;
      c->t_n[w] += 1;
      if (also >= 0 && also < SVR_T_COUNT) { c->t_ms[also] += ms; c->t_n[also] += 1; }
    }
    if (sp.a) c->ev_free.push_back(sp.a);
    if (sp.b) c->ev_free.push_back(sp.b);
  }
  c->ev_pending.clear();
}
inline void timer_close(svr_ctx *c, int which, hipEvent_t a) {
  hipEvent_t b = timer_event(c);
  if (b) (void)hipEventRecord(b, c->stream);
  c->ev_pending.push_back({which, a, b});
  if (c->ev_pending.size() >= TIMER_BATCH) timers_resolve(c);
}
struct ScopedTimer {
  svr_ctx *c;
  int which;
  hipEvent_t a = nullptr;
  ScopedTimer(svr_ctx *c_, int w) : c(c_), which(w) {
    if (c->timers) {
      a = timer_event(c);
      if (a) (void)hipEventRecord(a, c->stream);
    }
  }
  void also(int w2) { which = (which & 255) | ((w2 + 1) << 8); }   // the same interval under a second name (what kind of pass it was)
  void stop() {
    if (a) timer_close(c, which, a);
    a = nullptr;
  }
  ~ScopedTimer() {                                  // an error path left before stop(): the event goes back unused
    if (a) c->ev_free.push_back(a);
  }
};

inline unsigned nblk(size_t n, unsigned b = 256) { return (unsigned)((n + b - 1) / b); }

void free_volume(svr_ctx *c) {
  c->recon_cur = nullptr;
  free_dev(c->d_recon_volw);
  free_dev(c->d_addon_cmap);
  free_dev(c->d_snap);
  free_dev(c->d_recon_new);
  free_dev(c->d_volm);
}
void free_slices(svr_ctx *c) {
  free_dev(c->d_slices); free_dev(c->d_weights); free_dev(c->d_simslices); free_dev(c->d_simweights);
  free_dev(c->d_psf_sums); free_dev(c->d_siminside); free_dev(c->d_voxcount); free_dev(c->d_scales);
  free_dev(c->d_slice_weights); free_dev(c->d_scales_host_copy); free_dev(c->d_tmp_ns);
  c->mir_scales.clear(); c->mir_slice_weights.clear(); c->mir_scales_copy.clear();
  if (c->h_up) { (void)hipStreamSynchronize(c->stream); (void)hipHostFree(c->h_up); c->h_up = nullptr; }
  if (c->h_down) { (void)hipStreamSynchronize(c->stream); (void)hipHostFree(c->h_down); c->h_down = nullptr; c->down_cap = c->down_used = 0; c->down_items.clear(); }
  free_dev(c->d_em);
  for (int k = 0; k < svr_ctx::UP_SLOTS; ++k) {
    if (c->up_ev[k]) { (void)hipEventDestroy(c->up_ev[k]); c->up_ev[k] = nullptr; }
    c->up_busy[k] = false;
  }
  free_dev(c->d_slice_inside); free_dev(c->d_sc); free_dev(c->d_active); free_dev(c->d_psf_list);
  free_dev(c->d_tiles);
  free_dev(c->d_tiles_fwd);
  free_dev(c->d_gauss_flag);
  free_dev(c->d_tiles_tmp);
  free_dev(c->d_tiles_sample); c->tiles_sample_cap = 0;
  free_dev(c->d_tiles_fb);
  free_dev(c->d_tiles_fb2);
  free_dev(c->d_partial); free_dev(c->d_per_slice);
}

// builds the per-slice constants (host float math, literal order) and uploads them
int prepare_slice_consts(svr_ctx *ctx) {
  NEED(ctx->have_dims && ctx->have_mats && ctx->have_slices, "slice dims / matrices / storage not set");
  if (!ctx->sc_dirty) return SVR_OK;
  std::vector<SliceConst> h(ctx->ns);
  for (uint32_t s = 0; s < ctx->ns; ++s) {
    SliceConst &S = h[s];
    memset(&S, 0, sizeof(S));
    memcpy(S.I2W, &ctx->mI2W[16 * s], 12 * sizeof(float));
    memcpy(S.T, &ctx->mT[16 * s], 12 * sizeof(float));
    float tmp[16], A[16];
    matmul4(&ctx->mW2I[16 * s], &ctx->mTinv[16 * s], tmp);   // RC.cu:223 (left to right)
    matmul4(tmp, ctx->reconI2W, A);
    memcpy(S.A, A, 12 * sizeof(float));
    for (int k = 0; k < 3; ++k) S.dim[k] = ctx->slice_dims[3 * s + k];
    S.kx = S.dim[0] / 2.3548f;
    S.ky = S.dim[1] / 2.3548f;
    // SVR: sigma_z = dz / 2.3548 (RC.cu:114); PVR: sigma_z = dz and the offset is divided by 2.5
    // (pointSpreadFunction.cuh:76,112)
    float sigmaz = ctx->pvr ? S.dim[2] : S.dim[2] / 2.3548f;
    S.inv2s2 = 1.0f / (2.0f * sigmaz * sigmaz);
    for (int j = 0; j < 3; ++j) {
      S.Lp[0 * 3 + j] = (A[0 * 4 + j] * S.dim[0]) * S.kx;
      S.Lp[1 * 3 + j] = (A[1 * 4 + j] * S.dim[1]) * S.ky;
      S.Lp[2 * 3 + j] = ctx->pvr ? (A[2 * 4 + j] * S.dim[2]) / 2.5f : A[2 * 4 + j] * S.dim[2];
    }
    // Gaussian recurrence (gauss_pairs): the ratio of ratios, or -1 where the first ratio exp(+-2 z' dz' c - dd c)
    // of a row that passes the central test (a <= GAUSS_AMAX) could leave the float range
    S.dd = S.Lp[6] * S.Lp[6];
    const double tmax = 2.0 * sqrt((double)GAUSS_AMAX * (double)S.inv2s2) * fabs((double)S.Lp[6]) + (double)S.dd * (double)S.inv2s2;
    S.w = tmax <= 80.0 ? canon_exp_neg((2.0f * S.dd) * S.inv2s2) : -1.0f;
    S.own = fabsf(S.Lp[7]) > fabsf(S.Lp[8]) ? 1 : 2;
    {
      const int G = S.own == 1 ? 2 : 1;                    // the lane axis
      const float D = S.Lp[0] * S.Lp[3 + G] - S.Lp[G] * S.Lp[3];
      S.invD = fabsf(D) > 1e-4f ? 1.0f / D : 0.0f;
    }
  }
  HIPCHK(hipMemcpyAsync(ctx->d_sc, h.data(), h.size() * sizeof(SliceConst), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->sc_dirty = false;
  ctx->coeff_valid = false; cell_invalidate(ctx);                               // new slice geometry: new taps
  return SVR_OK;
}

// A dispatch holds at most 2^32 - 1 work-items per dimension (launch_fwd_unit below tells how that was found): every kernel
// that takes one workgroup per entry of a tile list sends long lists out in pieces.  f(offset, count) launches one piece.
template <class F>
int in_pieces(uint32_t n, uint32_t threads_per_tile, F &&f) {
  const long env = getenv("SVR_LIST_PIECE") ? atol(getenv("SVR_LIST_PIECE")) : 0;     // test hook: short pieces on a small problem
  uint32_t piece = (uint32_t)((((1ull << 32) - 1) / threads_per_tile) & ~1023ull);
  if (env > 0) piece = (uint32_t)std::min<long>(env, piece);
  for (uint32_t off = 0; off < n; off += piece) {
    const int r = f(off, std::min(piece, n - off));
    if (r) return r;
  }
  return SVR_OK;
}
// one wavefront per entry of a pixel list (psf_kernel, pvr_kernel, k_coeff_build): launch(args of one piece, offset of the piece)
template <class L>
int pixel_list_in_pieces(const PsfArgs &a0, L &&launch) {
  return in_pieces(a0.n, 64, [&](uint32_t off, uint32_t cnt) {
    PsfArgs a = a0;
    a.list = a0.list + off;
    a.n = cnt;
    return launch(a, off);
  });
}
// tiles of tw x th pixels that hold at least one active pixel (k_build_tiles: a wavefront per candidate tile)
int build_tile_list(svr_ctx *ctx, const float *psf_sums, const unsigned char *flag, int tiles_x, int tiles_y, int tw, int th, uint32_t *tiles,
                    uint32_t *counter) {
  const uint64_t total64 = (uint64_t)tiles_x * tiles_y * ctx->ns;
  if (total64 >= (1ull << 32)) return fail(ctx, SVR_E_ARG, "more than 2^32 candidate tiles");
  return in_pieces((uint32_t)total64, 64, [&](uint32_t off, uint32_t cnt) {
    hipLaunchKernelGGL(k_build_tiles, dim3(nblk(cnt, 16)), dim3(1024), 0, ctx->stream, ctx->d_slices, psf_sums, flag, (int)ctx->sx, (int)ctx->sy,
                       (int)ctx->ns, tiles_x, tiles_y, tw, th, tiles, counter, (const unsigned char *)nullptr, 0, off, off + cnt);
    KCHK("k_build_tiles");
    return (int)SVR_OK;
  });
}

int build_list(svr_ctx *ctx, bool with_psf) {
  HIPCHK(hipMemsetAsync(ctx->d_counter, 0, sizeof(uint32_t), ctx->stream));
  uint32_t *list = with_psf ? ctx->d_psf_list : ctx->d_active;
  hipLaunchKernelGGL(k_compact, dim3(std::min(nblk(ctx->np, 1024), 4096u)), dim3(1024), 0, ctx->stream, ctx->d_slices,
                     with_psf ? ctx->d_psf_sums : (const float *)nullptr, (uint32_t)ctx->np, list, ctx->d_counter);
  KCHK("k_compact");
  uint32_t n = 0;
  HIPCHK(hipMemcpyAsync(&n, ctx->d_counter, sizeof(n), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (with_psf) {
    ctx->n_psf = n;
    ctx->psf_list_valid = true;
    // the tile lists of the tile kernels (the table's gather, the fallback modes) are built when one of them asks (round 4: the default
    // path works on (cell, plane) items and used to pay for both lists, two more waits and an allocation per slice geometry)
    ctx->tiles_back_valid = ctx->tiles_fwd_valid = false;
  } else {
    ctx->n_active = n;
  }
  return SVR_OK;
}
int ensure_psf_list(svr_ctx *ctx);
// tiles of tile_w x tile_h pixels that hold at least one pixel of the PSF list (the tiled scatters)
int ensure_tiles_back(svr_ctx *ctx) {
  int r = ensure_psf_list(ctx);
  if (r || ctx->tiles_back_valid) return r;
  uint32_t n = 0;
  HIPCHK(hipMemsetAsync(ctx->d_counter, 0, sizeof(uint32_t), ctx->stream));
  if ((r = build_tile_list(ctx, ctx->d_psf_sums, nullptr, ctx->tiles_x, ctx->tiles_y, ctx->tile_w, ctx->tile_h, ctx->d_tiles, ctx->d_counter))) return r;
  HIPCHK(hipMemcpyAsync(&n, ctx->d_counter, sizeof(n), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->n_tiles = n;
  ctx->tiles_back_valid = true;
  return SVR_OK;
}
// ... of fwd_tw x fwd_th pixels (fwd_unit_kernel)
int ensure_tiles_fwd(svr_ctx *ctx) {
  int r = ensure_psf_list(ctx);
  if (r || ctx->tiles_fwd_valid) return r;
  uint32_t n = 0;
  free_dev(ctx->d_tiles_fwd);
  ctx->fwd_tiles_x = (int)((ctx->sx + ctx->fwd_tw - 1) / ctx->fwd_tw);
  ctx->fwd_tiles_y = (int)((ctx->sy + ctx->fwd_th - 1) / ctx->fwd_th);
  const uint32_t total_f = (uint32_t)ctx->fwd_tiles_x * ctx->fwd_tiles_y * ctx->ns;
  HIPCHK(hipMalloc(&ctx->d_tiles_fwd, (size_t)total_f * sizeof(uint32_t)));
  HIPCHK(hipMemsetAsync(ctx->d_counter, 0, sizeof(uint32_t), ctx->stream));
  if ((r = build_tile_list(ctx, ctx->d_psf_sums, nullptr, ctx->fwd_tiles_x, ctx->fwd_tiles_y, ctx->fwd_tw, ctx->fwd_th, ctx->d_tiles_fwd, ctx->d_counter))) return r;
  HIPCHK(hipMemcpyAsync(&n, ctx->d_counter, sizeof(n), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->n_tiles_fwd = n;
  ctx->tiles_fwd_valid = true;
  return SVR_OK;
}

PsfArgs make_args(svr_ctx *ctx) {
  PsfArgs a;
  memset(&a, 0, sizeof(a));
  a.sc = ctx->d_sc;
  a.vg.vx = (int)ctx->vx; a.vg.vy = (int)ctx->vy; a.vg.vz = (int)ctx->vz;
  memcpy(a.vg.W2I, ctx->reconW2I, 12 * sizeof(float));
  memcpy(a.vg.c0, ctx->psf_c0, 3 * sizeof(float));
  a.vg.pvr = ctx->pvr;
  a.spx = ctx->d_spx;
  a.sx = (int)ctx->sx; a.sy = (int)ctx->sy;
  a.slices = ctx->d_slices; a.mask = ctx->d_mask; a.psf_sums = ctx->d_psf_sums;
  a.scales = ctx->d_scales;
  a.bias = ctx->disable_bias ? nullptr : ctx->d_bias;
  a.recon = ctx->recon(); a.volw = ctx->volw(); a.voxcount = ctx->d_voxcount;
  a.vol = ctx->recon(); a.simslices = ctx->d_simslices; a.simweights = ctx->d_simweights;
  a.siminside = ctx->d_siminside;
  a.slice_inside = ctx->d_slice_inside;
  a.weights = ctx->d_weights; a.slice_weights = ctx->d_slice_weights;
  a.addon = ctx->addon(); a.cmap = ctx->cmap();
  return a;
}

void give_coeff(const svr_ctx *ctx, PsfArgs &a) { a.coeff = ctx->d_coeff; a.coeff_id = ctx->d_coeff_id; }   // the table goes to a pass's arguments

// bias-path buffers: slice-grid {bias, wb, wresidual, buffer} (RC.cu:1510-1539) and volume
// {bias, volume_weights, maskC} (RC.cu:1203-1214, 1129-1157); created on first use after
// svr_set_flags(ctx, disable_bias_correction = 0, ...)
int ensure_bias_buffers(svr_ctx *ctx) {
  if (ctx->disable_bias) return SVR_OK;
  if (ctx->np && !ctx->d_bias) {
    const size_t fb = ctx->np * sizeof(float);
    HIPCHK(hipMalloc(&ctx->d_bias, fb)); HIPCHK(hipMalloc(&ctx->d_wb, fb));
    HIPCHK(hipMalloc(&ctx->d_wr, fb)); HIPCHK(hipMalloc(&ctx->d_buffer, fb));
    HIPCHK(hipMemsetAsync(ctx->d_bias, 0, fb, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_wb, 0, fb, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_wr, 0, fb, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_buffer, 0, fb, ctx->stream));
  }
  if (ctx->nv && !ctx->d_bias_vol) {
    const size_t vb = ctx->nv * sizeof(float);
    HIPCHK(hipMalloc(&ctx->d_bias_vol, vb)); HIPCHK(hipMalloc(&ctx->d_volume_weights, vb));
    HIPCHK(hipMalloc(&ctx->d_maskC, vb)); HIPCHK(hipMalloc(&ctx->d_mbuf, vb));
    HIPCHK(hipMemsetAsync(ctx->d_bias_vol, 0, vb, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_volume_weights, 0, vb, ctx->stream));
    ctx->maskC_valid = false;
  }
  if (ctx->nv && ctx->have_mask && !ctx->maskC_valid) {
    // maskC_ = mask blurred with sigma_bias: x into a zeroed buffer, y back, z into the buffer, copy (RC.cu:1129-1157)
    const size_t vb = ctx->nv * sizeof(float);
    const dim3 grid((ctx->vx + 63) / 64, (ctx->vy + 3) / 4, ctx->vz);
    HIPCHK(hipMemcpyAsync(ctx->d_maskC, ctx->d_mask, vb, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_mbuf, 0, vb, ctx->stream));
    hipLaunchKernelGGL(k_gauss_conv3d, grid, dim3(256), 0, ctx->stream, ctx->d_maskC, ctx->d_mbuf, ctx->mask_sigma_bias, 0,
                       ctx->vdim[0], (int)ctx->vx, (int)ctx->vy, (int)ctx->vz);
    hipLaunchKernelGGL(k_gauss_conv3d, grid, dim3(256), 0, ctx->stream, ctx->d_mbuf, ctx->d_maskC, ctx->mask_sigma_bias, 1,
                       ctx->vdim[1], (int)ctx->vx, (int)ctx->vy, (int)ctx->vz);
    hipLaunchKernelGGL(k_gauss_conv3d, grid, dim3(256), 0, ctx->stream, ctx->d_maskC, ctx->d_mbuf, ctx->mask_sigma_bias, 2,
                       ctx->vdim[2], (int)ctx->vx, (int)ctx->vy, (int)ctx->vz);
    KCHK("k_gauss_conv3d(mask)");
    HIPCHK(hipMemcpyAsync(ctx->d_maskC, ctx->d_mbuf, vb, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->maskC_valid = true;
  }
  return SVR_OK;
}

int ready(svr_ctx *ctx) {
  NEED(ctx->nv > 0, "reconstruction volume not initialised");
  NEED(ctx->have_mask, "mask not set");
  NEED(ctx->have_slices, "slices not filled");
  NEED(ctx->have_scales, "scale vector not set");
  NEED(ctx->have_psf, "generatePSFVolume not called");
  int r = ensure_bias_buffers(ctx);
  if (r) return r;
  return prepare_slice_consts(ctx);
}

// The gather takes one workgroup of FWDU_WAVES wavefronts per tile, and a dispatch holds at most 2^32 - 1 work-items per
// dimension: beyond 8.4 M tiles (2 x 2-pixel tiles of a 0.5 mm patch-based case: 12.2 M) a single launch silently ran the
// tile count modulo 2^23 -- it looked 3.3 x faster in the tuner and left two thirds of the simulated slices stale.  Long
// lists go out in pieces of 2^22 tiles (a multiple of the XCD group of xcd_run_index).
template <bool GAUSS1>
void launch_fwd_unit(svr_ctx *ctx, const PsfArgs &a, TileArgs ta, size_t lds) {
  constexpr uint32_t PIECE_MAX = 1u << 22;
  static_assert((uint64_t)PIECE_MAX * FWDU_WAVES * 64 < (1ull << 32), "a piece must fit one dispatch");
  const char *env = getenv("SVR_FWD_PIECE");             // test hook: short pieces on a small problem
  const uint32_t PIECE = env && atol(env) > 0 ? (uint32_t)std::min<long>(atol(env), PIECE_MAX) : PIECE_MAX;
  const uint32_t *tiles = ta.tiles;
  const uint32_t total = ta.ntiles;
  for (uint32_t off = 0; off < total; off += PIECE) {
    ta.tiles = tiles + off;
    ta.ntiles = std::min(PIECE, total - off);
    const dim3 grid(ta.ntiles), block(FWDU_WAVES * 64);
    if (ctx->pvr && a.coeff) hipLaunchKernelGGL((fwd_unit_kernel<GAUSS1, PVR_N, true, true>), grid, block, lds, ctx->stream, a, ta);
    else if (ctx->pvr) hipLaunchKernelGGL((fwd_unit_kernel<GAUSS1, PVR_N, true>), grid, block, lds, ctx->stream, a, ta);
    else if (a.coeff) hipLaunchKernelGGL((fwd_unit_kernel<GAUSS1, PSF_SUPPORT, false, true>), grid, block, lds, ctx->stream, a, ta);
    else hipLaunchKernelGGL((fwd_unit_kernel<GAUSS1>), grid, block, lds, ctx->stream, a, ta);
  }
}

int ensure_psf_list(svr_ctx *ctx) {
  if (!ctx->psf_list_valid) return build_list(ctx, true);
  return SVR_OK;
}

// The tuners time their candidates on real launches whose results are thrown away.  On a long tile list (S8, the
// patch-based cases: millions of tiles, 0.05-0.2 s per launch, two dozen trials) a trial runs on one run of 4096 consecutive tiles out of every `stride` runs
// instead, so that tuning costs what it costs on P4 (a list of up to 2 x TUNE_TILES tiles is timed whole).
constexpr uint32_t TUNE_TILES = 131072;
__global__ void k_sample_list(const uint32_t *src, uint32_t n_out, uint32_t stride, uint32_t run, uint32_t *dst) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_out) dst[i] = src[(size_t)(i / run) * run * stride + i % run];
}
struct TileSample {
  uint32_t **list = nullptr, *saved = nullptr;
  uint32_t *n = nullptr, saved_n = 0;
  float scale = 1.0f;                    // full list / timed part: the candidates are compared on the time of a whole launch
  int begin(svr_ctx *ctx, uint32_t *&l, uint32_t &count) {
    static const long env_tiles = getenv("SVR_TUNE_TILES") ? atol(getenv("SVR_TUNE_TILES")) : -1;   // 0: always the whole list
    if (env_tiles == 0) return SVR_OK;
    static const long env_run = getenv("SVR_TUNE_RUN") ? atol(getenv("SVR_TUNE_RUN")) : -1;
    const uint32_t target = env_tiles > 0 ? (uint32_t)env_tiles : ctx->tune_tiles > 0 ? (uint32_t)ctx->tune_tiles : TUNE_TILES;
    const uint32_t TUNE_RUN = env_run > 0 ? (uint32_t)env_run : std::min(65536u, std::max(4096u, target / 2));
    const uint32_t stride = count / target;
    if (stride < 2 || (uint64_t)TUNE_RUN * stride > count) return SVR_OK;
    const uint32_t n_out = count / (TUNE_RUN * stride) * TUNE_RUN;     // whole runs only: the last index stays inside the list
    if (ctx->tiles_sample_cap < n_out) {
      free_dev(ctx->d_tiles_sample);
      ctx->tiles_sample_cap = 0;
      HIPCHK(hipMalloc(&ctx->d_tiles_sample, (size_t)n_out * sizeof(uint32_t)));
      ctx->tiles_sample_cap = n_out;
    }
    hipLaunchKernelGGL(k_sample_list, dim3(nblk(n_out)), dim3(256), 0, ctx->stream, l, n_out, stride, TUNE_RUN, ctx->d_tiles_sample);
    KCHK("k_sample_list");
    list = &l; n = &count; saved = l; saved_n = count;
    scale = (float)count / (float)n_out;
    l = ctx->d_tiles_sample; count = n_out;
    return SVR_OK;
  }
  void end() {
    if (list) { *list = saved; *n = saved_n; list = nullptr; }
  }
};

// The scatter over a tile list (back-projection into addon|cmap, or pass 2 of the Gaussian reconstruction into
// recon|volw): the wave-owned kernel with a box of `wave_cap` voxels; tiles whose planes do not fit it are re-run by the
// workgroup kernel (8 wavefronts, the largest box the CU can hold); what fits neither -- strongly oblique tiles of very fine
// volumes -- takes LDS atomics (SVR: back_tiled_kernel) or device atomics per tap (PVR: pvr_tiles_kernel).
// level: 4 = start with the wave-owned kernel, 3 = with the workgroup kernel, 1 = the last resort for every tile.
int coeff_order(svr_ctx *ctx, uint32_t n_active, const uint32_t **list);
// (Re)build the coefficient table if the option is on and the table does not match the current PSF pixels / geometry.
// Returns with ctx->coeff_valid set, or with the mode switched off when the table does not fit the free memory.
__global__ void k_coeff_ids(const uint32_t *list, uint32_t n, uint32_t *coeff_id) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) coeff_id[list[i]] = i;
}
// Can the table be left to the first gather of the SR iterations (coeff_lazy)?  Support 16 on the cell gather.
bool coeff_lazy_ok(const svr_ctx *ctx) { return ctx->coeff_lazy && !ctx->pvr && ctx->fwd_mode == 2; }
// The table's memory and the pixels' places in it (*list = the active pixels in table order).  Returns with ctx->d_coeff set, or with the mode
// switched off when the table does not fit the free memory.
int coeff_prepare(svr_ctx *ctx, const uint32_t **list) {
  // every pixel with s != -1: the Gaussian pass walks them all, the PSF pixels of the SR iterations are a subset
  const size_t npx = ctx->n_active;
  if (npx > ctx->coeff_cap) {
    free_dev(ctx->d_coeff);
    ctx->coeff_cap = 0;
    size_t fr = 0, tot = 0;
    const size_t per_px = ctx->pvr ? (size_t)PVR_N * (PVR_N / 4) * 16 : (size_t)PSF_SUPPORT * (PSF_SUPPORT / 4) * 16;   // float4 per pixel
    const size_t bytes = npx * per_px * sizeof(float4);
    const char *cap_gb = getenv("SVR_COEFF_MAX_GB");     // optional ceiling on the table (GiB): a deployment knob, and how the tests reach the fallback
    const bool over = cap_gb && (double)bytes > atof(cap_gb) * 1073741824.0;
    if (over || hipMemGetInfo(&fr, &tot) != hipSuccess || bytes + (size_t(2) << 30) > fr || hipMalloc(&ctx->d_coeff, bytes) != hipSuccess) {
      (void)hipGetLastError();
      ctx->d_coeff = nullptr;
      ctx->coeff_mode = 0;                                 // does not fit: evaluate on the fly (svr_get_option tells)
      return SVR_OK;
    }
    ctx->coeff_cap = npx;
  }
  if (!ctx->d_coeff_id) { HIPCHK(hipMalloc(&ctx->d_coeff_id, ctx->np * sizeof(uint32_t))); ctx->coeff_ids_valid = false; }
  if (ctx->coeff_ids_valid && ctx->coeff_list) { *list = ctx->coeff_list; return SVR_OK; }
  *list = ctx->d_active;
  // The pixels get their places in the table in the order of the scatter's cell lists (cell, slice, band, position): what a
  // (cell, plane) item of the scatter or a slice tile of the gather reads next then lies within a few megabytes instead of one
  // KiB per slice all over the table (a TLB miss per unit); pixels the cell lists dropped (footprint beyond the volume) follow.
  if (!getenv("SVR_COEFF_UNSORTED")) {
    const int r = coeff_order(ctx, (uint32_t)npx, list);
    if (r) return r;
  }
  ctx->coeff_list = *list;
  return SVR_OK;
}
// (Re)build the coefficient table if the option is on and the table does not match the current PSF pixels / geometry.
// Returns with ctx->coeff_valid set, or with the mode switched off when the table does not fit the free memory.
int ensure_coeff(svr_ctx *ctx) {
  if (!ctx->coeff_mode || ctx->coeff_valid) return SVR_OK;
  const size_t npx = ctx->n_active;
  if (!npx) return SVR_OK;
  PsfArgs a = make_args(ctx);
  a.n = (uint32_t)npx;
  { const int r = coeff_prepare(ctx, &a.list); if (r) return r; }
  if (!ctx->coeff_mode) return SVR_OK;
  ScopedTimer tb(ctx, SVR_T_COEFF_BUILD);
  {
    const int rr = pixel_list_in_pieces(a, [&](const PsfArgs &ap, uint32_t off) {
      if (ctx->pvr) hipLaunchKernelGGL((k_coeff_build<PVR_N, true>), dim3(nblk(ap.n, 4)), dim3(256), 0, ctx->stream, ap, ctx->d_coeff, ctx->d_coeff_id, off);
      else hipLaunchKernelGGL((k_coeff_build<PSF_SUPPORT, false>), dim3(nblk(ap.n, 4)), dim3(256), 0, ctx->stream, ap, ctx->d_coeff, ctx->d_coeff_id, off);
      KCHK("k_coeff_build");
      return (int)SVR_OK;
    });
    if (rr) return rr;
  }
  tb.stop();
  ctx->coeff_valid = true;
  ctx->coeff_full = true;
  if (!ctx->coeff_ids_valid) cell_pids_invalidate(ctx);    // (the records' table ids: svr_cell.inc cell_pids)
  ctx->coeff_ids_valid = true;
  return SVR_OK;
}
// The evaluating cell pass that is about to run writes the table (coeff_lazy): memory, the pixels' places and ids.  *store = the pass should store.
int coeff_begin_store(svr_ctx *ctx, bool *store) {
  *store = false;
  if (!(ctx->coeff_mode && !ctx->coeff_valid && coeff_lazy_ok(ctx) && ctx->n_active)) return SVR_OK;
  const uint32_t *order = nullptr;
  int r = coeff_prepare(ctx, &order);
  if (r) return r;
  if (!ctx->coeff_mode) return SVR_OK;                     // (the table does not fit)
  if (!ctx->coeff_ids_valid) {
    hipLaunchKernelGGL(k_coeff_ids, dim3(nblk(ctx->n_active)), dim3(256), 0, ctx->stream, order, (uint32_t)ctx->n_active, ctx->d_coeff_id);
    KCHK("k_coeff_ids");
    cell_pids_invalidate(ctx);
    ctx->coeff_ids_valid = true;
  }
  *store = true;
  return SVR_OK;
}
// a pass other than the gather of the SR iterations: with coeff_lazy it does not build the table -- it evaluates until that gather has written it
int ensure_coeff_unless_lazy(svr_ctx *ctx) {
  if (ctx->coeff_mode && !ctx->coeff_valid && coeff_lazy_ok(ctx)) return SVR_OK;
  return ensure_coeff(ctx);
}

int launch_slot(svr_ctx *ctx, bool pvr, const PsfArgs &a, const TileArgs &ta_, uint32_t *fb, uint32_t *cnt) {
  const size_t lds = (size_t)ta_.cap * 2 * sizeof(float);
  return in_pieces(ta_.ntiles, 8 * 64, [&](uint32_t off, uint32_t cntp) {
    TileArgs ta = ta_;
    ta.tiles = ta_.tiles + off; ta.ntiles = cntp;
    if (pvr) hipLaunchKernelGGL((back_slot_kernel<8, PVR_N, true>), dim3(ta.ntiles), dim3(8 * 64), lds, ctx->stream, a, ta, fb, cnt);
    else hipLaunchKernelGGL(back_slot_kernel<8>, dim3(ta.ntiles), dim3(8 * 64), lds, ctx->stream, a, ta, fb, cnt);
    KCHK("back_slot_kernel");
    return (int)SVR_OK;
  });
}
int launch_scatter(svr_ctx *ctx, int level, const PsfArgs &a_, TileArgs ta, const uint32_t *tiles, uint32_t n, int mode_last) {
  PsfArgs a = a_;
  const bool pvr = ctx->pvr != 0;
  uint32_t *cnt = ctx->d_counter;                         // [0]: did not fit the wavefront's box, [1]: did not fit the workgroup's
  HIPCHK(hipMemsetAsync(cnt, 0, 2 * sizeof(uint32_t), ctx->stream));
  ctx->n_tiles_fb8 = ctx->n_tiles_fb = 0;
  if (!n) return SVR_OK;
  int r;
  uint32_t nfb[2] = {0, 0};
  const uint32_t *cur = tiles;
  uint32_t ncur = n;
  if (level >= 4) {
    ta.tiles = cur; ta.ntiles = ncur; ta.cap = std::min(ctx->wave_cap, ctx->tile_cap);
    const size_t lds = (size_t)ta.cap * 2 * sizeof(float);
    {
      const TileArgs ta1 = ta;
      r = in_pieces(ncur, 64u * (uint32_t)ctx->wave_groups, [&](uint32_t off, uint32_t cntp) {
        TileArgs tp = ta1;
        tp.tiles = ta1.tiles + off; tp.ntiles = cntp;
        const dim3 grid(cntp * (uint32_t)ctx->wave_groups);
        if (pvr && a.coeff) hipLaunchKernelGGL((back_wave_kernel<PVR_N, true, true>), grid, dim3(64), lds, ctx->stream, a, tp, ctx->wave_groups, ctx->d_tiles_fb, cnt);
        else if (pvr) hipLaunchKernelGGL((back_wave_kernel<PVR_N, true>), grid, dim3(64), lds, ctx->stream, a, tp, ctx->wave_groups, ctx->d_tiles_fb, cnt);
        else if (a.coeff) hipLaunchKernelGGL((back_wave_kernel<PSF_SUPPORT, false, true>), grid, dim3(64), lds, ctx->stream, a, tp, ctx->wave_groups, ctx->d_tiles_fb, cnt);
        else hipLaunchKernelGGL(back_wave_kernel<>, grid, dim3(64), lds, ctx->stream, a, tp, ctx->wave_groups, ctx->d_tiles_fb, cnt);
        return (int)SVR_OK;
      });
      if (r) return r;
    }
    KCHK("back_wave_kernel");
    HIPCHK(hipMemcpyAsync(nfb, cnt, sizeof(nfb), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    cur = ctx->d_tiles_fb; ncur = nfb[0];
    if (ncur && a.coeff && ta.cap < 3696) {
      // table mode: the tiles whose planes did not fit go through the same kernel once more with the largest box worth
      // having (4 planes of 30 x 30 voxels: 5 wavefronts per CU) before the evaluating workgroup kernel gets them
      HIPCHK(hipMemsetAsync(cnt, 0, sizeof(uint32_t), ctx->stream));
      ta.tiles = cur; ta.ntiles = ncur; ta.cap = std::min(3696, ctx->tile_cap);
      if (pvr) hipLaunchKernelGGL((back_wave_kernel<PVR_N, true, true>), dim3(ncur * (uint32_t)ctx->wave_groups), dim3(64), (size_t)ta.cap * 2 * sizeof(float),
                                  ctx->stream, a, ta, ctx->wave_groups, ctx->d_tiles_fb2, cnt);
      else hipLaunchKernelGGL((back_wave_kernel<PSF_SUPPORT, false, true>), dim3(ncur * (uint32_t)ctx->wave_groups), dim3(64), (size_t)ta.cap * 2 * sizeof(float),
                              ctx->stream, a, ta, ctx->wave_groups, ctx->d_tiles_fb2, cnt);
      KCHK("back_wave_kernel (large box)");
      HIPCHK(hipMemcpyAsync(nfb, cnt, sizeof(nfb), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      cur = ctx->d_tiles_fb2; ncur = nfb[0];
      if (ncur) {                                          // the workgroup kernel writes its own rejects to d_tiles_fb2: move the list
        HIPCHK(hipMemcpyAsync(ctx->d_tiles_fb, ctx->d_tiles_fb2, (size_t)ncur * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
        cur = ctx->d_tiles_fb;
      }
    }
    ctx->n_tiles_fb8 = ncur;
    ctx->fallbacks[3] += ncur;
  }
  if (ncur && level >= 3) {
    ta.tiles = cur; ta.ntiles = ncur; ta.cap = ctx->tile_cap;
    if ((r = launch_slot(ctx, pvr, a, ta, ctx->d_tiles_fb2, cnt + 1))) return r;
    HIPCHK(hipMemcpyAsync(nfb, cnt, sizeof(nfb), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    cur = ctx->d_tiles_fb2; ncur = nfb[1];
  }
  if (level >= 3) ctx->n_tiles_fb = ncur;
  if (ncur) {
    ta.tiles = cur; ta.ntiles = ncur; ta.cap = ctx->tile_cap;
    if (pvr && mode_last == MODE_GAUSS2) { a.recon = a.addon; a.volw = a.cmap; }
    const TileArgs ta0 = ta;
    r = in_pieces(ncur, pvr ? WAVES_PER_BLOCK * 64 : TILE_WAVES * 64, [&](uint32_t off, uint32_t cntp) {
      ta.tiles = ta0.tiles + off; ta.ntiles = cntp;
      if (pvr) {
        if (mode_last == MODE_GAUSS2) hipLaunchKernelGGL(pvr_tiles_kernel<MODE_GAUSS2>, dim3(cntp), dim3(WAVES_PER_BLOCK * 64), 0, ctx->stream, a, ta);
        else hipLaunchKernelGGL(pvr_tiles_kernel<MODE_BACK>, dim3(cntp), dim3(WAVES_PER_BLOCK * 64), 0, ctx->stream, a, ta);
      } else {
        hipLaunchKernelGGL(back_tiled_kernel, dim3(cntp), dim3(TILE_WAVES * 64), (size_t)ctx->tile_cap * 2 * sizeof(float), ctx->stream, a, ta);
      }
      KCHK("scatter (last resort)");
      return (int)SVR_OK;
    });
    if (r) return r;
  }
  return SVR_OK;
}

#include "svr_cell.inc"
#include "svr_slab.inc"

// reduce partial[ns*chunks][K] -> per_slice[ns][K] (+ optionally -> d_out[K])
int reduce_partials(svr_ctx *ctx, int K, int mn, int mx, bool global) {
  if (K < 1 || K > REDUCE_MAXK) return fail(ctx, SVR_E_ARG, "reduce_partials: 1..8 quantities (k_reduce_slices holds them side by side; d_partial / d_per_slice are sized for them)");
  hipLaunchKernelGGL(k_reduce_chunks, dim3(nblk((size_t)ctx->ns * K)), dim3(256), 0, ctx->stream,
                     ctx->d_partial, (int)ctx->ns, ctx->chunks, K, mn, mx, ctx->d_per_slice);
  KCHK("k_reduce_chunks");
  if (global) {
    hipLaunchKernelGGL(k_reduce_slices, dim3(1), dim3(256), 0, ctx->stream, ctx->d_per_slice, (int)ctx->ns, K,
                       mn, mx, ctx->d_out);
    KCHK("k_reduce_slices");
  }
  return SVR_OK;
}

// a per-slice vector to the device, asynchronously: through a pinned slot (the caller's memory is free again on return), no
// stream synchronisation; `mirror` = what the device vector holds, an identical vector is not sent again
int upload_ns(svr_ctx *ctx, float *dst, const float *src, std::vector<float> &mirror) {
  const size_t n = ctx->ns;
  if (dst == ctx->d_slice_weights) ctx->sem_weights_current = false;
  if (mirror.size() == n && !memcmp(mirror.data(), src, n * sizeof(float))) return SVR_OK;
  if (!ctx->h_up) {
    HIPCHK(hipHostMalloc((void **)&ctx->h_up, (size_t)svr_ctx::UP_SLOTS * n * sizeof(float), hipHostMallocDefault));
    for (int k = 0; k < svr_ctx::UP_SLOTS; ++k) HIPCHK(hipEventCreateWithFlags(&ctx->up_ev[k], hipEventDisableTiming));
  }
  const int k = ctx->up_next;
  ctx->up_next = (k + 1) % svr_ctx::UP_SLOTS;
  if (ctx->up_busy[k]) HIPCHK(hipEventSynchronize(ctx->up_ev[k]));       // the copy that last used the slot (long done)
  float *slot = ctx->h_up + (size_t)k * n;
  memcpy(slot, src, n * sizeof(float));
  HIPCHK(hipMemcpyAsync(dst, slot, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipEventRecord(ctx->up_ev[k], ctx->stream));
  ctx->up_busy[k] = true;
  mirror.assign(src, src + n);
  return SVR_OK;
}

// queue a small device -> host copy; nothing is in `dst` before down_flush.  A call that fails drops everything queued so far
// (the destinations are the caller's memory: a later, unrelated down_flush must not write to them)
// make room for `total` more bytes of queued copies.  With copies already in flight into the arena the stream is waited for and what
// has arrived moves into the larger arena (round 5 refused here: a batch whose LATER items were larger than its first -- the
// slice-level EM's vectors of the GLOBAL slice count on a rank that holds a fraction of the slices -- failed with SVR_E_STATE)
int down_reserve(svr_ctx *ctx, size_t total) {
  auto drop = [&](int code, const std::string &msg) { ctx->down_items.clear(); ctx->down_used = 0; return fail(ctx, code, msg); };
  const size_t need = ctx->down_used + total;
  if (need <= ctx->down_cap) return SVR_OK;
  const size_t cap = std::max<size_t>(need + need / 2, (size_t)ctx->ns * 12 + 4096);
  unsigned char *fresh = nullptr;
  hipError_t e = hipHostMalloc((void **)&fresh, cap, hipHostMallocDefault);
  if (e != hipSuccess) return drop((int)e, std::string("down_queue: hipHostMalloc: ") + hipGetErrorString(e));
  if (!ctx->down_items.empty()) {
    e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipHostFree(fresh); return drop((int)e, std::string("down_queue: hipStreamSynchronize: ") + hipGetErrorString(e)); }
    memcpy(fresh, ctx->h_down, ctx->down_used);
  }
  if (ctx->h_down) (void)hipHostFree(ctx->h_down);
  ctx->h_down = fresh;
  ctx->down_cap = cap;
  return SVR_OK;
}
int down_queue(svr_ctx *ctx, void *dst, const void *src, size_t bytes) {
  auto drop = [&](int code, const std::string &msg) { ctx->down_items.clear(); ctx->down_used = 0; return fail(ctx, code, msg); };
  const size_t padded = (bytes + 63) & ~(size_t)63;
  { const int r = down_reserve(ctx, padded); if (r) return r; }
  const hipError_t e = hipMemcpyAsync(ctx->h_down + ctx->down_used, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
  if (e != hipSuccess) return drop((int)e, std::string("down_queue: hipMemcpyAsync: ") + hipGetErrorString(e));
  ctx->down_items.push_back({dst, ctx->down_used, bytes});
  ctx->down_used += padded;
  return SVR_OK;
}
int down_flush(svr_ctx *ctx) {
  const hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess)
    for (const auto &it : ctx->down_items) memcpy(it.dst, ctx->h_down + it.off, it.bytes);
  ctx->down_items.clear();
  ctx->down_used = 0;
  if (e != hipSuccess) return fail(ctx, (int)e, std::string("down_flush: ") + hipGetErrorString(e));
  return SVR_OK;
}

}  // namespace

// ==========================================================================================
// C-ABI
// ==========================================================================================
extern "C" {

int svr_create(int device, svr_ctx **out) {
  if (!out) return SVR_E_ARG;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return (int)e;
  if (device < 0 || device >= n) return SVR_E_ARG;
  e = hipSetDevice(device);
  if (e != hipSuccess) return (int)e;
  svr_ctx *ctx = new svr_ctx();
  ctx->device = device;
  e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete ctx; return (int)e; }
  ctx->own_stream = true;
  (void)hipEventCreate(&ctx->ev0);
  (void)hipEventCreate(&ctx->ev1);
  if (hipMalloc(&ctx->d_counter, 64) != hipSuccess || hipMalloc(&ctx->d_out, 16 * sizeof(double)) != hipSuccess) {
    svr_destroy(ctx);
    return (int)hipErrorOutOfMemory;
  }
  {
    // LDS accumulator of the tiled scatter: everything the CU has minus the static part
    int lds_max = 0;
    (void)hipDeviceGetAttribute(&ctx->n_cu, hipDeviceAttributeMultiprocessorCount, device);
    if (ctx->n_cu <= 0) ctx->n_cu = 256;
    (void)hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device);
    if (lds_max <= 0) lds_max = 65536;
    int dyn = lds_max - 1024;
    dyn -= 16384;  // static LDS of the kernels (pixel table, unit lists, partial sums: fwd_unit_kernel<GAUSS1> 14.6 KiB)
    const void *big_lds[] = {reinterpret_cast<const void *>(back_tiled_kernel),
                             reinterpret_cast<const void *>(fwd_unit_kernel<false>), reinterpret_cast<const void *>(fwd_unit_kernel<true>),
                             reinterpret_cast<const void *>(fwd_unit_kernel<false, PSF_SUPPORT, false, true>),
                             reinterpret_cast<const void *>(fwd_unit_kernel<false, PVR_N, true>), reinterpret_cast<const void *>(fwd_unit_kernel<true, PVR_N, true>),
                             reinterpret_cast<const void *>(back_wave_kernel<>), reinterpret_cast<const void *>(back_wave_kernel<PVR_N, true>),
                             reinterpret_cast<const void *>(back_wave_kernel<PSF_SUPPORT, false, true>), reinterpret_cast<const void *>(back_wave_kernel<PVR_N, true, true>),
                             reinterpret_cast<const void *>(fwd_unit_kernel<false, PVR_N, true, true>), reinterpret_cast<const void *>(fwd_unit_kernel<true, PVR_N, true, true>),
                             reinterpret_cast<const void *>(fwd_unit_kernel<true, PSF_SUPPORT, false, true>),
                             reinterpret_cast<const void *>(back_slot_kernel<8>), reinterpret_cast<const void *>(back_slot_kernel<8, PVR_N, true>)};
    bool ok = true;
    for (const void *f : big_lds) ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, dyn) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      dyn = 65536 - 1024 - 16384;  // the default 64 KiB minus the kernels' static LDS
    }
    ctx->tile_cap = dyn / (2 * (int)sizeof(float));
  }
  // SVR_TILE_PIN="GWxGH,SWxSH,BOX": the gather's tile, the tiled scatter's tile and its LDS box fixed instead of timed, so that a
  // run can be repeated with the same shapes (on every rank, in every process); bench.py prints the shapes in config.tuned
  if (const char *v = getenv("SVR_CELL_ORDER")) ctx->cell_order = std::min(20, std::max(0, atoi(v)));   // (experiments: the options' defaults)
  if (const char *v = getenv("SVR_CELL_BALANCE")) ctx->cell_balance = std::min(1024, std::max(0, atoi(v)));
  if (const char *pin = getenv("SVR_TILE_PIN")) {
    int gw, gh, sw, sh, box;
    if (sscanf(pin, "%dx%d,%dx%d,%d", &gw, &gh, &sw, &sh, &box) == 5 && gw > 0 && gh > 0 && gw * gh <= 32 && sw > 0 && sh > 0 && sw * sh <= 64 && box >= 1024) {
      ctx->fwd_tw = gw; ctx->fwd_th = gh; ctx->fwd_tile_user = true;
      ctx->tile_w = sw; ctx->tile_h = sh; ctx->tile_user = true;
      ctx->wave_cap = box; ctx->wave_cap_user = true;
    } else {
      fprintf(stderr, "svr_create: SVR_TILE_PIN=\"%s\" ignored (expected e.g. 6x5,6x4,2096)\n", pin);
    }
  }
  *out = ctx;
  return SVR_OK;
}

int svr_set_option(svr_ctx *ctx, const char *name, int value) {
  SVR_ENTER(ctx);
  if (!ctx || !name) return SVR_E_ARG;
  // Round 6: the scatter generations before the wave-owned tile kernel (0 = a wavefront per pixel with device atomics per tap, 1 = LDS atomics per tile,
  // 3 = the workgroup's slot kernel as a mode of its own) and the wave-per-pixel gather (fwd_mode 0) stay in the library for the tests that pin the
  // generations against each other and as the tile path's own re-run kernels -- asking for one by option needs "legacy_kernels" 1 first.
  if (!strcmp(name, "legacy_kernels")) { ctx->legacy_ok = value != 0; return SVR_OK; }
  if (!strcmp(name, "back_mode")) {
    if (value < 0 || value > 5 || value == 2) return fail(ctx, SVR_E_ARG, "back_mode: 5 (cells), 4 (tile fallback); 0 / 1 / 3 behind legacy_kernels");
    if (value < 4 && !ctx->legacy_ok && !ctx->pvr) return fail(ctx, SVR_E_ARG, "back_mode 0 / 1 / 3: set option legacy_kernels 1 first (kept for the tests; the product runs 5 with 4 as its fallback)");
    ctx->back_mode = value; ctx->back_mode_user = true; return SVR_OK;
  }
  if (!strcmp(name, "sr_no_wait")) { ctx->sr_no_wait = value != 0; return SVR_OK; }   // svr_superresolution_backproject / _update / svr_slab_finish return without waiting for the device (a sharded run whose collectives share the stream)
  if (!strcmp(name, "reg_tile")) { if (value < -1 || value > 2) return fail(ctx, SVR_E_ARG, "reg_tile: -1 .. 2"); ctx->reg_tile = value; return SVR_OK; }
  if (!strcmp(name, "reg_mode")) { if (value < 0 || value > 1) return fail(ctx, SVR_E_ARG, "reg_mode: 0 or 1"); ctx->reg_mode = value; return SVR_OK; }
  if (!strcmp(name, "fwd_mode")) {                         // 2: cell gather, 1: unit-based gather per slice tile (fallback), 0: wave-per-pixel kernel (legacy_kernels)
    if (value < 0 || value > 2) return fail(ctx, SVR_E_ARG, "fwd_mode: 2 (cells), 1 (tile fallback); 0 behind legacy_kernels");
    if (value == 0 && !ctx->legacy_ok && !ctx->pvr) return fail(ctx, SVR_E_ARG, "fwd_mode 0: set option legacy_kernels 1 first (kept for the tests; the product runs 2 with 1 as its fallback)");
    ctx->fwd_mode = value; ctx->fwd_mode_user = true; return SVR_OK;
  }
  if (!strcmp(name, "pvr_mode")) { ctx->pvr_mode = value; return SVR_OK; }
  if (!strcmp(name, "gauss_mode")) { ctx->gauss_mode = value; return SVR_OK; }
  if (!strcmp(name, "fwd_unit_cap")) { ctx->fwd_unit_cap = std::max(2048, value); return SVR_OK; }
  if (!strcmp(name, "fwd_tile_w") || !strcmp(name, "fwd_tile_h")) {
    int w = !strcmp(name, "fwd_tile_w") ? value : ctx->fwd_tw, h = !strcmp(name, "fwd_tile_h") ? value : ctx->fwd_th;
    if (w < 1 || h < 1 || w * h > 32) return fail(ctx, SVR_E_ARG, "fwd tile must hold 1..32 pixels");   // FWDU_MAXPIX: the gather's pixel tables
    ctx->fwd_tw = w; ctx->fwd_th = h; ctx->psf_list_valid = false;
    ctx->fwd_tile_user = true;                           // an explicit shape switches the tuning off
    return SVR_OK;
  }
  if (!strcmp(name, "fwd_autotune")) {
    ctx->fwd_autotune = value ? 1 : 0;
    ctx->fwd_tune_pending = ctx->back_tune_pending = value != 0;
    return SVR_OK;
  }
  if (!strcmp(name, "pvr")) { ctx->pvr = value ? 1 : 0; if (!ctx->coeff_user) ctx->coeff_mode = value ? 0 : 1; ctx->sc_dirty = true; ctx->psf_list_valid = false; ctx->coeff_valid = false; cell_invalidate(ctx); free_dev(ctx->d_coeff); ctx->coeff_cap = 0; return SVR_OK; }
  if (!strcmp(name, "coeff_lazy")) { ctx->coeff_lazy = value ? 1 : 0; return SVR_OK; }
  if (!strcmp(name, "coeff_invalidate")) { ctx->coeff_valid = false; return SVR_OK; }   // (what a new slice geometry does to the table: bench.py's outer iterations)
  if (!strcmp(name, "coeff_table")) {
    if ((value ? 1 : 0) != ctx->coeff_mode) ctx->fwd_tune_pending = ctx->back_tune_pending = ctx->fwd_autotune != 0;   // other shapes win
    ctx->coeff_mode = value ? 1 : 0;
    ctx->coeff_user = true;
    if (!value) { free_dev(ctx->d_coeff); ctx->coeff_cap = 0; ctx->coeff_valid = false; cell_invalidate(ctx); }
    return SVR_OK;
  }
  if (!strcmp(name, "cell_band")) {
    if (value < 0 || value > 10) return fail(ctx, SVR_E_ARG, "cell_band: 0..10 (a run holds 2^cell_band centre planes)");
    ctx->cell_band = value;
    cell_invalidate(ctx);
    return SVR_OK;
  }
  if (!strcmp(name, "cell_qx")) {
    if (value != 1 && value != 2 && value != 4) return fail(ctx, SVR_E_ARG, "cell_qx: 1, 2 or 4");
    ctx->cell_qx = value;
    cell_invalidate(ctx);
    return SVR_OK;
  }
  if (!strcmp(name, "cell_w") || !strcmp(name, "cell_h") || !strcmp(name, "cell_gw") || !strcmp(name, "cell_gh")) {
    if (value < 0 || value > 32) return fail(ctx, SVR_E_ARG, "cell_w / cell_h / cell_gw / cell_gh: 1..32, 0 = by the pixel density");
    (!strcmp(name, "cell_w") ? ctx->cell_w : !strcmp(name, "cell_h") ? ctx->cell_h : !strcmp(name, "cell_gw") ? ctx->cell_gw : ctx->cell_gh) = value;
    cell_invalidate(ctx);
    return SVR_OK;
  }
  if (!strcmp(name, "cell_order")) {
    if (value < 0 || value > 20) return fail(ctx, SVR_E_ARG, "cell_order: 0..20");
    ctx->cell_order = value;
    cell_invalidate(ctx);
    return SVR_OK;
  }
  if (!strcmp(name, "cell_combine")) { if (value < 0 || value > 2) return fail(ctx, SVR_E_ARG, "cell_combine: 0, 1 or 2"); ctx->cell_combine = value; return SVR_OK; }
  if (!strcmp(name, "cell_balance")) {
    if (value < 0 || value > 1024) return fail(ctx, SVR_E_ARG, "cell_balance: 0..1024");
    ctx->cell_balance = value;
    cell_invalidate(ctx);
    return SVR_OK;
  }
  if (!strcmp(name, "cell_split")) {
    if (value < 1 || value > 32) return fail(ctx, SVR_E_ARG, "cell_split: 1..32");
    ctx->cell_split = value;
    cell_invalidate(ctx);
    return SVR_OK;
  }
  if (!strcmp(name, "dbg_back")) { ctx->dbg_back = value; return SVR_OK; }
  if (!strcmp(name, "dbg_fwd_lds")) { ctx->dbg_fwd_lds = value; return SVR_OK; }
  if (!strcmp(name, "wave_groups")) { ctx->wave_groups = std::max(1, value); return SVR_OK; }
  if (!strcmp(name, "wave_cap")) { ctx->wave_cap = std::max(1024, value); ctx->wave_cap_user = true; return SVR_OK; }
  if (!strcmp(name, "reg_blind")) { ctx->reg_blind = std::max(0, value); return SVR_OK; }
  if (!strcmp(name, "tune_tiles")) { ctx->tune_tiles = std::max(0, value); return SVR_OK; }
  if (!strcmp(name, "reg_red_threads")) {
    if (value != 0 && value != 256 && value != 1024) return fail(ctx, SVR_E_ARG, "reg_red_threads: 0, 256 or 1024");
    ctx->reg_red_threads = value;
    return SVR_OK;
  }
  if (!strcmp(name, "reg_batch")) { ctx->reg_batch = value ? 1 : 0; return SVR_OK; }
  if (!strcmp(name, "pvr_reg_levels")) { ctx->pvr_reg_levels = std::min(3, std::max(1, value)); return SVR_OK; }
  if (!strcmp(name, "pvr_reg_steps")) { ctx->pvr_reg_steps = std::max(1, value); return SVR_OK; }
  if (!strcmp(name, "pvr_reg_iterations")) { ctx->pvr_reg_iterations = std::max(1, value); return SVR_OK; }
  if (!strcmp(name, "tile_w") || !strcmp(name, "tile_h")) {
    int w = !strcmp(name, "tile_w") ? value : ctx->tile_w, h = !strcmp(name, "tile_h") ? value : ctx->tile_h;
    if (w < 1 || h < 1 || w * h > 64) return fail(ctx, SVR_E_ARG, "tile_w * tile_h must be in 1..64");
    ctx->tile_w = w; ctx->tile_h = h;
    if (!ctx->in_tune) ctx->tile_user = true;            // an explicit shape switches the tuning off
    if (ctx->np) {
      free_dev(ctx->d_tiles); free_dev(ctx->d_tiles_fb); free_dev(ctx->d_tiles_fb2);
      ctx->tiles_x = (int)((ctx->sx + w - 1) / w);
      ctx->tiles_y = (int)((ctx->sy + h - 1) / h);
      const size_t nb = (size_t)ctx->tiles_x * ctx->tiles_y * ctx->ns * sizeof(uint32_t);
      HIPCHK(hipMalloc(&ctx->d_tiles, nb));
      HIPCHK(hipMalloc(&ctx->d_tiles_fb, nb));
      HIPCHK(hipMalloc(&ctx->d_tiles_fb2, nb));
      ctx->psf_list_valid = false;
    }
    return SVR_OK;
  }
  return fail(ctx, SVR_E_ARG, std::string("unknown option ") + name);
}

int svr_get_option(svr_ctx *ctx, const char *name, int *value) {
  SVR_ENTER(ctx);
  if (!ctx || !name || !value) return SVR_E_ARG;
  int csw, csh, cgw, cgh;
  cell_sizes(ctx, csw, csh, cgw, cgh);                   // the cell sizes in effect (0 = automatic resolved)
  const struct { const char *n; int v; } tab[] = {
      {"back_mode", back_mode_eff(ctx)}, {"reg_mode", ctx->reg_mode}, {"fwd_mode", ctx->fwd_mode}, {"gauss_mode", ctx->gauss_mode}, {"pvr_mode", ctx->pvr_mode},
      {"pvr", ctx->pvr}, {"coeff_table", ctx->coeff_mode}, {"coeff_lazy", ctx->coeff_lazy}, {"coeff_valid", ctx->coeff_valid ? 1 : 0}, {"tile_w", ctx->tile_w}, {"tile_h", ctx->tile_h},
      {"fwd_tile_w", ctx->fwd_tw}, {"fwd_tile_h", ctx->fwd_th}, {"wave_cap", ctx->wave_cap}, {"cell_w", csw}, {"cell_h", csh}, {"cell_gw", cgw}, {"cell_gh", cgh}, {"cell_split", ctx->cell_split}, {"cell_order", ctx->cell_order}, {"cell_balance", ctx->cell_balance}, {"cell_combine", ctx->cell_combine}, {"fwd_autotune", ctx->fwd_autotune}, {"cell_qx", ctx->cell_qx}, {"fwd_unit_cap", ctx->fwd_unit_cap}, {"reg_batch", ctx->reg_batch}, {"reg_blind", ctx->reg_blind}};
  for (const auto &e : tab)
    if (!strcmp(name, e.n)) { *value = e.v; return SVR_OK; }
  return fail(ctx, SVR_E_ARG, std::string("unknown option ") + name);
}

void svr_destroy(svr_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  free_volume(ctx);
  free_slices(ctx);
  free_dev(ctx->d_mask); free_dev(ctx->d_pair_pack);
  free_dev(ctx->d_bias); free_dev(ctx->d_wb); free_dev(ctx->d_wr); free_dev(ctx->d_buffer);
  free_dev(ctx->d_bias_vol); free_dev(ctx->d_volume_weights); free_dev(ctx->d_maskC); free_dev(ctx->d_mbuf);
  free_dev(ctx->d_reg_targets);
  free_dev(ctx->d_reg_source);
  free_dev(ctx->d_pyr_full[0]); free_dev(ctx->d_pyr_full[1]); free_dev(ctx->d_pyr_a); free_dev(ctx->d_pyr_b); free_dev(ctx->d_pyr_meta);
  free_dev(ctx->d_ncc_idx); free_dev(ctx->d_ncc_m); free_dev(ctx->d_ncc_s);
  free_dev(ctx->d_coeff); free_dev(ctx->d_coeff_id); free_dev(ctx->d_coeff_order);
  reg_free(ctx->reg);
  cell_free(ctx->cell);
  cell_free(ctx->cell_g);
  free_dev(ctx->d_cellc);
  slab_free(ctx->slab);
  slice_em_free(ctx->sem);
  free_dev(ctx->d_spx);
  free_dev(ctx->d_counter);
  free_dev(ctx->d_out);
  free_dev(ctx->d_em_all);
  timers_resolve(ctx);
  if (ctx->ev_open) (void)hipEventDestroy(ctx->ev_open);
  for (hipEvent_t e : ctx->ev_free) (void)hipEventDestroy(e);
  ctx->ev_free.clear();
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char *svr_last_error(const svr_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int svr_set_flags(svr_ctx *ctx, int disable_bias_correction, int debug_gpu) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  ctx->disable_bias = disable_bias_correction != 0;
  ctx->debug_gpu = debug_gpu != 0;
  return SVR_OK;
}

int svr_set_stream(svr_ctx *ctx, void *hip_stream) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  ctx->stream = (hipStream_t)hip_stream;
  ctx->own_stream = false;
  return SVR_OK;
}

void *svr_get_stream(svr_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
int svr_stream_sync(svr_ctx *ctx) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}
int svr_device(svr_ctx *ctx) { return ctx ? ctx->device : -1; }
int svr_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int svr_init_reconstruction_volume(svr_ctx *ctx, const uint32_t size[3], const float dim[3],
                                   const float *data, float sigma_bias) {
  SVR_ENTER(ctx);
  if (!ctx || !size || !dim) return SVR_E_ARG;
  (void)sigma_bias;
  HIPCHK(hipSetDevice(ctx->device));
  size_t nv = (size_t)size[0] * size[1] * size[2];
  if (nv == 0 || nv >= 0xFFFFFFFFull) return fail(ctx, SVR_E_ARG, "volume size out of range");
  free_volume(ctx);
  free_dev(ctx->d_bias_vol); free_dev(ctx->d_volume_weights); free_dev(ctx->d_maskC); free_dev(ctx->d_mbuf);
  ctx->vx = size[0]; ctx->vy = size[1]; ctx->vz = size[2];
  memcpy(ctx->vdim, dim, 3 * sizeof(float));
  ctx->nv = nv;
  ctx->coeff_valid = false; cell_invalidate(ctx);
  HIPCHK(hipMalloc(&ctx->d_recon_volw, 2 * nv * sizeof(float)));
  HIPCHK(hipMalloc(&ctx->d_addon_cmap, 2 * nv * sizeof(float)));
  HIPCHK(hipMalloc(&ctx->d_recon_new, nv * sizeof(float)));
  ctx->recon_cur = ctx->d_recon_volw;
  ctx->prep_pending = false; ctx->cmap_from_scatter = false;
  ctx->vol_clean[0] = ctx->vol_clean[1] = false;
  if (ctx->slab) ctx->slab->valid = false;
  HIPCHK(hipMemsetAsync(ctx->d_recon_volw, 0, 2 * nv * sizeof(float), ctx->stream));   // RC.cu:1199-1229
  HIPCHK(hipMemsetAsync(ctx->d_addon_cmap, 0, 2 * nv * sizeof(float), ctx->stream));
  if (data) HIPCHK(hipMemcpyAsync(ctx->recon(), data, nv * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}

int svr_set_mask(svr_ctx *ctx, const uint32_t size[3], const float dim[3], const float *data,
                 float sigma_bias) {
  SVR_ENTER(ctx);
  if (!ctx || !size || !data) return SVR_E_ARG;
  (void)dim;   // the blurred maskC_ (RC.cu:1129-1157) is built lazily: it only feeds NormaliseBias
  NEED(ctx->nv > 0, "InitReconstructionVolume first");
  if ((size_t)size[0] * size[1] * size[2] != ctx->nv || size[0] != ctx->vx || size[1] != ctx->vy)
    return fail(ctx, SVR_E_ARG, "mask grid differs from the reconstruction volume");
  free_dev(ctx->d_mask);
  HIPCHK(hipMalloc(&ctx->d_mask, ctx->nv * sizeof(float)));
  HIPCHK(hipMemcpyAsync(ctx->d_mask, data, ctx->nv * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->have_mask = true;
  if (ctx->slab) ctx->slab->valid = false;
  ctx->vol_clean[0] = ctx->vol_clean[1] = false;      // "zero outside the dilated mask" was a statement about the old mask
  ctx->mask_sigma_bias = sigma_bias;
  ctx->maskC_valid = false;
  {
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {-1, -1, -1};
    const size_t sxy = (size_t)ctx->vx * ctx->vy;
    for (uint32_t z = 0; z < ctx->vz; ++z)
      for (uint32_t y = 0; y < ctx->vy; ++y) {
        const float *row = data + z * sxy + (size_t)y * ctx->vx;
        int x0 = -1, x1 = -1;
        for (uint32_t x = 0; x < ctx->vx; ++x)
          if (row[x] != 0.0f) { if (x0 < 0) x0 = (int)x; x1 = (int)x; }
        if (x0 >= 0) {
          lo[0] = std::min(lo[0], x0); hi[0] = std::max(hi[0], x1);
          lo[1] = std::min(lo[1], (int)y); hi[1] = std::max(hi[1], (int)y);
          lo[2] = std::min(lo[2], (int)z); hi[2] = std::max(hi[2], (int)z);
        }
      }
    ctx->mbox_valid = hi[0] >= 0;
    for (int k = 0; k < 3; ++k) { ctx->mbox_lo[k] = lo[k]; ctx->mbox_hi[k] = hi[k]; }
  }
  return SVR_OK;
}

// ---- the part of a volume pair a sharded run has to exchange -------------------------------------------------------------
__global__ void k_pair_pack(const float *vol, float *packed, int nvols, size_t nv, int vx, int vy, int lx, int ly, int lz, int bx, int by, int bz, int unpack,
                            float *vol_out) {
  const size_t nb = (size_t)bx * by * bz;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nb * nvols) return;
  const int v = (int)(i / nb);
  const size_t r = i - (size_t)v * nb;
  const int z = (int)(r / ((size_t)bx * by)), y = (int)((r / bx) % by), x = (int)(r % bx);
  const size_t g = (size_t)v * nv + (size_t)(x + lx) + (size_t)(y + ly) * vx + (size_t)(z + lz) * vx * vy;
  if (unpack) vol_out[g] = packed[i];
  else packed[i] = vol[g];
}
int svr_pair_pack(svr_ctx *ctx, int which, size_t n_floats, void **packed, size_t *n_packed) {
  SVR_ENTER(ctx);
  if (!ctx || !packed || !n_packed) return SVR_E_ARG;
  *packed = nullptr; *n_packed = 0;
  float *vol = static_cast<float *>(svr_device_ptr(ctx, which));
  if (!vol || !ctx->mbox_valid || !ctx->nv || n_floats % ctx->nv) return SVR_OK;          // nothing to gain: the caller reduces the whole buffer
  const int nvols = (int)(n_floats / ctx->nv);
  const int bx = ctx->mbox_hi[0] - ctx->mbox_lo[0] + 1, by = ctx->mbox_hi[1] - ctx->mbox_lo[1] + 1, bz = ctx->mbox_hi[2] - ctx->mbox_lo[2] + 1;
  const size_t nb = (size_t)bx * by * bz;
  if (nb * 5 > ctx->nv * 4) return SVR_OK;                // the box is more than 80 % of the volume: not worth two copies
  if (nb * nvols > ctx->pair_pack_cap) {
    free_dev(ctx->d_pair_pack);
    ctx->pair_pack_cap = 0;
    HIPCHK(hipMalloc(&ctx->d_pair_pack, nb * nvols * sizeof(float)));
    ctx->pair_pack_cap = nb * nvols;
  }
  hipLaunchKernelGGL(k_pair_pack, dim3(nblk(nb * nvols)), dim3(256), 0, ctx->stream, vol, ctx->d_pair_pack, nvols, ctx->nv, (int)ctx->vx, (int)ctx->vy,
                     ctx->mbox_lo[0], ctx->mbox_lo[1], ctx->mbox_lo[2], bx, by, bz, 0, (float *)nullptr);
  KCHK("k_pair_pack");
  *packed = ctx->d_pair_pack;
  *n_packed = nb * nvols;
  return SVR_OK;
}
int svr_pair_unpack(svr_ctx *ctx, int which, size_t n_floats) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  float *vol = static_cast<float *>(svr_device_ptr(ctx, which));
  NEED(vol && ctx->mbox_valid && ctx->d_pair_pack && ctx->nv && n_floats % ctx->nv == 0, "svr_pair_pack first");
  const int nvols = (int)(n_floats / ctx->nv);
  const int bx = ctx->mbox_hi[0] - ctx->mbox_lo[0] + 1, by = ctx->mbox_hi[1] - ctx->mbox_lo[1] + 1, bz = ctx->mbox_hi[2] - ctx->mbox_lo[2] + 1;
  const size_t nb = (size_t)bx * by * bz;
  hipLaunchKernelGGL(k_pair_pack, dim3(nblk(nb * nvols)), dim3(256), 0, ctx->stream, (const float *)nullptr, ctx->d_pair_pack, nvols, ctx->nv, (int)ctx->vx,
                     (int)ctx->vy, ctx->mbox_lo[0], ctx->mbox_lo[1], ctx->mbox_lo[2], bx, by, bz, 1, vol);
  KCHK("k_pair_unpack");
  return SVR_OK;
}

int svr_init_storage_volumes(svr_ctx *ctx, const uint32_t size[3], const float dim[3]) {
  SVR_ENTER(ctx);
  if (!ctx || !size) return SVR_E_ARG;
  (void)dim;
  HIPCHK(hipSetDevice(ctx->device));
  size_t np = (size_t)size[0] * size[1] * size[2];
  // slice-grid indices are 32-bit; every launch that takes a wavefront per pixel or per tile goes out in pieces of < 2^32
  // work-items (in_pieces), like the reference's MAX_SLICES_PER_RUN chunks (RC.cu:2207-2219, 2414-2432, 2701-2716)
  if (np == 0 || np >= (1ull << 31)) return fail(ctx, SVR_E_ARG, "slice grid out of range: 1 .. 2^31 - 1 pixels per context (shard the slices over more ranks)");
  free_slices(ctx);
  free_dev(ctx->d_bias); free_dev(ctx->d_wb); free_dev(ctx->d_wr); free_dev(ctx->d_buffer);
  ctx->sx = size[0]; ctx->sy = size[1]; ctx->ns = size[2];
  ctx->np = np;
  ctx->have_slices = false; ctx->have_scales = false; ctx->have_dims = false; ctx->have_mats = false;
  ctx->sc_dirty = true; ctx->psf_list_valid = false; ctx->n_active = ctx->n_psf = 0;
  ctx->coeff_valid = false; cell_invalidate(ctx); free_dev(ctx->d_coeff_id); free_dev(ctx->d_coeff); ctx->coeff_cap = 0;
  const size_t fb = np * sizeof(float);
  HIPCHK(hipMalloc(&ctx->d_slices, fb));
  HIPCHK(hipMalloc(&ctx->d_weights, fb));
  HIPCHK(hipMalloc(&ctx->d_simslices, fb));
  HIPCHK(hipMalloc(&ctx->d_simweights, fb));
  HIPCHK(hipMalloc(&ctx->d_psf_sums, fb));
  HIPCHK(hipMalloc(&ctx->d_siminside, np));
  HIPCHK(hipMalloc(&ctx->d_voxcount, np * sizeof(int)));
  HIPCHK(hipMalloc(&ctx->d_active, np * sizeof(uint32_t)));
  HIPCHK(hipMalloc(&ctx->d_psf_list, np * sizeof(uint32_t)));
  ctx->tiles_x = (int)((ctx->sx + ctx->tile_w - 1) / ctx->tile_w);
  ctx->tiles_y = (int)((ctx->sy + ctx->tile_h - 1) / ctx->tile_h);
  HIPCHK(hipMalloc(&ctx->d_tiles, (size_t)ctx->tiles_x * ctx->tiles_y * ctx->ns * sizeof(uint32_t)));
  HIPCHK(hipMalloc(&ctx->d_tiles_fb, (size_t)ctx->tiles_x * ctx->tiles_y * ctx->ns * sizeof(uint32_t)));
  HIPCHK(hipMalloc(&ctx->d_tiles_fb2, (size_t)ctx->tiles_x * ctx->tiles_y * ctx->ns * sizeof(uint32_t)));
  HIPCHK(hipMalloc(&ctx->d_scales, ctx->ns * sizeof(float)));
  HIPCHK(hipMalloc(&ctx->d_slice_weights, ctx->ns * sizeof(float)));
  ctx->sem_weights_current = false;
  slice_em_invalidate(ctx->sem);                          // the slice-level EM's arrays and rank ranges were set up for the old slice count
  HIPCHK(hipMalloc(&ctx->d_scales_host_copy, ctx->ns * sizeof(float)));
  HIPCHK(hipMalloc(&ctx->d_tmp_ns, ctx->ns * sizeof(float)));
  HIPCHK(hipMalloc(&ctx->d_slice_inside, ctx->ns));
  HIPCHK(hipMalloc(&ctx->d_sc, ctx->ns * sizeof(SliceConst)));
  ctx->chunks = (int)(((size_t)ctx->sx * ctx->sy + CHUNK_PIX - 1) / CHUNK_PIX);
  HIPCHK(hipMalloc(&ctx->d_partial, (size_t)ctx->ns * ctx->chunks * REDUCE_MAXK * sizeof(double)));    // (reduce_partials' guard is the same constant)
  HIPCHK(hipMalloc(&ctx->d_per_slice, (size_t)ctx->ns * REDUCE_MAXK * sizeof(double)));
  // RC.cu:1555-1568: everything cleared once at allocation (v_PSF_sums is never cleared again)
  HIPCHK(hipMemsetAsync(ctx->d_slices, 0, fb, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_weights, 0, fb, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_simslices, 0, fb, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_simweights, 0, fb, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_psf_sums, 0, fb, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_siminside, 0, np, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_slice_inside, 0, ctx->ns, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_voxcount, 0, np * sizeof(int), ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}

int svr_fill_slices(svr_ctx *ctx, const float *sdata, const int *sizes_x, const int *sizes_y) {
  SVR_ENTER(ctx);
  if (!ctx || !sdata) return SVR_E_ARG;
  (void)sizes_x; (void)sizes_y;   // only used by the reference's dead code (RC.cu:3252-3253)
  NEED(ctx->np > 0, "initStorageVolumes first");
  HIPCHK(hipMemcpyAsync(ctx->d_slices, sdata, ctx->np * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  ctx->have_slices = true;
  ctx->psf_list_valid = false;
  ctx->coeff_valid = false; cell_invalidate(ctx);
  return build_list(ctx, false);
}

int svr_set_slice_dims(svr_ctx *ctx, const float *slice_dims, float quality_factor) {
  SVR_ENTER(ctx);
  if (!ctx || !slice_dims) return SVR_E_ARG;
  NEED(ctx->ns > 0, "initStorageVolumes first");
  ctx->slice_dims.assign(slice_dims, slice_dims + 3 * (size_t)ctx->ns);
  ctx->quality_factor = quality_factor;   // only sizes the (unused) finite-support dim, RC.cu:772-784
  ctx->fwd_tune_pending = ctx->back_tune_pending = ctx->fwd_autotune != 0;   // new slice geometry: time the tile shapes again
  ctx->have_dims = true;
  ctx->sc_dirty = true;
  return SVR_OK;
}

int svr_set_slice_matrices(svr_ctx *ctx, const float *T, const float *Tinv, const float *i2w_init,
                           const float *w2i_init, const float *i2w, const float *w2i,
                           const float recon_i2w[16], const float recon_w2i[16]) {
  SVR_ENTER(ctx);
  if (!ctx || !T || !Tinv || !i2w || !w2i || !recon_i2w || !recon_w2i) return SVR_E_ARG;
  (void)i2w_init; (void)w2i_init;   // registration-only in the reference (RC.cu:3486)
  NEED(ctx->ns > 0, "initStorageVolumes first");
  const size_t n = 16 * (size_t)ctx->ns;
  ctx->mT.assign(T, T + n);
  ctx->mTinv.assign(Tinv, Tinv + n);
  ctx->mI2W.assign(i2w, i2w + n);
  ctx->mW2I.assign(w2i, w2i + n);
  memcpy(ctx->reconI2W, recon_i2w, 16 * sizeof(float));
  memcpy(ctx->reconW2I, recon_w2i, 16 * sizeof(float));
  ctx->have_mats = true;
  ctx->sc_dirty = true;
  return SVR_OK;
}

int svr_generate_psf_volume(svr_ctx *ctx, const float *cpu_psf, const uint32_t psf_size[3],
                            const float slice_voxel_dim[3], const float psf_dim[3],
                            const float psf_i2w[16], const float psf_w2i[16], float quality_factor) {
  SVR_ENTER(ctx);
  if (!ctx || !psf_size || !psf_i2w) return SVR_E_ARG;
  (void)cpu_psf; (void)slice_voxel_dim; (void)psf_dim; (void)psf_w2i;
  // d_PSFI2W * ((PSFsize - 1) * 0.5f)   RC.cu:172
  float v[3] = {((float)psf_size[0] - 1) * 0.5f, ((float)psf_size[1] - 1) * 0.5f, ((float)psf_size[2] - 1) * 0.5f};
  matvec3_host(psf_i2w, v, ctx->psf_c0);
  ctx->coeff_valid = false; cell_invalidate(ctx);                               // the taps' residuals carry c0
  ctx->quality_factor = quality_factor;
  ctx->have_psf = true;
  return SVR_OK;
}

int svr_update_scale_vector(svr_ctx *ctx, const float *scales, const float *slice_weights) {
  SVR_ENTER(ctx);
  if (!ctx || !scales || !slice_weights) return SVR_E_ARG;
  NEED(ctx->ns > 0, "initStorageVolumes first");
  ctx->h_slice_weights.assign(slice_weights, slice_weights + ctx->ns);
  int r = upload_ns(ctx, ctx->d_scales, scales, ctx->mir_scales);
  if (r) return r;
  r = upload_ns(ctx, ctx->d_scales_host_copy, scales, ctx->mir_scales_copy);    // the device's copy of h_scales (below)
  if (r) return r;
  r = upload_ns(ctx, ctx->d_slice_weights, slice_weights, ctx->mir_slice_weights);
  if (r) return r;
  ctx->have_scales = true;
  return SVR_OK;
}

int svr_update_slice_weights(svr_ctx *ctx, const float *slice_weights) {
  SVR_ENTER(ctx);
  if (!ctx || !slice_weights) return SVR_E_ARG;
  NEED(ctx->have_scales, "UpdateScaleVector first");
  ctx->h_slice_weights.assign(slice_weights, slice_weights + ctx->ns);
  return upload_ns(ctx, ctx->d_slice_weights, slice_weights, ctx->mir_slice_weights);
}

int svr_update_reconstructed(svr_ctx *ctx, const uint32_t size[3], const float *data) {
  SVR_ENTER(ctx);
  if (!ctx || !size || !data) return SVR_E_ARG;
  NEED(ctx->nv > 0, "InitReconstructionVolume first");
  if ((size_t)size[0] * size[1] * size[2] != ctx->nv) return fail(ctx, SVR_E_ARG, "size mismatch");
  HIPCHK(hipMemcpyAsync(ctx->recon(), data, ctx->nv * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}

int svr_sync_cpu(svr_ctx *ctx, float *out) {
  SVR_ENTER(ctx);
  if (!ctx || !out) return SVR_E_ARG;
  NEED(ctx->nv > 0, "InitReconstructionVolume first");
  HIPCHK(hipMemcpyAsync(out, ctx->recon(), ctx->nv * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}

// Reconstruction::updateStackSizes (RC.cuh:210): the sizes are stored; nothing in the reference reads them back
int svr_update_stack_sizes(svr_ctx *ctx, const uint32_t *sizes3, int n_stacks) {
  SVR_ENTER(ctx);
  if (!ctx || n_stacks < 0 || (n_stacks > 0 && !sizes3)) return SVR_E_ARG;
  ctx->stack_sizes.assign(sizes3, sizes3 + 3 * (size_t)n_stacks);
  return SVR_OK;
}

// Reconstruction::combineWeights (RC.cu:5091-5097): the bias path's accumulated volume weights (dev_volume_weights_) of device 0;
// zeros while no NormaliseBias has run (the reference's buffer is cleared at allocation, RC.cu:1214)
int svr_combine_weights(svr_ctx *ctx, float *out) {
  SVR_ENTER(ctx);
  if (!ctx || !out) return SVR_E_ARG;
  NEED(ctx->nv > 0, "InitReconstructionVolume first");
  if (!ctx->d_volume_weights) { memset(out, 0, ctx->nv * sizeof(float)); return SVR_OK; }
  HIPCHK(hipMemcpyAsync(out, ctx->d_volume_weights, ctx->nv * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}

int svr_get_vol_weights(svr_ctx *ctx, float *out) {
  SVR_ENTER(ctx);
  if (!ctx || !out) return SVR_E_ARG;
  NEED(ctx->nv > 0, "InitReconstructionVolume first");
  HIPCHK(hipMemcpyAsync(out, ctx->volw(), ctx->nv * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}

// ---- Gaussian reconstruction -----------------------------------------------------------
int svr_gaussian_reconstruction_local(svr_ctx *ctx) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  int r = ready(ctx);
  if (r) return r;
  const size_t fb = ctx->np * sizeof(float);
  // RC.cu:2401-2411.  PVR: ReconVolume::reset() clears only the volume and its weights
  // (R2/include/reconVolume.cuh:93-98); the patch buffers keep their values between outer iterations.
  if (!ctx->pvr) {
    HIPCHK(hipMemsetAsync(ctx->d_weights, 0, fb, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_simweights, 0, fb, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_simslices, 0, fb, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_siminside, 0, ctx->np, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_slice_inside, 0, ctx->ns, ctx->stream));
  }
  HIPCHK(hipMemsetAsync(ctx->d_voxcount, 0, ctx->np * sizeof(int), ctx->stream));
  ctx->recon_cur = ctx->d_recon_volw;      // recon | volw as one allocation again (the pair a sharded run all-reduces); the old volume is not read
  HIPCHK(hipMemsetAsync(ctx->d_recon_volw, 0, 2 * ctx->nv * sizeof(float), ctx->stream));
  const bool tiled = ctx->gauss_mode == 1 && (!ctx->pvr || ctx->pvr_mode == 1);
  if (tiled) {
    r = ensure_coeff_unless_lazy(ctx);
    if (r) return r;
  }
  PsfArgs a = make_args(ctx);
  a.list = ctx->d_active;
  a.n = ctx->n_active;
  if (tiled && ctx->coeff_mode && ctx->coeff_valid && ctx->coeff_full) give_coeff(ctx, a);   // (every pixel with s != -1: a table written by a gather does not hold them all)
  bool stored = false;
  ScopedTimer t(ctx, SVR_T_GAUSS);
  if (a.n && tiled) {
    // pass 1 = the unit-based walk of the gather (sume, gate, v_PSF_sums, voxel-count flag), pass 2 = the scatter of the
    // back-projection with {recon|volw} as targets and unit voxel / slice weights
    if (!ctx->d_gauss_flag) HIPCHK(hipMalloc(&ctx->d_gauss_flag, ctx->np));
    HIPCHK(hipMemsetAsync(ctx->d_gauss_flag, 0, ctx->np, ctx->stream));
    // pass 1 keeps 4-pixel-wide tiles when the gather of the SR iterations chose 6 x 4 (measured: 6 x 4 costs pass 1 1.6 ms on P4)
    const int gtw = ctx->fwd_tw == 6 ? 4 : ctx->fwd_tw, gth = ctx->fwd_th;
    const int ftx = (int)((ctx->sx + gtw - 1) / gtw), fty = (int)((ctx->sy + gth - 1) / gth);
    const size_t max_tiles = std::max((size_t)ftx * fty, (size_t)ctx->tiles_x * ctx->tiles_y) * ctx->ns;
    if (max_tiles > ctx->tiles_tmp_cap) {
      free_dev(ctx->d_tiles_tmp);
      HIPCHK(hipMalloc(&ctx->d_tiles_tmp, max_tiles * sizeof(uint32_t)));
      ctx->tiles_tmp_cap = max_tiles;
    }
    uint32_t n1 = 0, n2 = 0;
    TileArgs ta;
    ta.tiles_x = ftx; ta.tiles_y = fty; ta.tw = gtw; ta.th = gth; ta.gauss = 1;
    ta.cap = std::min(ctx->fwd_unit_cap, ctx->tile_cap); ta.dbg = ctx->dbg_back;
    a.flag = nullptr; a.flag_out = ctx->d_gauss_flag;
    // pass 1 follows the gather: over the (cell, plane) items when that is the gather's mode (fwd_mode 2; round 4), per slice tile
    // otherwise (fwd_mode 1, the coefficient table, geometries the cell lists cannot hold).  The same bits.
    bool pass1_cells = false;
    if (ctx->fwd_mode == 2 && !a.coeff) {
      CellState *gcs = nullptr;
      if ((r = cell_prepare_gather(ctx, gcs))) return r;
      if (gcs && gcs->usable) {
        if ((r = launch_cell_gauss1(ctx, *gcs, a))) return r;
        pass1_cells = true;
      } else {
        ctx->note_fallback(2, "pass 1 of the Gaussian reconstruction left the cell path (the cell lists cannot hold this geometry): tile kernel, same bits");
      }
    }
    if (!pass1_cells) {
    HIPCHK(hipMemsetAsync(ctx->d_counter, 0, sizeof(uint32_t), ctx->stream));
    if ((r = build_tile_list(ctx, nullptr, nullptr, ftx, fty, gtw, gth, ctx->d_tiles_tmp, ctx->d_counter))) return r;
    HIPCHK(hipMemcpyAsync(&n1, ctx->d_counter, sizeof(n1), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ta.tiles = ctx->d_tiles_tmp; ta.ntiles = n1;
    if (n1) {
      const size_t lds = (size_t)ta.cap * 2 * sizeof(float);
      launch_fwd_unit<true>(ctx, a, ta, lds);
      KCHK("fwd_unit_kernel<GAUSS1>");
    }
    }
    a.flag = ctx->d_gauss_flag;
    a.addon = ctx->recon(); a.cmap = ctx->volw();       // scatter targets of pass 2 (RC.cu:279-282)
    ta.tiles_x = ctx->tiles_x; ta.tiles_y = ctx->tiles_y; ta.tw = ctx->tile_w; ta.th = ctx->tile_h; ta.dbg = 0;
    bool cells = false;
    if (back_mode_eff(ctx) == 5) {
      if ((r = cell_prepare(ctx))) return r;
      cells = ctx->cell->usable;
      if (!cells) ctx->note_fallback(0, "the scatter left the cell path (the cell lists cannot hold this geometry): back_mode 4, float atomics, last bits depend on the run");
    }
    if (cells) {
      // (coeff_lazy: pass 2 evaluates every unit the SR iterations will read -- it writes the table, and the SimulateSlices that follows streams it)
      if (!a.coeff && !ctx->pvr && (r = coeff_begin_store(ctx, &stored))) return r;
      if (stored) give_coeff(ctx, a);
      r = launch_cell_scatter(ctx, a, 1, ctx->recon(), ctx->volw(), stored);
    }
    else {
      // the tiles of the pixels that passed the gate (the tiled scatters only)
      HIPCHK(hipMemsetAsync(ctx->d_counter, 0, sizeof(uint32_t), ctx->stream));
      if ((r = build_tile_list(ctx, nullptr, ctx->d_gauss_flag, ctx->tiles_x, ctx->tiles_y, ctx->tile_w, ctx->tile_h, ctx->d_tiles_tmp, ctx->d_counter))) return r;
      HIPCHK(hipMemcpyAsync(&n2, ctx->d_counter, sizeof(n2), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      r = launch_scatter(ctx, ctx->pvr ? 4 : std::min(4, std::max(1, back_mode_eff(ctx))), a, ta, ctx->d_tiles_tmp, n2, MODE_GAUSS2);
    }
    if (r) return r;
  } else if (a.n && ctx->pvr) {
    { const int rr = pixel_list_in_pieces(a, [&](const PsfArgs &ap, uint32_t) {
        hipLaunchKernelGGL(pvr_kernel<MODE_GAUSS>, dim3(nblk(ap.n, WAVES_PER_BLOCK)), dim3(WAVES_PER_BLOCK * 64), 0, ctx->stream, ap);
        KCHK("pvr_kernel<GAUSS>");
        return (int)SVR_OK; });
      if (rr) return rr; }
  } else if (a.n) {
    { const int rr = pixel_list_in_pieces(a, [&](const PsfArgs &ap, uint32_t) {
        hipLaunchKernelGGL(psf_kernel<MODE_GAUSS>, dim3(nblk(ap.n, WAVES_PER_BLOCK)), dim3(WAVES_PER_BLOCK * 64), 0, ctx->stream, ap);
        KCHK("psf_kernel<GAUSS>");
        return (int)SVR_OK; });
      if (rr) return rr; }
  }
  t.stop();
  ctx->psf_list_valid = false;
  if (stored) { ctx->coeff_valid = true; ctx->coeff_full = false; }   // (written by pass 2 for the new v_PSF_sums)
  else if (!ctx->coeff_full) ctx->coeff_valid = false;     // new v_PSF_sums: a table written by an earlier pass holds the PSF pixels of that pass
  cell_gf_invalidate(ctx);                                 // ... and the gather's 1 / v_PSF_sums per sorted pixel
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}

int svr_gaussian_reconstruction_finish(svr_ctx *ctx, int *voxel_num) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  NEED(ctx->nv > 0 && ctx->have_slices, "volume / slices not set");
  hipLaunchKernelGGL(k_equalize, dim3(nblk(ctx->nv)), dim3(256), 0, ctx->stream, ctx->recon(), ctx->volw(), ctx->nv);
  KCHK("k_equalize");
  unsigned long long cnt = 0;
  unsigned long long *d_cnt = reinterpret_cast<unsigned long long *>(ctx->d_out);
  HIPCHK(hipMemsetAsync(d_cnt, 0, sizeof(cnt), ctx->stream));
  hipLaunchKernelGGL(k_count_positive, dim3(std::min(nblk(ctx->np), 2048u)), dim3(256), 0, ctx->stream, ctx->d_voxcount, ctx->np, d_cnt);
  KCHK("k_count_positive");
  HIPCHK(hipMemcpyAsync(&cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (voxel_num) *voxel_num = (int)cnt;
  return build_list(ctx, true);
}

int svr_gaussian_reconstruction(svr_ctx *ctx, int *voxel_num) {
  SVR_ENTER(ctx);
  int r = svr_gaussian_reconstruction_local(ctx);
  if (r) return r;
  return svr_gaussian_reconstruction_finish(ctx, voxel_num);
}

// The tile shape of the unit gather (fwd_unit_kernel: fwd_mode 1, and the coefficient table's gather) from the geometry, not from a
// timed trial: what the trials of rounds 1-3 picked, as a rule of the pixel density d = voxel area / pixel area in the slice plane.
// With the table (the pass waits for HBM; 24-32 pixels per tile keep two workgroups per CU): P4 (d 0.72) 6 x 4 2.79 ms against 2.85 for
// 6 x 5 and 3.07 for 4 x 4; S8 (d 0.56) 6 x 5 26.8 against 28.6 / 28.4 for 6 x 4 / 8 x 4; patches (support 12: smaller boxes) 8 x 4
// 8.1 against 9.0-9.5 ms.  On the fly: 6 x 4 (round 2: 4.21 against 4.60 ms for 4 x 4 on P4).  Near-ties all (2-6 %): a rule that
// repeats is worth more than the last per cent -- the table line of round 3 moved by 4 % from run to run with the trials' picks.
double pixel_density(const svr_ctx *ctx) {              // voxel area / pixel area in the slice plane
  double d = 1.0;
  if (ctx->slice_dims.size() >= 3 && ctx->slice_dims[0] > 0 && ctx->slice_dims[1] > 0)
    d = (double)ctx->vdim[0] * ctx->vdim[1] / ((double)ctx->slice_dims[0] * ctx->slice_dims[1]);
  return d;
}
void tile_shape_rule(const svr_ctx *ctx, bool table, int &w, int &h) {
  const double d = pixel_density(ctx);
  if (ctx->pvr) { w = 8; h = 4; }
  else if (table && d < 0.65) { w = 6; h = 5; }
  else { w = 6; h = 4; }
}

// ---- forward projection ----------------------------------------------------------------
int svr_simulate_slices(svr_ctx *ctx, uint8_t *slice_inside) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  int r = ready(ctx);
  if (r) return r;
  r = ensure_psf_list(ctx);
  if (r) return r;
  // Round 6 (coeff_lazy): with the table wanted and not there, THIS pass writes it -- it evaluates every tap of every PSF pixel anyway
  // (fwd_cell_kernel<.., 3>); needs the cell gather.  Otherwise k_coeff_build, as before.
  bool store = false;
  CellState *gcs = nullptr;
  if (ctx->coeff_mode && !ctx->coeff_valid && coeff_lazy_ok(ctx) && ctx->n_psf && ctx->n_active) {
    if ((r = cell_prepare_gather(ctx, gcs))) return r;
    if (gcs->usable && (r = coeff_begin_store(ctx, &store))) return r;
  }
  if (!store) {
    r = ensure_coeff(ctx);
    if (r) return r;
  }
  PsfArgs a = make_args(ctx);
  a.list = ctx->d_psf_list;
  a.n = ctx->n_psf;
  if (store || (ctx->coeff_mode && ctx->coeff_valid && (ctx->pvr ? ctx->pvr_mode == 1 : ctx->fwd_mode >= 1))) give_coeff(ctx, a);
  if (!ctx->pvr && ctx->fwd_mode >= 1 && a.n) {
    if (!ctx->d_volm) HIPCHK(hipMalloc(&ctx->d_volm, ctx->nv * sizeof(float2)));
    hipLaunchKernelGGL(k_pack_volm, dim3(nblk(ctx->nv)), dim3(256), 0, ctx->stream, ctx->recon(), ctx->d_mask, ctx->d_volm, ctx->nv);
    KCHK("k_pack_volm");
    a.volm = ctx->d_volm;
  } else if (ctx->pvr && ctx->pvr_mode == 1 && a.n) {
    if (!ctx->d_volm) HIPCHK(hipMalloc(&ctx->d_volm, ctx->nv * sizeof(float2)));
    hipLaunchKernelGGL(k_pack_volm_pvr, dim3(nblk(ctx->nv)), dim3(256), 0, ctx->stream, ctx->recon(), ctx->d_mask, ctx->d_volm, a.vg);
    KCHK("k_pack_volm_pvr");
    a.volm = ctx->d_volm;
  }
  // fwd_mode 2 (the default for SVR on the fly): the gather over the (cell, plane) items of the scatter without atomics
  bool cells = false;
  // (with the coefficient table: the cell gather for support 12 only -- PVR8spx 24.5 -> 20.7 ms; for support 16 its ring and its 16
  // box values do not fit the registers and the unit gather streams the table faster: P4 2.76 against 3.04 ms)
  // ... and there only on large cells, i.e. fine volumes: PVR4 (9 x 6) 8.1 ms on tiles against 8.5 ms on cells)
  // Round 6: support 16 takes the cell gather too -- the table's rows land in the LDS by LDS-DMA instead of in a ring of registers (fwd_cell_kernel,
  // COEFF == 2): P4 2.39 against 2.84 ms; option fwd_mode 1 keeps the tile kernel reachable
  // ... where the slices' pixels are not much coarser than the voxels (the density of tile_shape_rule): S8 (d 0.56, a 174 GB table) 28.1 ms on the
  // cells whatever their size (6 x 4 .. 12 x 6) against 26.9 ms on 6 x 5 slice tiles
  bool table_on_cells = a.coeff && !ctx->pvr && (store || ctx->fwd_mode_user || pixel_density(ctx) >= 0.65);
  if (a.coeff && ctx->pvr) {
    int sw, sh, gw, gh;
    cell_sizes(ctx, sw, sh, gw, gh);
    table_on_cells = gw * gh >= 96 || ctx->fwd_mode_user;
  }
  if (ctx->fwd_mode == 2 && (!ctx->pvr || ctx->pvr_mode == 1) && (!a.coeff || table_on_cells) && a.n) {
    if ((r = cell_prepare_gather(ctx, gcs))) return r;
    cells = gcs->usable;
    if (!cells) ctx->note_fallback(1, "the gather left the cell path (the cell lists cannot hold this geometry): tile kernel, same bits");
  }
  auto launch_forward = [&]() -> int {
    const bool tiled_ = ctx->pvr ? ctx->pvr_mode == 1 : ctx->fwd_mode >= 1;
    if (cells) {
      const int rr = launch_cell_gather(ctx, *gcs, a, store);
      if (rr) return rr;
      if (store) { ctx->coeff_valid = true; ctx->coeff_full = false; }   // (the PSF pixels' live units: what the SR iterations' passes read)
    } else if (a.n && tiled_) {
      { const int rr = ensure_tiles_fwd(ctx); if (rr) return rr; }
      TileArgs ta;
      ta.tiles = ctx->d_tiles_fwd; ta.ntiles = ctx->n_tiles_fwd; ta.tiles_x = ctx->fwd_tiles_x; ta.tiles_y = ctx->fwd_tiles_y;
      ta.cap = std::min(ctx->fwd_unit_cap, ctx->tile_cap); ta.dbg = ctx->dbg_back; ta.tw = ctx->fwd_tw; ta.th = ctx->fwd_th;
      ta.gauss = 0;
      const size_t lds = (size_t)ta.cap * 2 * sizeof(float);
      launch_fwd_unit<false>(ctx, a, ta, lds);
      KCHK("fwd_unit_kernel");
    } else if (a.n && ctx->pvr) {
      { const int rr = pixel_list_in_pieces(a, [&](const PsfArgs &ap, uint32_t) {
          hipLaunchKernelGGL(pvr_kernel<MODE_FWD>, dim3(nblk(ap.n, WAVES_PER_BLOCK)), dim3(WAVES_PER_BLOCK * 64), 0, ctx->stream, ap);
          KCHK("pvr_kernel<FWD>");
          return (int)SVR_OK; });
        if (rr) return rr; }
    } else if (a.n) {
      { const int rr = pixel_list_in_pieces(a, [&](const PsfArgs &ap, uint32_t) {
          hipLaunchKernelGGL(psf_kernel<MODE_FWD>, dim3(nblk(ap.n, WAVES_PER_BLOCK)), dim3(WAVES_PER_BLOCK * 64), (size_t)ctx->dbg_fwd_lds, ctx->stream, ap);
          KCHK("psf_kernel<FWD>");
          return (int)SVR_OK; });
        if (rr) return rr; }
    }
    return SVR_OK;
  };
  const bool tiled = a.n && (ctx->pvr ? ctx->pvr_mode == 1 : ctx->fwd_mode >= 1);
  if (tiled && !cells && !ctx->fwd_autotune && !ctx->fwd_tile_user) {
    int w, h;
    tile_shape_rule(ctx, a.coeff != nullptr, w, h);
    if (w != ctx->fwd_tw || h != ctx->fwd_th) {
      ctx->fwd_tw = w; ctx->fwd_th = h; ctx->psf_list_valid = false;
      r = ensure_psf_list(ctx);
      if (r) return r;
      a.list = ctx->d_psf_list; a.n = ctx->n_psf;
    }
  }
  if (tiled && !cells && ctx->fwd_tune_pending && !ctx->fwd_tile_user) {
    ctx->fwd_tune_pending = false;
    static const int cand[6][2] = {{4, 4}, {6, 4}, {6, 5}, {8, 4}, {4, 2}, {2, 2}};   // the first is the default; smaller boxes for finer volumes (at most FWDU_MAXPIX = 32 pixels)
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    float best = 3.0e38f;
    int pick = 0;
    for (int c = 0; c < 6; ++c) {
      ctx->fwd_tw = cand[c][0]; ctx->fwd_th = cand[c][1]; ctx->psf_list_valid = false;
      r = ensure_psf_list(ctx);
      if (r) return r;
      a.list = ctx->d_psf_list; a.n = ctx->n_psf;
      TileSample sample;
      if (!r) r = ensure_tiles_fwd(ctx);
      if (r) return r;
      r = sample.begin(ctx, ctx->d_tiles_fwd, ctx->n_tiles_fwd);
      float ms = 0.0f;
      for (int rep = 0; rep < 2 && !r; ++rep) {          // the second run is the one that counts (a new tile list costs its first launch)
        hipError_t he = hipEventRecord(e0, ctx->stream);
        r = launch_forward();
        if (he == hipSuccess) he = hipEventRecord(e1, ctx->stream);
        if (he == hipSuccess) he = hipEventSynchronize(e1);
        if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
        if (!r && he != hipSuccess) r = fail(ctx, (int)he, std::string("tile tuning: ") + hipGetErrorString(he));
      }
      sample.end();                                      // before anything can rebuild or free the list
      if (r) return r;
      ms *= sample.scale;
      if (getenv("SVR_TUNE_DEBUG")) fprintf(stderr, "[tune] gather %dx%d: %.3f ms (whole launch; timed 1/%.1f of the tiles)\n", cand[c][0], cand[c][1], ms, sample.scale);
      if (ms < best) { best = ms; pick = c; }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    ctx->fwd_tw = cand[pick][0]; ctx->fwd_th = cand[pick][1]; ctx->psf_list_valid = false;
    r = ensure_psf_list(ctx);
    if (r) return r;
    a.list = ctx->d_psf_list; a.n = ctx->n_psf;
  }
  ScopedTimer t(ctx, SVR_T_FORWARD);
  if (store) t.also(SVR_T_FORWARD_STORE); else if (a.coeff) t.also(SVR_T_FORWARD_TABLE);
  r = launch_forward();
  if (r) return r;
  t.stop();
  if (!cells) {                                            // (the cell gather's finish raises the slices' flags itself)
    hipLaunchKernelGGL(k_slice_inside, dim3(ctx->ns), dim3(256), 0, ctx->stream, ctx->d_siminside,
                       (int)(ctx->sx * ctx->sy), ctx->d_slice_inside);
    KCHK("k_slice_inside");
  }
  // slice_inside == NULL: the flags stay on the device (svr_get_slice_inside, svr_mstep_estep) and the call does not wait
  if (slice_inside) {
    if ((r = down_queue(ctx, slice_inside, ctx->d_slice_inside, ctx->ns))) return r;
    return down_flush(ctx);
  }
  return SVR_OK;
}

int svr_get_slice_inside(svr_ctx *ctx, uint8_t *slice_inside) {
  SVR_ENTER(ctx);
  if (!ctx || !slice_inside) return SVR_E_ARG;
  NEED(ctx->have_slices, "slices not filled");
  int r = down_queue(ctx, slice_inside, ctx->d_slice_inside, ctx->ns);
  if (r) return r;
  return down_flush(ctx);
}

// ---- EM --------------------------------------------------------------------------------
int svr_initialize_em_values(svr_ctx *ctx) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  NEED(ctx->have_slices, "slices not filled");
  hipLaunchKernelGGL(k_init_em, dim3(nblk(ctx->np)), dim3(256), 0, ctx->stream, ctx->d_slices, ctx->d_weights, ctx->np, ctx->pvr);
  KCHK("k_init_em");
  if (!ctx->disable_bias) {                              // RC.cu:3305-3309
    int r = ensure_bias_buffers(ctx);
    if (r) return r;
    HIPCHK(hipMemsetAsync(ctx->d_bias, 0, ctx->np * sizeof(float), ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}

int svr_robust_statistics_sums(svr_ctx *ctx, double out2[2]) {
  SVR_ENTER(ctx);
  if (!ctx || !out2) return SVR_E_ARG;
  NEED(ctx->have_slices, "slices not filled");
  hipLaunchKernelGGL(k_robust, dim3(ctx->chunks, ctx->ns), dim3(256), 0, ctx->stream, ctx->d_slices, ctx->d_siminside,
                     ctx->d_simslices, ctx->d_simweights, (int)(ctx->sx * ctx->sy), ctx->d_partial);
  KCHK("k_robust");
  int r = reduce_partials(ctx, 2, 0, 0, true);
  if (r) return r;
  HIPCHK(hipMemcpyAsync(out2, ctx->d_out, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}

int svr_initialize_robust_statistics(svr_ctx *ctx, float *sigma) {
  SVR_ENTER(ctx);
  if (!ctx || !sigma) return SVR_E_ARG;
  double s2[2];
  int r = svr_robust_statistics_sums(ctx, s2);
  if (r) return r;
  *sigma = (float)s2[0] / (float)s2[1];   // RC.cu:2301-2305
  return SVR_OK;
}

namespace {
// the E-step's kernels; em = NULL: (m, sigma, mix) by value, else {sigma, mix, m} read from the device (svr_mstep_estep)
int launch_estep(svr_ctx *ctx, float m, float sigma, float mix, const float *em) {
  ScopedTimer t(ctx, SVR_T_ESTEP);
  hipLaunchKernelGGL(k_estep, dim3(ctx->chunks, ctx->ns), dim3(256), 0, ctx->stream, ctx->d_slices, ctx->d_simslices,
                     ctx->d_simweights, ctx->d_scales, ctx->disable_bias ? (const float *)nullptr : ctx->d_bias, m, sigma,
                     mix, (int)(ctx->sx * ctx->sy), ctx->d_weights, ctx->d_partial, ctx->pvr, em);
  KCHK("k_estep");
  hipLaunchKernelGGL(k_potential_finish, dim3(nblk(ctx->ns, 64)), dim3(64), 0, ctx->stream, ctx->d_partial,
                     (int)ctx->ns, ctx->chunks, ctx->d_tmp_ns);
  KCHK("k_potential_finish");
  t.stop();
  return SVR_OK;
}
// the M-step's sums -> d_out[5]
int launch_mstep(svr_ctx *ctx) {
  // the reference fills its per-pixel scale buffer from the HOST copy h_scales (RC.cu:3091-3094): d_scales_host_copy
  ScopedTimer t(ctx, SVR_T_MSTEP);
  hipLaunchKernelGGL(k_mstep, dim3(ctx->chunks, ctx->ns), dim3(256), 0, ctx->stream, ctx->d_slices, ctx->d_weights,
                     ctx->d_simslices, ctx->d_simweights, ctx->d_scales_host_copy,
                     ctx->disable_bias ? (const float *)nullptr : ctx->d_bias, (int)(ctx->sx * ctx->sy), ctx->d_partial);
  KCHK("k_mstep");
  int r = reduce_partials(ctx, 5, 1 << 3, 1 << 4, true);
  if (r) return r;
  t.stop();
  return SVR_OK;
}
// Reconstruction::MStep's host part RC.cu:3016-3071 (patch-based: patchBasedRobustStatistics_gpu.cu:570-640, no FLT_MAX /
// FLT_MIN clamps), the same float operations on either side
__host__ __device__ inline void mstep_scalars(const double s5[5], int iter, float step, int pvr, float &sigma_io, float &mix_io, float &m_out) {
  const float sigma = (float)s5[0], mix = (float)s5[1], num = (float)s5[2];
  float min_ = (float)s5[3], max_ = (float)s5[4];
  if (!pvr) {
    min_ = min_ < 3.402823466e+38f ? min_ : 3.402823466e+38f;       // std::min(FLT_MAX, .), std::max(FLT_MIN, .)
    max_ = 1.175494351e-38f < max_ ? max_ : 1.175494351e-38f;
  }
  if (mix > 0) sigma_io = sigma / mix;
  if (sigma_io < step * step / 6.28f) sigma_io = step * step / 6.28f;
  if (iter > 1) mix_io = mix / num;
  m_out = 1.0f / (max_ - min_);
}
__global__ void k_mstep_scalars(const double *s5, int iter, float step, int pvr, float sigma, float mix, float *em) {
  float m = 0.0f;
  mstep_scalars(s5, iter, step, pvr, sigma, mix, m);
  em[0] = sigma; em[1] = mix; em[2] = m;
}
// ... of a sharded run: all[world][8] = every rank's five sums (an all-gather on the device), added up in rank order exactly like
// the hosts do after their exchange (svr::irtkReconstruction::mstep_exchange) -- the same bits on every rank
__global__ void k_mstep_scalars_ranks(const double *all, int world, int iter, float step, int pvr, float sigma, float mix, float *em) {
  double s5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int r = 0; r < world; ++r) {
    for (int k = 0; k < 3; ++k) s5[k] += all[8 * r + k];
    s5[3] = r ? fmin(s5[3], all[8 * r + 3]) : all[3];
    s5[4] = r ? fmax(s5[4], all[8 * r + 4]) : all[4];
  }
  float m = 0.0f;
  mstep_scalars(s5, iter, step, pvr, sigma, mix, m);
  em[0] = sigma; em[1] = mix; em[2] = m;
}
}  // namespace

int svr_estep(svr_ctx *ctx, float m, float sigma, float mix, float *slice_potential) {
  SVR_ENTER(ctx);
  if (!ctx || !slice_potential) return SVR_E_ARG;
  NEED(ctx->have_slices && ctx->have_scales, "slices / scales not set");
  int r = launch_estep(ctx, m, sigma, mix, nullptr);
  if (r) return r;
  if ((r = down_queue(ctx, slice_potential, ctx->d_tmp_ns, ctx->ns * sizeof(float)))) return r;
  return down_flush(ctx);
}

int svr_mstep_sums(svr_ctx *ctx, double out5[5]) {
  SVR_ENTER(ctx);
  if (!ctx || !out5) return SVR_E_ARG;
  NEED(ctx->have_slices && ctx->have_scales, "slices / scales not set");
  int r = launch_mstep(ctx);
  if (r) return r;
  if ((r = down_queue(ctx, out5, ctx->d_out, 5 * sizeof(double)))) return r;
  return down_flush(ctx);
}

// svr_mstep_sums plus what deferred calls left on the device (each may be NULL), in the same wait: for a sharded host, which needs
// the sums on the host for its exchange anyway
int svr_mstep_sums_fetch(svr_ctx *ctx, double out5[5], float *scale_vec, uint8_t *slice_inside) {
  SVR_ENTER(ctx);
  if (!ctx || !out5) return SVR_E_ARG;
  NEED(ctx->have_slices && ctx->have_scales, "slices / scales not set");
  int r = launch_mstep(ctx);
  if (r) return r;
  if ((r = down_queue(ctx, out5, ctx->d_out, 5 * sizeof(double)))) return r;
  if (scale_vec && (r = down_queue(ctx, scale_vec, ctx->d_scales_host_copy, ctx->ns * sizeof(float)))) return r;
  if (slice_inside && (r = down_queue(ctx, slice_inside, ctx->d_slice_inside, ctx->ns))) return r;
  if ((r = down_flush(ctx))) return r;
  if (scale_vec) ctx->mir_scales_copy.assign(scale_vec, scale_vec + ctx->ns);
  return SVR_OK;
}

int svr_mstep(svr_ctx *ctx, int iter, float step, float *sigma_io, float *mix_io, float *m_out) {
  SVR_ENTER(ctx);
  if (!ctx || !sigma_io || !mix_io || !m_out) return SVR_E_ARG;
  double s5[5];
  int r = svr_mstep_sums(ctx, s5);
  if (r) return r;
  if (!((float)s5[1] > 0)) fprintf(stderr, "Something went wrong: sigma= %f mix= %f\n", (float)s5[0], (float)s5[1]);
  mstep_scalars(s5, iter, step, 0, *sigma_io, *mix_io, *m_out);
  return SVR_OK;
}

// MStep followed by EStep (reconstruction.cc:1093-1108, irtkPatchBasedReconstruction.cpp:540-545) with ONE wait for the
// device: the M-step's scalars are worked out by a one-thread kernel with the host's float operations and read by the
// E-step's kernel from device memory.  em3 = {sigma, mix, m} in/out.  scale_vec / slice_inside (each may be NULL): what
// a deferred svr_calculate_scale_vector(ctx, NULL) / svr_simulate_slices(ctx, NULL) left on the device, in the same wait.
int svr_mstep_estep(svr_ctx *ctx, int iter, float step, float em3[3], float *slice_potential, float *scale_vec, uint8_t *slice_inside) {
  SVR_ENTER(ctx);
  if (!ctx || !em3 || !slice_potential) return SVR_E_ARG;
  NEED(ctx->have_slices && ctx->have_scales, "slices / scales not set");
  if (!ctx->d_em) HIPCHK(hipMalloc(&ctx->d_em, 4 * sizeof(float)));
  int r = launch_mstep(ctx);
  if (r) return r;
  hipLaunchKernelGGL(k_mstep_scalars, dim3(1), dim3(1), 0, ctx->stream, ctx->d_out, iter, step, ctx->pvr ? 1 : 0, em3[0], em3[1], ctx->d_em);
  KCHK("k_mstep_scalars");
  if ((r = launch_estep(ctx, 0.0f, 0.0f, 0.0f, ctx->d_em))) return r;
  if ((r = down_queue(ctx, slice_potential, ctx->d_tmp_ns, ctx->ns * sizeof(float)))) return r;
  if ((r = down_queue(ctx, em3, ctx->d_em, 3 * sizeof(float)))) return r;
  if (scale_vec && (r = down_queue(ctx, scale_vec, ctx->d_scales_host_copy, ctx->ns * sizeof(float)))) return r;
  if (slice_inside && (r = down_queue(ctx, slice_inside, ctx->d_slice_inside, ctx->ns))) return r;
  if ((r = down_flush(ctx))) return r;
  if (scale_vec) ctx->mir_scales_copy.assign(scale_vec, scale_vec + ctx->ns);
  return SVR_OK;
}

// The same for a SHARDED run without a wait in between: svr_mstep_partial leaves this rank's five sums in *send (8 doubles = 16
// floats, device memory) and names the buffer the launcher's all-gather fills (*recv: world x 8 doubles); svr_mstep_estep_ranks
// adds the ranks' sums up on the device in rank order, works the scalars out there, runs the E-step on them and fetches this rank's
// potentials, the scalars and the deferred vectors in ONE wait.  The host's exchange of the M-step's sums is gone.
int svr_mstep_partial(svr_ctx *ctx, int world, void **send, void **recv) {
  SVR_ENTER(ctx);
  if (!ctx || !send || !recv || world < 1) return SVR_E_ARG;
  NEED(ctx->have_slices && ctx->have_scales, "slices / scales not set");
  if ((size_t)world > ctx->em_all_cap) {
    free_dev(ctx->d_em_all);
    ctx->em_all_cap = 0;
    HIPCHK(hipMalloc(&ctx->d_em_all, (size_t)world * 8 * sizeof(double)));
    ctx->em_all_cap = world;
  }
  int r = launch_mstep(ctx);
  if (r) return r;
  *send = ctx->d_out;
  *recv = ctx->d_em_all;
  return SVR_OK;
}
int svr_mstep_estep_ranks(svr_ctx *ctx, int world, int iter, float step, float em3[3], float *slice_potential, float *scale_vec, uint8_t *slice_inside) {
  SVR_ENTER(ctx);
  if (!ctx || !em3 || !slice_potential || world < 1) return SVR_E_ARG;
  NEED(ctx->have_slices && ctx->have_scales && ctx->d_em_all && (size_t)world <= ctx->em_all_cap, "svr_mstep_partial first");
  if (!ctx->d_em) HIPCHK(hipMalloc(&ctx->d_em, 4 * sizeof(float)));
  hipLaunchKernelGGL(k_mstep_scalars_ranks, dim3(1), dim3(1), 0, ctx->stream, ctx->d_em_all, world, iter, step, ctx->pvr ? 1 : 0, em3[0], em3[1], ctx->d_em);
  KCHK("k_mstep_scalars_ranks");
  int r;
  if ((r = launch_estep(ctx, 0.0f, 0.0f, 0.0f, ctx->d_em))) return r;
  if ((r = down_queue(ctx, slice_potential, ctx->d_tmp_ns, ctx->ns * sizeof(float)))) return r;
  if ((r = down_queue(ctx, em3, ctx->d_em, 3 * sizeof(float)))) return r;
  if (scale_vec && (r = down_queue(ctx, scale_vec, ctx->d_scales_host_copy, ctx->ns * sizeof(float)))) return r;
  if (slice_inside && (r = down_queue(ctx, slice_inside, ctx->d_slice_inside, ctx->ns))) return r;
  if ((r = down_flush(ctx))) return r;
  if (scale_vec) ctx->mir_scales_copy.assign(scale_vec, scale_vec + ctx->ns);
  return SVR_OK;
}

int svr_calculate_scale_vector(svr_ctx *ctx, float *scale_vec) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  NEED(ctx->have_slices && ctx->have_scales, "slices / scales not set");
  ScopedTimer t(ctx, SVR_T_SCALE);
  hipLaunchKernelGGL(k_scale, dim3(ctx->chunks, ctx->ns), dim3(256), 0, ctx->stream, ctx->d_slices, ctx->d_weights,
                     ctx->d_simslices, ctx->d_simweights, ctx->disable_bias ? (const float *)nullptr : ctx->d_bias,
                     (int)(ctx->sx * ctx->sy), ctx->d_partial);
  KCHK("k_scale");
  int r;
  hipLaunchKernelGGL(k_scale_finish, dim3(nblk(ctx->ns, 64)), dim3(64), 0, ctx->stream, ctx->d_partial, (int)ctx->ns, ctx->chunks,
                     ctx->d_tmp_ns, ctx->d_scales, ctx->d_scales_host_copy);
  KCHK("k_scale_finish");
  t.stop();
  // Reference quirk, reproduced: CalculateScaleVectorOnX uploads h_scales -- still the PREVIOUS
  // scale vector -- to the device (RC.cu:3238) and only afterwards h_scales = scale_vec
  // (RC.cu:3195).  The E-step / back-projection kernels therefore see scales that lag one call
  // behind; the M-step reads h_scales (RC.cu:3093) and sees the new ones.  h_scales is d_scales_host_copy here: both
  // moves are device-to-device, and with scale_vec == NULL (svr_get_scale_vector / svr_mstep_estep fetch it later) the
  // call does not wait for the device.  (Both moves happen in k_scale_finish.)
  ctx->mir_scales = ctx->mir_scales_copy;          // (empty = unknown: the next upload is not skipped)
  ctx->mir_scales_copy.clear();
  if (scale_vec) {
    if ((r = down_queue(ctx, scale_vec, ctx->d_scales_host_copy, ctx->ns * sizeof(float)))) return r;
    if ((r = down_flush(ctx))) return r;
    ctx->mir_scales_copy.assign(scale_vec, scale_vec + ctx->ns);
  }
  return SVR_OK;
}

int svr_get_scale_vector(svr_ctx *ctx, float *scale_vec) {
  SVR_ENTER(ctx);
  if (!ctx || !scale_vec) return SVR_E_ARG;
  NEED(ctx->have_slices && ctx->have_scales, "slices / scales not set");
  int r = down_queue(ctx, scale_vec, ctx->d_scales_host_copy, ctx->ns * sizeof(float));
  if (r) return r;
  if ((r = down_flush(ctx))) return r;
  ctx->mir_scales_copy.assign(scale_vec, scale_vec + ctx->ns);
  return SVR_OK;
}

// the scale vector last calculated becomes the one the kernels see (the patch-based loop's copyToScales: no lag,
// patchBasedRobustStatistics_gpu.cu:672-745) -- svr_update_scale_vector(scale_vec, unchanged weights) without the host
int svr_adopt_scale_vector(svr_ctx *ctx) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  NEED(ctx->have_slices && ctx->have_scales, "slices / scales not set");
  HIPCHK(hipMemcpyAsync(ctx->d_scales, ctx->d_scales_host_copy, ctx->ns * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  ctx->mir_scales = ctx->mir_scales_copy;
  return SVR_OK;
}

// ---- super-resolution ------------------------------------------------------------------
int svr_superresolution_backproject(svr_ctx *ctx, const float *slice_weight) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  int r = ready(ctx);
  if (r) return r;
  if (slice_weight) {
    r = svr_update_slice_weights(ctx, slice_weight);   // RC.cu:2123
    if (r) return r;
  }
  if (!ctx->fwd_autotune && !ctx->tile_user && !ctx->in_tune && back_mode_eff(ctx) != 5 && (ctx->pvr ? ctx->pvr_mode == 1 : back_mode_eff(ctx) >= 3) &&
      (ctx->tile_w != 6 || ctx->tile_h != 4)) {
    // the tiled scatters (back_mode 3 / 4: fallbacks of the cell scatter, and what a caller names) without trials: 6 x 4 pixels with
    // the 2096-voxel box -- the trials' pick or within 2 % of it on P4 (4.97 ms against 4.96-5.08) and S8 (6 x 5 / 2096: 50.5 against 52.2)
    ctx->in_tune = true;
    r = svr_set_option(ctx, "tile_w", 6);
    if (!r) r = svr_set_option(ctx, "tile_h", 4);
    ctx->tile_user = false;
    ctx->in_tune = false;
    if (r) return r;
  }
  if (ctx->back_tune_pending && !ctx->tile_user && !ctx->in_tune && back_mode_eff(ctx) != 5 && (ctx->pvr ? ctx->pvr_mode == 1 : back_mode_eff(ctx) >= 3)) {   // (mode 5: cells, no tile shape to time)
    ctx->back_tune_pending = false;
    ctx->in_tune = true;
    // with the coefficient table the pass waits for memory, not for the ALUs: larger tiles (fewer flushed voxels per
    // pixel) are tried first, with the largest box, before the box sizes are timed for the shape that won
    static const int cand_eval[6][2] = {{6, 5}, {6, 4}, {5, 4}, {4, 4}, {4, 2}, {2, 2}};   // the first four always, then smaller ones while they win
    static const int cand_tab[5][2] = {{8, 4}, {6, 4}, {4, 4}, {4, 2}, {2, 2}};
    const bool tab = ctx->coeff_mode && (ctx->pvr ? ctx->pvr_mode == 1 : back_mode_eff(ctx) == 4);
    const int (*cand)[2] = tab ? cand_tab : cand_eval;
    const int ncand = tab ? 5 : 6;
    const int cap0 = ctx->wave_cap;
    const auto timing = ctx->timers;
    ctx->timers = false;                                 // the trial runs stay out of the kernel timers
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    // one trial: the whole-launch time of the current shape with box `cap`
    bool warm = false;
    auto trial = [&](int cap, TileSample &sample, float &ms) -> int {
      ctx->wave_cap = cap;
      int rr = SVR_OK;
      ms = 0.0f;
      for (int rep = warm ? 1 : 0; rep < 2 && !rr; ++rep) {   // one timed run per box, after one warm-up run per shape (a new tile list
        warm = true;                                          // costs its first launch 0.1 ms on P4, enough to turn a near-tie)
        hipError_t he = hipEventRecord(e0, ctx->stream);
        rr = svr_superresolution_backproject(ctx, nullptr);
        if (he == hipSuccess) he = hipEventRecord(e1, ctx->stream);
        if (he == hipSuccess) he = hipEventSynchronize(e1);
        if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
        if (!rr && he != hipSuccess) rr = fail(ctx, (int)he, std::string("tile tuning: ") + hipGetErrorString(he));
      }
      ms *= sample.scale;
      return rr;
    };
    // the wave-owned scatter's LDS request decides how many wavefronts a CU holds; the smallest box that still takes
    // (nearly) every tile wins -- tiles that do not fit are re-run by the workgroup kernel, so any value is correct.
    // LDS is handed out in 1280-byte granules (measured: the time steps between 1764 and 1850 box voxels, not where
    // 160 KiB / request changes), so a CU holds floor(128 / granules) wavefronts: the candidates are the largest boxes
    // (with the kernel's 1152 static bytes) that still give 8, 9, 10, 11 and 12 of them.  A larger tile flushes fewer
    // voxels per pixel but needs the larger box, so shape and box are timed together: per shape the boxes from the
    // largest down while they get faster (6x4 with 2096 beats 6x5, which wins at 2416 and then cannot shrink).
    static const int caps[5] = {2416, 2096, 1776, 1616, 1456};
    const bool wave = (ctx->pvr || back_mode_eff(ctx) == 4) && !ctx->wave_cap_user;
    float best = 3.0e38f;
    int pick = 0, pick_cap = wave ? caps[0] : cap0;
    for (int c = 0; c < ncand && !r; ++c) {
      r = svr_set_option(ctx, "tile_w", cand[c][0]);
      if (!r) r = svr_set_option(ctx, "tile_h", cand[c][1]);
      if (!r) r = ensure_psf_list(ctx);                  // the list of this shape, so that the trials below find it valid
      TileSample sample;
      if (!r) r = ensure_tiles_back(ctx);
      if (!r) r = sample.begin(ctx, ctx->d_tiles, ctx->n_tiles);
      float shape_best = 3.0e38f;
      warm = false;
      for (int k = 0; k < (wave ? 5 : 1) && !r; ++k) {
        float ms;
        r = trial(wave ? caps[k] : cap0, sample, ms);
        if (r) break;
        if (getenv("SVR_TUNE_DEBUG"))
          fprintf(stderr, "[tune] scatter %dx%d box %d: %.3f ms (whole launch; timed 1/%.1f of the tiles)\n", cand[c][0], cand[c][1], ctx->wave_cap, ms, sample.scale);
        if (ms < best) { best = ms; pick = c; pick_cap = ctx->wave_cap; }
        if (ms < shape_best) shape_best = ms;
        else break;                                      // a smaller box stopped paying for this shape
        if (ms > 1.3f * best) break;                     // ... or the shape is out of reach
      }
      sample.end();
      if (r) break;
      if (pick != c && c >= (tab ? 2 : 3)) break;        // the small shapes only while they win
    }
    ctx->wave_cap = pick_cap;
    if (!r) r = svr_set_option(ctx, "tile_w", cand[pick][0]);
    if (!r) r = svr_set_option(ctx, "tile_h", cand[pick][1]);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    ctx->timers = timing;
    ctx->in_tune = false;
    if (r) return r;
  }
  r = ensure_psf_list(ctx);
  if (r) return r;
  // (coeff_lazy: a scatter that finds no table evaluates -- and, on the cell path, writes it: whichever PSF pass comes first after a new geometry does)
  r = ensure_coeff_unless_lazy(ctx);
  if (r) return r;
  bool store = false;
  // RC.cu:2202-2203 -- not needed where the cell scatter runs: its combine writes EVERY voxel of addon | cmap (0 outside the mask); decided below
  bool need_clear = true;
  ctx->prep_pending = false;
  ctx->cmap_from_scatter = true;
  PsfArgs a = make_args(ctx);
  a.list = ctx->d_psf_list;
  a.n = ctx->n_psf;
  if (ctx->coeff_mode && ctx->coeff_valid && (ctx->pvr ? ctx->pvr_mode == 1 : back_mode_eff(ctx) >= 4)) give_coeff(ctx, a);
  const bool tiled = ctx->pvr ? ctx->pvr_mode == 1 : back_mode_eff(ctx) >= 1;
  if (!a.coeff && a.n && tiled && back_mode_eff(ctx) == 5 && !ctx->pvr) {
    if ((r = cell_prepare(ctx))) return r;
    if (ctx->cell->usable) {
      if ((r = coeff_begin_store(ctx, &store))) return r;
      if (store) give_coeff(ctx, a);
    }
  }
  ScopedTimer t(ctx, SVR_T_BACKPROJECT);
  if (store) t.also(SVR_T_BACKPROJECT_STORE); else if (a.coeff) t.also(SVR_T_BACKPROJECT_TABLE);
  if (!(a.n && tiled)) { need_clear = false; HIPCHK(hipMemsetAsync(ctx->d_addon_cmap, 0, 2 * ctx->nv * sizeof(float), ctx->stream)); }
  if (a.n && tiled) {
    TileArgs ta;
    ta.tiles_x = ctx->tiles_x; ta.tiles_y = ctx->tiles_y; ta.dbg = ctx->dbg_back;
    ta.tw = ctx->tile_w; ta.th = ctx->tile_h; ta.gauss = 0;
    bool cells = false;
    if (back_mode_eff(ctx) == 5) {
      if ((r = cell_prepare(ctx))) return r;
      cells = ctx->cell->usable;
      if (!cells) ctx->note_fallback(0, "the scatter left the cell path (the cell lists cannot hold this geometry): back_mode 4, float atomics, last bits depend on the run");
    }
    need_clear = !cells;
    if (need_clear) { need_clear = false; HIPCHK(hipMemsetAsync(ctx->d_addon_cmap, 0, 2 * ctx->nv * sizeof(float), ctx->stream)); }
    if (cells) {
      r = launch_cell_scatter(ctx, a, 0, ctx->addon(), ctx->cmap(), store);
      if (!r && store) { ctx->coeff_valid = true; ctx->coeff_full = false; }   // (the PSF pixels' live units: what the SR iterations' passes read)
    }
    else if (!(r = ensure_tiles_back(ctx))) r = launch_scatter(ctx, ctx->pvr ? 4 : std::min(4, back_mode_eff(ctx)), a, ta, ctx->d_tiles, ctx->n_tiles, MODE_BACK);
    if (r) return r;
  } else if (a.n && ctx->pvr) {
    { const int rr = pixel_list_in_pieces(a, [&](const PsfArgs &ap, uint32_t) {
        hipLaunchKernelGGL(pvr_kernel<MODE_BACK>, dim3(nblk(ap.n, WAVES_PER_BLOCK)), dim3(WAVES_PER_BLOCK * 64), 0, ctx->stream, ap);
        KCHK("pvr_kernel<BACK>");
        return (int)SVR_OK; });
      if (rr) return rr; }
  } else if (a.n) {
    { const int rr = pixel_list_in_pieces(a, [&](const PsfArgs &ap, uint32_t) {
        hipLaunchKernelGGL(psf_kernel<MODE_BACK>, dim3(nblk(ap.n, WAVES_PER_BLOCK)), dim3(WAVES_PER_BLOCK * 64), 0, ctx->stream, ap);
        KCHK("psf_kernel<BACK>");
        return (int)SVR_OK; });
      if (rr) return rr; }
  }
  t.stop();
  if (!ctx->sr_no_wait) HIPCHK(hipStreamSynchronize(ctx->stream));   // (svr_superresolution: the update follows on the same stream)
  return SVR_OK;
}

// (double)r < (double)min * 0.9 and (double)r > (double)max * 1.1 (RC.cu:1964-1967) for a float r, as float comparisons
static void regul_thresholds(float min_i, float max_i, RegulArgs &ra) {
  const double dlo = (double)min_i * 0.9, dhi = (double)max_i * 1.1;
  ra.val_lo = (float)dlo;
  ra.val_hi = (float)dhi;
  ra.thr_lo = (double)ra.val_lo >= dlo ? ra.val_lo : nextafterf(ra.val_lo, INFINITY);
  ra.thr_hi = (double)ra.val_hi <= dhi ? ra.val_hi : nextafterf(ra.val_hi, -INFINITY);
}

// Prep's own changes of addon / cmap, left out by k_regul_fused, for whoever reads the two buffers
static int regul_settle(svr_ctx *ctx) {
  if (!ctx->prep_pending) return SVR_OK;
  ctx->prep_pending = false;
  hipLaunchKernelGGL(k_regul_settle, dim3(nblk(ctx->nv)), dim3(256), 0, ctx->stream, ctx->addon(), ctx->cmap(), ctx->nv);
  KCHK("k_regul_settle");
  return SVR_OK;
}

static int superresolution_update_planes(svr_ctx *ctx, int adaptive, float alpha, float min_intensity, float max_intensity, float delta,
                                         float lambda, int z_lo, int z_hi) {
  if (ctx->reg_mode == 0) {
    if (!ctx->d_snap) HIPCHK(hipMalloc(&ctx->d_snap, ctx->nv * sizeof(float)));
    if (z_lo != 0 || z_hi != (int)ctx->vz) return fail(ctx, SVR_E_ARG, "reg_mode 0 updates whole volumes only");
    float *out = ctx->recon_cur == ctx->d_recon_volw ? ctx->d_recon_new : ctx->d_recon_volw;
    hipLaunchKernelGGL(k_reg_prep, dim3(nblk(ctx->nv)), dim3(256), 0, ctx->stream, adaptive, alpha, ctx->recon(),
                       ctx->addon(), ctx->cmap(), min_intensity, max_intensity, ctx->d_snap, ctx->nv);
    KCHK("k_reg_prep");
    hipLaunchKernelGGL(k_regularize, dim3((ctx->vx + 63) / 64, (ctx->vy + 3) / 4, ctx->vz), dim3(256), 0, ctx->stream,
                       (int)ctx->vx, (int)ctx->vy, (int)ctx->vz, delta, alpha, lambda, ctx->d_snap, ctx->recon(),
                       ctx->cmap(), out);
    KCHK("k_regularize");
    return SVR_OK;
  }
  RegulArgs ra;
  ra.recon = ctx->recon(); ra.addon = ctx->addon(); ra.cmap = ctx->cmap();
  ra.out = ctx->recon_cur == ctx->d_recon_volw ? ctx->d_recon_new : ctx->d_recon_volw;
  ra.vx = (int)ctx->vx; ra.vy = (int)ctx->vy; ra.vz = (int)ctx->vz;
  ra.z_lo = z_lo; ra.z_hi = z_hi;
  // tile of a workgroup (option reg_tile).  Measured on one MI355X (tools/exp_regul_tile.py; 64 x 8 / 32 x 16 / 32 x 8): P4 19.4 / 20.2 /
  // 18.2 us, S8 (273^3) 189 / 177 / 164 us, PVR8spx (400 x 400 x 320) 429 / 394 / 395 us: the small tile wastes the fewest lanes
  // beyond the volume's extent and keeps more workgroups per CU between its barriers
  static const int shapes[3][2] = {{64, 8}, {32, 16}, {32, 8}};
  const int pick = ctx->reg_tile < 0 ? 2 : ctx->reg_tile;
  const int TW = shapes[pick][0], TH = shapes[pick][1];
  const int tiles = (int)((ctx->vx + TW - 1) / TW * ((ctx->vy + TH - 1) / TH));
  const int nz = z_hi - z_lo;
  const int want = std::max(1, (2048 * 512 / (TW * TH) + tiles - 1) / tiles);   // chunks along z for >= 2048 workgroups of 512 lanes
  ra.zc = std::min(32, std::max(4, (nz + want - 1) / want));
  const int chunks = (nz + ra.zc - 1) / ra.zc;
  for (int k = 0; k < 3; ++k) { ra.blo[k] = 0; ra.bhi[k] = -1; }
  if (ctx->mbox_valid && ctx->cmap_from_scatter) {
    const int dims[3] = {ra.vx, ra.vy, ra.vz};
    for (int k = 0; k < 3; ++k) { ra.blo[k] = std::max(0, ctx->mbox_lo[k] - 1); ra.bhi[k] = std::min(dims[k] - 1, ctx->mbox_hi[k] + 1); }
  }
  ra.alpha = alpha;
  regul_thresholds(min_intensity, max_intensity, ra);
  ra.k = alpha * lambda / (delta * delta);                                // RC.cu:2107 (float)
  const double d2 = (double)delta * (double)delta;
  ra.kap1 = (float)(1.0 / d2); ra.kap2 = (float)(0.5 / d2); ra.kap3 = (float)((double)(1.0f / 3.0f) / d2);
  if (nz > 0) {
    const dim3 grid((ctx->vx + TW - 1) / TW, (ctx->vy + TH - 1) / TH, chunks);
    const dim3 block(TW * TH);
#define REGUL_LAUNCH(AD, W, H) hipLaunchKernelGGL((k_regul_fused<AD, W, H>), grid, block, 0, ctx->stream, ra)
    if (adaptive) { if (pick == 0) REGUL_LAUNCH(true, 64, 8); else if (pick == 1) REGUL_LAUNCH(true, 32, 16); else REGUL_LAUNCH(true, 32, 8); }
    else { if (pick == 0) REGUL_LAUNCH(false, 64, 8); else if (pick == 1) REGUL_LAUNCH(false, 32, 16); else REGUL_LAUNCH(false, 32, 8); }
#undef REGUL_LAUNCH
    KCHK("k_regul_fused");
  }
  if (!adaptive) ctx->prep_pending = true;
  return SVR_OK;
}

int svr_superresolution_update(svr_ctx *ctx, int adaptive, float alpha, float min_intensity,
                               float max_intensity, float delta, float lambda) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  NEED(ctx->nv > 0, "volume not set");
  if (alpha * lambda / (delta * delta) > 0.068)   // RC.cu:2124-2127
    fprintf(stderr, "Warning: regularization might not have smoothing effect! Ensure that alpha*lambda/delta^2 is below 0.068.");
  ScopedTimer t(ctx, SVR_T_REGULARIZE);
  int r = superresolution_update_planes(ctx, adaptive, alpha, min_intensity, max_intensity, delta, lambda, 0, (int)ctx->vz);
  if (r) return r;
  ctx->recon_cur = ctx->recon_cur == ctx->d_recon_volw ? ctx->d_recon_new : ctx->d_recon_volw;   // the update wrote the other buffer
  // a whole-volume update carries whatever the old volume held outside the dilated mask over into the buffer it wrote: the slab update
  // (svr_slab.inc), which only writes inside it, must clear that buffer before it uses it next
  ctx->vol_clean[ctx->recon_cur == ctx->d_recon_new ? 1 : 0] = false;
  t.stop();
  if (!ctx->sr_no_wait) HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}

// ---- the volume update of a sharded run: reduce-scatter -> the rank's z-slab -> all-gather (svr_slab.inc) ----------------
int svr_slab_plan(svr_ctx *ctx, int world, int rank, size_t *rs_floats_per_rank, size_t *ag_floats_per_rank) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  const int r = slab_plan(ctx, world, rank);
  if (r) return r;
  if (rs_floats_per_rank) *rs_floats_per_rank = (size_t)2 * ctx->slab->rs_chunk;
  if (ag_floats_per_rank) *ag_floats_per_rank = ctx->slab->ag_chunk;
  return SVR_OK;
}
int svr_slab_rs_pack(svr_ctx *ctx, void **send, void **recv) {
  SVR_ENTER(ctx);
  if (!ctx || !send || !recv) return SVR_E_ARG;
  NEED(ctx->slab && ctx->slab->valid && ctx->slab->world > 0, "svr_slab_plan first");
  NEED(ctx->cmap_from_scatter, "svr_slab_rs_pack: addon | cmap must come from svr_superresolution_backproject");
  SlabPlan &p = *ctx->slab;
  hipLaunchKernelGGL(k_slab_rs_pack, dim3(nblk((size_t)p.world * 2 * p.rs_chunk)), dim3(256), 0, ctx->stream, ctx->addon(), ctx->cmap(), p.d_midx, p.d_tab,
                     p.world, p.rs_chunk, p.d_send);
  KCHK("k_slab_rs_pack");
  *send = p.d_send; *recv = p.d_recv;
  return SVR_OK;
}
int svr_slab_update(svr_ctx *ctx, int adaptive, float alpha, float min_intensity, float max_intensity, float delta, float lambda,
                    void **send, void **recv) {
  SVR_ENTER(ctx);
  if (!ctx || !send || !recv) return SVR_E_ARG;
  NEED(ctx->slab && ctx->slab->valid && ctx->slab->world > 0, "svr_slab_plan first");
  if (ctx->reg_mode == 0) return fail(ctx, SVR_E_STATE, "svr_slab_update needs reg_mode 1");
  SlabPlan &p = *ctx->slab;
  const int R = p.rank;
  ScopedTimer t(ctx, SVR_T_REGULARIZE);
  if (p.rs_count[R]) {
    hipLaunchKernelGGL(k_slab_rs_unpack, dim3(nblk((size_t)2 * p.rs_count[R])), dim3(256), 0, ctx->stream, ctx->addon(), ctx->cmap(), p.d_midx, p.rs_start[R],
                       p.rs_count[R], p.rs_chunk, p.d_recv);
    KCHK("k_slab_rs_unpack");
  }
  const int ob = ctx->recon_cur == ctx->d_recon_volw ? 1 : 0;           // the buffer the update writes
  float *out = ob ? ctx->d_recon_new : ctx->d_recon_volw;
  if (!ctx->vol_clean[ob]) {
    HIPCHK(hipMemsetAsync(out, 0, ctx->nv * sizeof(float), ctx->stream));
    ctx->vol_clean[ob] = true;
  }
  int r = superresolution_update_planes(ctx, adaptive, alpha, min_intensity, max_intensity, delta, lambda, p.zb[R], p.zb[R + 1]);
  if (r) return r;
  hipLaunchKernelGGL(k_slab_ag_pack, dim3(nblk(p.ag_chunk)), dim3(256), 0, ctx->stream, out, p.d_didx, p.ag_start[R], p.ag_count[R], p.ag_chunk, p.d_send);
  KCHK("k_slab_ag_pack");
  t.stop();
  *send = p.d_send; *recv = p.d_recv;
  return SVR_OK;
}
int svr_slab_finish(svr_ctx *ctx) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  NEED(ctx->slab && ctx->slab->valid && ctx->slab->world > 0, "svr_slab_plan first");
  SlabPlan &p = *ctx->slab;
  float *out = ctx->recon_cur == ctx->d_recon_volw ? ctx->d_recon_new : ctx->d_recon_volw;
  hipLaunchKernelGGL(k_slab_ag_unpack, dim3(nblk((size_t)p.world * p.ag_chunk)), dim3(256), 0, ctx->stream, out, p.d_didx, p.d_tab, p.world, p.rank, p.ag_chunk, p.d_recv);
  KCHK("k_slab_ag_unpack");
  ctx->recon_cur = out;
  // addon | cmap now hold the sums of this rank's slab only: not a state a reader should see as "the scatter's result"
  ctx->cmap_from_scatter = false;
  ctx->prep_pending = false;
  if (!ctx->sr_no_wait) HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}

int svr_superresolution(svr_ctx *ctx, int iter, const float *slice_weight, int adaptive, float alpha,
                        float min_intensity, float max_intensity, float delta, float lambda,
                        int global_bias_correction, float sigma_bias, float low_intensity_cutoff) {
  SVR_ENTER(ctx);
  (void)iter; (void)sigma_bias; (void)low_intensity_cutoff;
  // back-projection and update without a wait for the device in between or after: whatever reads the volume next is ordered
  // behind them on the stream, and every call that hands results to the host waits itself (down_flush, svr_sync_cpu, debug_get)
  ctx->sr_no_wait = true;
  int r = svr_superresolution_backproject(ctx, slice_weight);
  if (!r) r = svr_superresolution_update(ctx, adaptive, alpha, min_intensity, max_intensity, delta, lambda);
  ctx->sr_no_wait = false;
  if (r) return r;
  if (global_bias_correction) printf("_global_bias_correction not implemented\n");   // RC.cu:2182-2185
  return SVR_OK;
}

int svr_mask_volume(svr_ctx *ctx) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  NEED(ctx->nv > 0 && ctx->have_mask, "volume / mask not set");
  hipLaunchKernelGGL(k_mask_volume, dim3(nblk(ctx->nv)), dim3(256), 0, ctx->stream, ctx->recon(), ctx->d_mask, ctx->nv);
  KCHK("k_mask_volume");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}

int svr_scale_volume_sums(svr_ctx *ctx, double out2[2]) {
  SVR_ENTER(ctx);
  if (!ctx || !out2) return SVR_E_ARG;
  NEED(ctx->have_slices && ctx->have_scales, "slices / scales not set");
  hipLaunchKernelGGL(k_scalevol, dim3(ctx->chunks, ctx->ns), dim3(256), 0, ctx->stream, ctx->d_slices, ctx->d_weights,
                     ctx->d_simslices, ctx->d_simweights, ctx->d_slice_weights, (int)(ctx->sx * ctx->sy),
                     ctx->d_partial);
  KCHK("k_scalevol");
  int r = reduce_partials(ctx, 2, 0, 0, true);
  if (r) return r;
  HIPCHK(hipMemcpyAsync(out2, ctx->d_out, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}

int svr_scale_volume_apply(svr_ctx *ctx, float scale) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  NEED(ctx->nv > 0, "volume not set");
  hipLaunchKernelGGL(k_scale_volume, dim3(nblk(ctx->nv)), dim3(256), 0, ctx->stream, ctx->recon(), scale, ctx->nv);
  KCHK("k_scale_volume");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}

int svr_scale_volume(svr_ctx *ctx) {
  SVR_ENTER(ctx);
  double s2[2];
  int r = svr_scale_volume_sums(ctx, s2);
  if (r) return r;
  float scale = (float)(s2[0] / s2[1]);   // RC.cu:3459
  printf("Volume scale GPU: %f\n", scale);
  return svr_scale_volume_apply(ctx, scale);
}

int svr_restore_slice_intensities(svr_ctx *ctx, const float *stack_factors, int n_stacks,
                                  const int *stack_index) {
  SVR_ENTER(ctx);
  if (!ctx || !stack_factors || !stack_index || n_stacks <= 0) return SVR_E_ARG;
  NEED(ctx->have_slices, "slices not filled");
  for (uint32_t i = 0; i < ctx->ns; ++i)
    if (stack_index[i] < 0 || stack_index[i] >= n_stacks) return fail(ctx, SVR_E_ARG, "stack index out of range");
  float *d_f = nullptr;
  int *d_i = nullptr;
  HIPCHK(hipMalloc(&d_f, n_stacks * sizeof(float)));
  HIPCHK(hipMalloc(&d_i, ctx->ns * sizeof(int)));
  HIPCHK(hipMemcpyAsync(d_f, stack_factors, n_stacks * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(d_i, stack_index, ctx->ns * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_restore, dim3(nblk(ctx->np)), dim3(256), 0, ctx->stream, ctx->d_slices, d_f, d_i,
                     (int)(ctx->sx * ctx->sy), ctx->np);
  hipError_t e = hipGetLastError();
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_f);
  (void)hipFree(d_i);
  if (e != hipSuccess) return fail(ctx, (int)e, "k_restore");
  return SVR_OK;
}

// ---- buffers ---------------------------------------------------------------------------
static int buffer_info(svr_ctx *ctx, int which, void **ptr, size_t *bytes) {
  if ((which == SVR_BUF_ADDON || which == SVR_BUF_CONFIDENCE_MAP) && ctx->nv) {
    const int rs = regul_settle(ctx);      // (the fused volume update leaves Prep's changes of the two buffers to their readers)
    if (rs) return rs;
  }
  switch (which) {
    case SVR_BUF_RECONSTRUCTED: *ptr = ctx->nv ? ctx->recon() : nullptr; *bytes = ctx->nv * 4; break;
    case SVR_BUF_VOL_WEIGHTS: *ptr = ctx->nv ? ctx->volw() : nullptr; *bytes = ctx->nv * 4; break;
    case SVR_BUF_ADDON: *ptr = ctx->nv ? ctx->addon() : nullptr; *bytes = ctx->nv * 4; break;
    case SVR_BUF_CONFIDENCE_MAP: *ptr = ctx->nv ? ctx->cmap() : nullptr; *bytes = ctx->nv * 4; break;
    case SVR_BUF_MASK: *ptr = ctx->d_mask; *bytes = ctx->nv * 4; break;
    case SVR_BUF_BIAS_VOLUME: *ptr = ctx->d_bias_vol; *bytes = ctx->nv * 4; break;
    case SVR_BUF_SMOOTH_MASK: *ptr = ctx->d_maskC; *bytes = ctx->nv * 4; break;
    case SVR_BUF_BIAS: *ptr = ctx->d_bias; *bytes = ctx->np * 4; break;
    case SVR_BUF_SLICES: *ptr = ctx->d_slices; *bytes = ctx->np * 4; break;
    case SVR_BUF_WEIGHTS: *ptr = ctx->d_weights; *bytes = ctx->np * 4; break;
    case SVR_BUF_SIMSLICES: *ptr = ctx->d_simslices; *bytes = ctx->np * 4; break;
    case SVR_BUF_SIMWEIGHTS: *ptr = ctx->d_simweights; *bytes = ctx->np * 4; break;
    case SVR_BUF_PSF_SUMS: *ptr = ctx->d_psf_sums; *bytes = ctx->np * 4; break;
    case SVR_BUF_SIMINSIDE: *ptr = ctx->d_siminside; *bytes = ctx->np; break;
    case SVR_BUF_VOXEL_COUNT: *ptr = ctx->d_voxcount; *bytes = ctx->np * 4; break;
    default: return SVR_E_ARG;
  }
  return *ptr ? SVR_OK : SVR_E_STATE;
}

int svr_debug_get(svr_ctx *ctx, int which, void *host_out, size_t bytes) {
  SVR_ENTER(ctx);
  if (!ctx || !host_out) return SVR_E_ARG;
  void *p; size_t b;
  int r = buffer_info(ctx, which, &p, &b);
  if (r) return fail(ctx, r, "svr_debug_get: buffer not available");
  if (bytes != b) return fail(ctx, SVR_E_ARG, "svr_debug_get: size mismatch");
  HIPCHK(hipMemcpyAsync(host_out, p, b, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}

int svr_debug_set(svr_ctx *ctx, int which, const void *host_in, size_t bytes) {
  SVR_ENTER(ctx);
  if (!ctx || !host_in) return SVR_E_ARG;
  void *p; size_t b;
  int r = buffer_info(ctx, which, &p, &b);
  if (r) return fail(ctx, r, "svr_debug_set: buffer not available");
  if (bytes != b) return fail(ctx, SVR_E_ARG, "svr_debug_set: size mismatch");
  HIPCHK(hipMemcpyAsync(p, host_in, b, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (which == SVR_BUF_PSF_SUMS || which == SVR_BUF_SLICES) { ctx->psf_list_valid = false; if (!ctx->coeff_full) ctx->coeff_valid = false; cell_gf_invalidate(ctx); }
  if (which == SVR_BUF_SIMINSIDE) {                      // (the per-slice flags follow: the cell gather only ever raises them)
    hipLaunchKernelGGL(k_slice_inside, dim3(ctx->ns), dim3(256), 0, ctx->stream, ctx->d_siminside, (int)(ctx->sx * ctx->sy), ctx->d_slice_inside);
    KCHK("k_slice_inside");
  }
  if (which == SVR_BUF_SLICES) { ctx->coeff_valid = false; cell_invalidate(ctx); }   // the table and the cell lists cover the pixels with s != -1
  if (which == SVR_BUF_MASK) ctx->mbox_valid = false;                               // (a mask set behind svr_set_mask's back: the whole pair is exchanged)
  if (which == SVR_BUF_MASK && ctx->slab) ctx->slab->valid = false;
  if (which == SVR_BUF_MASK) ctx->vol_clean[0] = ctx->vol_clean[1] = false;
  if (which == SVR_BUF_RECONSTRUCTED) ctx->vol_clean[ctx->recon_cur == ctx->d_recon_new ? 1 : 0] = false;
  if (which == SVR_BUF_ADDON || which == SVR_BUF_CONFIDENCE_MAP) ctx->cmap_from_scatter = false;   // (no longer known to vanish outside the mask)
  if (which == SVR_BUF_SLICES) return build_list(ctx, false);
  return SVR_OK;
}

void *svr_device_ptr(svr_ctx *ctx, int which) {
  SVR_ENTER(ctx);
  if (!ctx) return nullptr;
  void *p; size_t b;
  if (buffer_info(ctx, which, &p, &b)) return nullptr;
  return p;
}

size_t svr_volume_voxels(const svr_ctx *ctx) { return ctx ? ctx->nv : 0; }

int svr_debug_probe_pixel(svr_ctx *ctx, int slice, int px, int py, float *vals4096, int *centre3) {
  SVR_ENTER(ctx);
  if (!ctx || !vals4096 || !centre3) return SVR_E_ARG;
  int r = ready(ctx);
  if (r) return r;
  if (slice < 0 || slice >= (int)ctx->ns || px < 0 || px >= (int)ctx->sx || py < 0 || py >= (int)ctx->sy)
    return fail(ctx, SVR_E_ARG, "pixel out of range");
  float *d_v = nullptr;
  int *d_c = nullptr;
  HIPCHK(hipMalloc(&d_v, 4096 * sizeof(float)));
  HIPCHK(hipMalloc(&d_c, 3 * sizeof(int)));
  PsfArgs a = make_args(ctx);
  uint32_t idx = (uint32_t)px + (uint32_t)py * ctx->sx + (uint32_t)slice * ctx->sx * ctx->sy;
  hipLaunchKernelGGL(k_probe_pixel, dim3(1), dim3(64), 0, ctx->stream, a, idx, d_v, d_c);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(vals4096, d_v, 4096 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(centre3, d_c, 3 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_v);
  (void)hipFree(d_c);
  if (e != hipSuccess) return fail(ctx, (int)e, "svr_debug_probe_pixel");
  return SVR_OK;
}

// ---- PVR superpixel masks (ImagePatch2D::spxMask, R2/include/ImagePatch2D.cuh:51) ----------
int svr_set_spx_masks(svr_ctx *ctx, const char *masks_or_null) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  NEED(ctx->ns > 0, "initStorageVolumes first");
  free_dev(ctx->d_spx);
  if (!masks_or_null) return SVR_OK;
  const size_t bytes = (size_t)ctx->ns * 4096;
  HIPCHK(hipMalloc(&ctx->d_spx, bytes));
  HIPCHK(hipMemcpyAsync(ctx->d_spx, masks_or_null, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}

// ---- bias correction --------------------------------------------------------------------
int svr_correct_bias(svr_ctx *ctx, float sigma_bias, int global_bias_correction) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  NEED(!ctx->disable_bias, "bias correction is disabled (svr_set_flags)");
  NEED(ctx->have_slices && ctx->have_scales && ctx->have_dims, "slices / scales / slice dims not set");
  int r = ensure_bias_buffers(ctx);
  if (r) return r;
  r = prepare_slice_consts(ctx);
  if (r) return r;
  const size_t fb = ctx->np * sizeof(float);
  const int n2 = (int)(ctx->sx * ctx->sy);
  HIPCHK(hipMemsetAsync(ctx->d_wb, 0, fb, ctx->stream));       // RC.cu:1873-1875
  HIPCHK(hipMemsetAsync(ctx->d_wr, 0, fb, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_buffer, 0, fb, ctx->stream));
  hipLaunchKernelGGL(k_bias_residual, dim3(nblk(ctx->np)), dim3(256), 0, ctx->stream, ctx->d_slices, ctx->d_bias,
                     ctx->d_weights, ctx->d_simweights, ctx->d_simslices, ctx->d_scales, n2, ctx->np, ctx->d_wb, ctx->d_wr);
  KCHK("k_bias_residual");
  const dim3 grid((ctx->sx + 63) / 64, (ctx->sy + 3) / 4, ctx->ns);
#define CONV(in, out, horiz) \
  hipLaunchKernelGGL(k_gauss_conv_slices, grid, dim3(256), 0, ctx->stream, in, out, ctx->d_sc, (int)ctx->sx, (int)ctx->sy, \
                     (int)ctx->ns, sigma_bias, horiz)
  CONV(ctx->d_wb, ctx->d_buffer, 1);       // RC.cu:1886
  CONV(ctx->d_buffer, ctx->d_wb, 0);       // RC.cu:1888
  CONV(ctx->d_wr, ctx->d_buffer, 1);       // RC.cu:1889 (buffer still holds the first pass where the result is 0)
  CONV(ctx->d_buffer, ctx->d_wr, 0);       // RC.cu:1891
#undef CONV
  KCHK("k_gauss_conv_slices");
  hipLaunchKernelGGL(k_bias_update, dim3(nblk(ctx->np)), dim3(256), 0, ctx->stream, ctx->d_slices, ctx->d_bias, ctx->d_wb,
                     ctx->d_wr, ctx->np);
  KCHK("k_bias_update");
  if (!global_bias_correction) {
    hipLaunchKernelGGL(k_bias_mean, dim3(ctx->chunks, ctx->ns), dim3(256), 0, ctx->stream, ctx->d_slices, ctx->d_bias, n2,
                       ctx->d_partial);
    KCHK("k_bias_mean");
    r = reduce_partials(ctx, 2, 0, 0, false);
    if (r) return r;
    hipLaunchKernelGGL(k_bias_sub_mean, dim3(nblk(ctx->np)), dim3(256), 0, ctx->stream, ctx->d_slices, ctx->d_bias,
                       ctx->d_per_slice, n2, ctx->np);
    KCHK("k_bias_sub_mean");
  } else {
    printf("_global_bias_correction is not implemented in CUDA yet\n");   // RC.cu:1932
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}

int svr_normalise_bias_local(svr_ctx *ctx) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  NEED(!ctx->disable_bias, "bias correction is disabled (svr_set_flags)");
  int r = ready(ctx);
  if (r) return r;
  r = ensure_psf_list(ctx);
  if (r) return r;
  HIPCHK(hipMemsetAsync(ctx->d_bias_vol, 0, ctx->nv * sizeof(float), ctx->stream));   // RC.cu:2621
  PsfArgs a = make_args(ctx);
  a.list = ctx->d_psf_list;
  a.n = ctx->n_psf;
  a.recon = ctx->d_bias_vol;            // scattered value: psf/sume * (bias - log scale)
  a.volw = ctx->d_volume_weights;       // dev_volume_weights_: accumulates, never cleared (RC.cu:2633)
  if (a.n) {
    { const int rr = pixel_list_in_pieces(a, [&](const PsfArgs &ap, uint32_t) {
        hipLaunchKernelGGL(psf_kernel<MODE_BIAS>, dim3(nblk(ap.n, WAVES_PER_BLOCK)), dim3(WAVES_PER_BLOCK * 64), 0, ctx->stream, ap);
        KCHK("psf_kernel<BIAS>");
        return (int)SVR_OK; });
      if (rr) return rr; }
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}

int svr_normalise_bias_finish(svr_ctx *ctx, float sigma_bias) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  NEED(!ctx->disable_bias && ctx->d_bias_vol && ctx->maskC_valid, "bias buffers not ready");
  const size_t nv = ctx->nv;
  const dim3 grid((ctx->vx + 63) / 64, (ctx->vy + 3) / 4, ctx->vz);
  hipLaunchKernelGGL(k_div_s, dim3(nblk(nv)), dim3(256), 0, ctx->stream, ctx->d_bias_vol, ctx->volw(), nv);   // RC.cu:2553-2556
  // the reference's mbuf is uninitialised device memory (RC.cu:2563-2564); zero it so a NaN result is defined
  HIPCHK(hipMemsetAsync(ctx->d_mbuf, 0, nv * sizeof(float), ctx->stream));
  hipLaunchKernelGGL(k_gauss_conv3d, grid, dim3(256), 0, ctx->stream, ctx->d_bias_vol, ctx->d_mbuf, sigma_bias, 0,
                     ctx->vdim[0], (int)ctx->vx, (int)ctx->vy, (int)ctx->vz);
  hipLaunchKernelGGL(k_gauss_conv3d, grid, dim3(256), 0, ctx->stream, ctx->d_mbuf, ctx->d_bias_vol, sigma_bias, 1,
                     ctx->vdim[1], (int)ctx->vx, (int)ctx->vy, (int)ctx->vz);
  hipLaunchKernelGGL(k_gauss_conv3d, grid, dim3(256), 0, ctx->stream, ctx->d_bias_vol, ctx->d_mbuf, sigma_bias, 2,
                     ctx->vdim[2], (int)ctx->vx, (int)ctx->vy, (int)ctx->vz);
  KCHK("k_gauss_conv3d");
  HIPCHK(hipMemcpyAsync(ctx->d_bias_vol, ctx->d_mbuf, nv * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  hipLaunchKernelGGL(k_div_s, dim3(nblk(nv)), dim3(256), 0, ctx->stream, ctx->d_bias_vol, ctx->d_maskC, nv);   // RC.cu:2575-2577
  hipLaunchKernelGGL(k_divexp, dim3(nblk(nv)), dim3(256), 0, ctx->stream, ctx->recon(), ctx->d_bias_vol, nv);  // RC.cu:2579-2581
  KCHK("k_divexp");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}

int svr_normalise_bias(svr_ctx *ctx, int iter, float sigma_bias) {
  SVR_ENTER(ctx);
  (void)iter;
  int r = svr_normalise_bias_local(ctx);
  if (r) return r;
  return svr_normalise_bias_finish(ctx, sigma_bias);
}

// ---- slice-to-volume registration cost --------------------------------------------------
int svr_ncc_set_targets(svr_ctx *ctx, int n, int tx, int ty, const int16_t *targets) {
  SVR_ENTER(ctx);
  if (!ctx || !targets || n <= 0 || tx <= 0 || ty <= 0) return SVR_E_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  free_dev(ctx->d_reg_targets);
  const size_t bytes = (size_t)n * tx * ty * sizeof(short);
  HIPCHK(hipMalloc(&ctx->d_reg_targets, bytes));
  HIPCHK(hipMemcpyAsync(ctx->d_reg_targets, targets, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->reg_n = n; ctx->reg_tx = tx; ctx->reg_ty = ty;
  return SVR_OK;
}

int svr_ncc_set_source(svr_ctx *ctx, const uint32_t size[3], const int16_t *source_or_null) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  if (!source_or_null) NEED(ctx->nv > 0, "no source given and no reconstruction volume");
  uint32_t sx = source_or_null ? size[0] : ctx->vx, sy = source_or_null ? size[1] : ctx->vy,
           sz = source_or_null ? size[2] : ctx->vz;
  const size_t n = (size_t)sx * sy * sz;
  if (n == 0) return SVR_E_ARG;
  free_dev(ctx->d_reg_source);
  HIPCHK(hipMalloc(&ctx->d_reg_source, n * sizeof(short)));
  if (source_or_null) {
    HIPCHK(hipMemcpyAsync(ctx->d_reg_source, source_or_null, n * sizeof(short), hipMemcpyHostToDevice, ctx->stream));
  } else {
    // irtkGreyImage source = _reconstructed  (RG.cc:2031): static_cast<short> of every voxel
    hipLaunchKernelGGL(k_f32_to_i16, dim3(nblk(n)), dim3(256), 0, ctx->stream, ctx->recon(), ctx->d_reg_source, n);
    KCHK("k_f32_to_i16");
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->reg_vx = sx; ctx->reg_vy = sy; ctx->reg_vz = sz;
  return SVR_OK;
}

int svr_ncc_evaluate(svr_ctx *ctx, int n_eval, const int *target_index, const double *matrices,
                     int64_t *sums6, double *ncc) {
  SVR_ENTER(ctx);
  if (!ctx || n_eval <= 0 || !target_index || !matrices) return SVR_E_ARG;
  NEED(ctx->d_reg_targets && ctx->d_reg_source, "svr_ncc_set_targets / svr_ncc_set_source first");
  for (int i = 0; i < n_eval; ++i)
    if (target_index[i] < 0 || target_index[i] >= ctx->reg_n) return fail(ctx, SVR_E_ARG, "target index out of range");
  // grow-only scratch of the context (the optimiser calls this ~60 k times per registration pass)
  if ((size_t)n_eval > ctx->ncc_cap) {
    free_dev(ctx->d_ncc_idx); free_dev(ctx->d_ncc_m); free_dev(ctx->d_ncc_s);
    ctx->ncc_cap = 0;
    const size_t cap = std::max<size_t>(256, (size_t)n_eval * 3 / 2);
    HIPCHK(hipMalloc(&ctx->d_ncc_idx, cap * sizeof(int)));
    HIPCHK(hipMalloc(&ctx->d_ncc_m, cap * 16 * sizeof(double)));
    HIPCHK(hipMalloc(&ctx->d_ncc_s, cap * 6 * sizeof(long long)));
    ctx->ncc_cap = cap;
  }
  int *d_idx = ctx->d_ncc_idx;
  double *d_m = ctx->d_ncc_m;
  long long *d_s = ctx->d_ncc_s;
  std::vector<long long> h((size_t)n_eval * 6);
  hipError_t e = hipMemcpyAsync(d_idx, target_index, n_eval * sizeof(int), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_m, matrices, (size_t)n_eval * 16 * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_ncc, dim3(n_eval), dim3(256), 0, ctx->stream, ctx->d_reg_targets, ctx->reg_tx, ctx->reg_ty,
                       d_idx, d_m, ctx->d_reg_source, (int)ctx->reg_vx, (int)ctx->reg_vy, (int)ctx->reg_vz, d_s);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_s, h.size() * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return fail(ctx, (int)e, "svr_ncc_evaluate");
  for (int i = 0; i < n_eval; ++i) {
    const long long *s = &h[6 * (size_t)i];
    if (sums6) for (int k = 0; k < 6; ++k) sums6[6 * (size_t)i + k] = s[k];
    if (ncc) {
      // irtkCrossCorrelationSimilarityMetric::Evaluate (…Metric.h:158-165)
      const double n = (double)s[0], x = (double)s[1], y = (double)s[2], x2 = (double)s[3], y2 = (double)s[4],
                   xy = (double)s[5];
      ncc[i] = n > 0 ? (xy - (x * y) / n) / (sqrt(x2 - x * x / n) * sqrt(y2 - y * y / n)) : 0.0;
    }
  }
  return SVR_OK;
}

// ---- measurement -----------------------------------------------------------------------
int svr_timer_enable(svr_ctx *ctx, int enable) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  timers_resolve(ctx);
  ctx->timers = enable != 0;
  return SVR_OK;
}
int svr_timer_reset(svr_ctx *ctx) {
  SVR_ENTER(ctx);
  if (!ctx) return SVR_E_ARG;
  timers_resolve(ctx);
  for (int i = 0; i < SVR_T_COUNT; ++i) { ctx->t_ms[i] = 0; ctx->t_n[i] = 0; }
  return SVR_OK;
}
int svr_timer_get(svr_ctx *ctx, int which, double *ms_total, long *launches) {
  SVR_ENTER(ctx);
  if (!ctx || which < 0 || which >= SVR_T_COUNT) return SVR_E_ARG;
  timers_resolve(ctx);
  if (ms_total) *ms_total = ctx->t_ms[which];
  if (launches) *launches = ctx->t_n[which];
  return SVR_OK;
}
int svr_unit_counts(svr_ctx *ctx, uint64_t out3[3]) {
  SVR_ENTER(ctx);
  if (!ctx || !out3) return SVR_E_ARG;
  int r = ready(ctx);
  if (r) return r;
  if ((r = cell_prepare(ctx))) return r;
  const CellState &cs = *ctx->cell;
  const uint64_t ns = ctx->pvr ? PVR_N : PSF_SUPPORT;
  out3[0] = cs.n_sorted;
  out3[1] = (uint64_t)cs.n_sorted * ns - cs.dead_units;
  out3[2] = cs.dead_units;
  return SVR_OK;
}
// what the cell lists of the current slice geometry hold (the per-item fixed costs of a sharded run: bench.py --shard)
int svr_cell_stats(svr_ctx *ctx, uint64_t out8[8]) {
  SVR_ENTER(ctx);
  if (!ctx || !out8) return SVR_E_ARG;
  int r = ready(ctx);
  if (r) return r;
  for (int k = 0; k < 8; ++k) out8[k] = 0;
  if ((r = cell_prepare(ctx))) return r;
  const CellState &cs = *ctx->cell;
  out8[0] = cs.a.nitems; out8[1] = cs.n_runs; out8[2] = cs.n_sorted;
  out8[3] = (uint64_t)cs.a.nents * cs.a.PP * sizeof(f2);      // staging bytes written by the scatter, read by the combine
  CellState *g = nullptr;
  if ((r = cell_prepare_gather(ctx, g))) return r;
  if (g) { out8[4] = g->a.nitems; out8[5] = g->n_runs; out8[6] = (uint64_t)g->n_sorted * (ctx->pvr ? PVR_N : PSF_SUPPORT) * sizeof(f2); }
  out8[7] = (uint64_t)cs.a.CX << 32 | (uint64_t)cs.a.CL;
  return SVR_OK;
}
int svr_timer_begin(svr_ctx *ctx, int which) {
  SVR_ENTER(ctx);
  if (!ctx || which < 0 || which >= SVR_T_COUNT) return SVR_E_ARG;
  if (ctx->timers) {
    if (!ctx->ev_open) ctx->ev_open = timer_event(ctx);
    if (ctx->ev_open) HIPCHK(hipEventRecord(ctx->ev_open, ctx->stream));
  }
  return SVR_OK;
}
int svr_timer_end(svr_ctx *ctx, int which) {
  SVR_ENTER(ctx);
  if (!ctx || which < 0 || which >= SVR_T_COUNT) return SVR_E_ARG;
  if (ctx->timers && ctx->ev_open) {
    timer_close(ctx, which, ctx->ev_open);
    ctx->ev_open = nullptr;
  }
  return SVR_OK;
}
int svr_timer_add(svr_ctx *ctx, int which, double ms) {
  if (!ctx || which < 0 || which >= SVR_T_COUNT) return SVR_E_ARG;
  if (ctx->timers) { ctx->t_ms[which] += ms; ctx->t_n[which] += 1; }
  return SVR_OK;
}
// A fixed amount of dependent f32 work on every SIMD of the chip (4 wavefronts per SIMD, each a chain of `n` fmas): its time is a fixed number of
// cycles over the shader clock the device sustains under a full vector load -- what differs between the boxes of a pool (power caps, temperature)
// when the same binary reads 158 on one and 189 MVoxels/s on another.  bench.py reports it next to the headline.
__global__ __launch_bounds__(256) void k_clock_probe(float *out, int n) {
  // eight independent chains of packed fmas per lane: the vector pipe at its full issue rate, i.e. the power draw of the PSF kernels -- a box
  // that holds its clock on a light load and drops it on this one is what the probe is there to show
  f2 x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = (f2){1.0f + 1.0e-7f * (float)(threadIdx.x + k), 1.0f - 1.0e-7f * (float)(threadIdx.x + k)};
  const f2 a = (f2){0.99999994f, 1.00000012f}, b = (f2){1.0e-9f, -1.0e-9f};
  for (int i = 0; i < n; i += 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = fma2(x[k], a, b);
  }
  f2 t = x[0];
#pragma unroll
  for (int k = 1; k < 8; ++k) t = t + x[k];
  if (t.x == 12345.678f && t.y == 1.0f) out[blockIdx.x] = t.x;   // (never: keeps the chains)
}
int svr_clock_probe(svr_ctx *ctx, int chain, double *ms) {
  SVR_ENTER(ctx);
  if (!ctx || !ms || chain <= 0) return SVR_E_ARG;
  float *d = nullptr;
  HIPCHK(hipMalloc(&d, 4096 * sizeof(float)));
  hipEvent_t a = nullptr, b = nullptr;
  hipError_t e = hipEventCreate(&a);
  if (e == hipSuccess) e = hipEventCreate(&b);
  float best = 3.0e38f;
  for (int rep = 0; rep < 4 && e == hipSuccess; ++rep) {   // the first launch warms the clocks up; the shortest of the rest
    e = hipEventRecord(a, ctx->stream);
    hipLaunchKernelGGL(k_clock_probe, dim3(1024), dim3(256), 0, ctx->stream, d, chain);   // 256 CUs x 4 workgroups of 4 wavefronts: 4 per SIMD
    if (e == hipSuccess) e = hipEventRecord(b, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(b);
    float t = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, a, b);
    if (rep > 0 && t < best) best = t;
  }
  if (a) (void)hipEventDestroy(a);
  if (b) (void)hipEventDestroy(b);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(ctx, (int)e, std::string("svr_clock_probe: ") + hipGetErrorString(e));
  *ms = best;
  return SVR_OK;
}
int svr_fallbacks(svr_ctx *ctx, uint64_t out4[4]) {
  SVR_ENTER(ctx);
  if (!ctx || !out4) return SVR_E_ARG;
  for (int k = 0; k < 4; ++k) out4[k] = ctx->fallbacks[k];
  return SVR_OK;
}

int svr_counters(svr_ctx *ctx, uint64_t out5[8]) {
  SVR_ENTER(ctx);
  if (!ctx || !out5) return SVR_E_ARG;
  if (ctx->have_slices) {
    int r = ensure_tiles_back(ctx);
    if (r) return r;
  }
  out5[0] = ctx->np; out5[1] = ctx->n_active; out5[2] = ctx->n_psf; out5[3] = ctx->nv; out5[4] = ctx->ns;
  out5[5] = ctx->n_tiles; out5[6] = ctx->n_tiles_fb; out5[7] = ctx->n_tiles_fb8;
  return SVR_OK;
}

}  // extern "C"

#include "svr_reg.inc"
#include "svr_pyr.inc"
#include "svr_em.inc"
