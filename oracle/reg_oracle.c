/*
 * reg_oracle.c -- CPU restatement of the reference's GPU slice-to-volume registration
 * (SURVEY 8a17 + 8f1): genenerateRegistrationSlices, FilterGaussStack, averageIf,
 * computeNCCAndReduce, addNccValues, writeSimilarities, evaluateCostsMultipleSlices and the
 * batched gradient-ascent optimiser registerMultipleSlicesToVolume.
 *
 * TEST INFRASTRUCTURE ONLY (see svr_oracle.c header); PARITY UNPINNED for the same reasons.
 *
 * Citations: RC.cu = /root/reference/source/reconstructionGPU2/reconstruction_cuda2.cu,
 *            GF.cu = /root/reference/source/reconstructionGPU2/GPUGauss/gaussfilter.cu.
 *
 * The restatement keeps the reference's temp-buffer layout and its memset LITERALLY
 * (evaluateCostsMultipleSlices, RC.cu:4150-4221), because they change the numbers:
 *   - dev_temp_float is float[6*slices]; with a = active_slices the call uses
 *       [0,a)   sum of target values        [a,2a)  sum of sampled values
 *       [2a,3a) accumulated NCC             [3a,6a) the three NCC moments per slot,
 *     and dev_temp_int [0,a) / [a,2a) the matching counts.  Only [0,6a) is cleared at entry.
 *   - the sampled-slice sums/counts are NOT cleared between the three through-plane offsets,
 *     so the "mean" of offset o is the mean over offsets <= o (RC.cu:4195).
 *   - per offset the code clears float[2*slices, 5*slices) -- indexed by `slices`, not by
 *     `active_slices` (RC.cu:4200) -- which wipes part of the accumulated NCC and part of the
 *     moments depending on slot index, a and slices.  With a == slices the accumulated NCC of
 *     every slot is wiped before offsets 0 and +1, and the moments of slots >= 2a/3 are never
 *     wiped.  Reproduced as is.
 * Not reproduced (no effect on the value or undefined there):
 *   - averageIf / computeNCCAndReduce let all 32 threads with threadIdx.x == 0 add the block
 *     result (RC.cu:4491-4495, 4544-4549): sums, counts and moments are 32x too large, which
 *     cancels in mean = sum/count and in m0/sqrt(m1*m2).  Here they are added once.
 *   - float atomics in unspecified order: here sums are accumulated in double and rounded once.
 *   - checkImprovement indexes its shared prefix-sum array with the global thread id
 *     (RC.cu:4408), undefined for > 512 slices per device; here compaction is stable for any n.
 *   - texture filtering hardware arithmetic is not public.  tex3D with normalised coordinates,
 *     linear filter and border mode is restated from the CUDA programming guide (appendix
 *     "Texture Fetching"): xB = N*u - 0.5, i = floor(xB), alpha = frac(xB) kept to 8 fractional
 *     bits, zero outside the array; the 8 products are summed in the guide's order in float.
 *   - sin/cos/asin/atan2 of the per-slice parameter updates are evaluated in double and rounded
 *     to float (the device intrinsics of the reference are not bit-specified either).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct orc_reg {
  int W, H, slices;           /* resampled slice grid (RC.cu:4893-4949) */
  int vx, vy, vz;             /* volume */
  const float *volume;        /* snapshot bound to reconstructedTex_ (RC.cu:3909-3959) */
  const float *reconW2I;      /* d_reconstructedW2I */
  const float *ofs;           /* [slices][16] dev_d_slicesOfs (RC.cu:4742-4757) */
  const float *resampled;     /* [slices][H][W] dev_v_slices_resampled */
  float *resampled_float;     /* [slices][H][W] blurred targets */
  float *reg;                 /* [slices][H][W] dev_regSlices (layer = slot) */
  float *tmp;                 /* [slices][H][W] dev_temp_slices */
  float *matrices;            /* [slices][16] dev_recon_matrices */
  float *matrices_orig;       /* [slices][16] */
  float *similarities;        /* [5][slices] */
  float *gradient;            /* [7][slices] */
  int *active, *active2, *active_prev;
  float *temp_float;          /* [6*slices] */
  int *temp_int;              /* [2*slices] */
  int levels, steps, iterations;
  float epsilon;
  float blurring[8], length_of_steps[8];
  float *reg_dbg;             /* optional [3][slices][H][W]: blurred sampled slices of the last evaluate */
} orc_reg;

static void matvec3(const float *M, const float v[3], float out[3]) {   /* RVH:134-145 */
  float a = M[0] * v[0] + M[1] * v[1] + M[2] * v[2] + M[3];
  float b = M[4] * v[0] + M[5] * v[1] + M[6] * v[2] + M[7];
  float c = M[8] * v[0] + M[9] * v[1] + M[10] * v[2] + M[11];
  out[0] = a; out[1] = b; out[2] = c;
}

/* generateGaussianKernel + the length rule of FilterGaussStack (GF.cu:56-87,195-196).
 * Returns klength; half[i] = kernel[mid + i], i < (klength+1)/2. */
int orc_reg_gauss_kernel(float sigma, float *half) {
  int klength = (int)(sigma * 5);
  if (klength > 64 - 1) klength = 64 - 1;       /* MAX_LENGTH_SK = BLOCK_SIZE_SK_1-1 */
  if (klength < 7) klength = 7;
  klength -= 1 - klength % 2;
  float k[64];
  float sum = 0;
  int mid = klength / 2;
  for (int i = 0; i < klength; ++i) {
    int d = abs(i - mid);
    k[i] = expf(-(float)d * (float)d / (2 * sigma * sigma));
    sum += k[i];
  }
  for (int i = 0; i < klength; ++i) k[i] /= sum;
  for (int i = 0; i < (klength + 1) / 2; ++i) half[i] = k[mid + i];
  return klength;
}

static inline int clampi(int v, int n) { return v < 0 ? 0 : (v > n - 1 ? n - 1 : v); }
static inline float max0(float v) { return v > 0.0f ? v : 0.0f; }

/* GaussXKernel / GaussYKernel (GF.cu:92-173): pixels equal to -1 stay, the others get the
 * un-normalised convolution with clamped addressing and negative neighbours read as 0. */
void orc_reg_blur_stack(float *img, float *tmp, int W, int H, int n, float sigma) {
  float half[32];
  int klength = orc_reg_gauss_kernel(sigma, half);
  int nh = (klength + 1) / 2;
  for (int z = 0; z < n; ++z) {
    float *in = img + (size_t)z * W * H, *t = tmp + (size_t)z * W * H;
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        float v = in[y * W + x];
        if (v != -1) {
          v = v * half[0];
          for (int i = 1; i < nh; ++i)
            v = v + half[i] * (max0(in[y * W + clampi(x + i, W)]) + max0(in[y * W + clampi(x - i, W)]));
        }
        t[y * W + x] = v;
      }
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        float v = t[y * W + x];
        if (v != -1) {
          v = v * half[0];
          for (int i = 1; i < nh; ++i)
            v = v + half[i] * (max0(t[clampi(y + i, H) * W + x]) + max0(t[clampi(y - i, H) * W + x]));
        }
        in[y * W + x] = v;
      }
  }
}

static inline float vol_fetch(const orc_reg *r, int x, int y, int z) {   /* cudaAddressModeBorder */
  if (x < 0 || y < 0 || z < 0 || x >= r->vx || y >= r->vy || z >= r->vz) return 0.0f;
  return r->volume[(size_t)x + (size_t)y * r->vx + (size_t)z * r->vx * r->vy];
}

static inline void tex_axis(float p, float n, int *i, float *a) {
  float u = p / n;                 /* the kernel passes normalised coordinates (RC.cu:3525) */
  float xb = u * n - 0.5f;
  float fl = floorf(xb);
  float fr = xb - fl;
  *i = (int)fl;
  *a = floorf(fr * 256.0f + 0.5f) / 256.0f;
}

float orc_reg_tex3d(const orc_reg *r, const float p[3]) {
  int i, j, k;
  float a, b, c;
  tex_axis(p[0], (float)r->vx, &i, &a);
  tex_axis(p[1], (float)r->vy, &j, &b);
  tex_axis(p[2], (float)r->vz, &k, &c);
  float oa = 1.0f - a, ob = 1.0f - b, oc = 1.0f - c;
  float t = oa * ob * oc * vol_fetch(r, i, j, k);
  t = t + a * ob * oc * vol_fetch(r, i + 1, j, k);
  t = t + oa * b * oc * vol_fetch(r, i, j + 1, k);
  t = t + a * b * oc * vol_fetch(r, i + 1, j + 1, k);
  t = t + oa * ob * c * vol_fetch(r, i, j, k + 1);
  t = t + a * ob * c * vol_fetch(r, i + 1, j, k + 1);
  t = t + oa * b * c * vol_fetch(r, i, j + 1, k + 1);
  t = t + a * b * c * vol_fetch(r, i + 1, j + 1, k + 1);
  return t;
}

/* genenerateRegistrationSlices RC.cu:3504-3529: layer = slot (blockIdx.z), slice = activelayers[slot] */
static void generate_slices(orc_reg *r, int active_slices, int insofs) {
  for (int slot = 0; slot < active_slices; ++slot) {
    int sl = r->active[slot];
    const float *T = r->matrices + 16 * (size_t)sl, *O = r->ofs + 16 * (size_t)sl;
    float *out = r->reg + (size_t)slot * r->W * r->H;
    for (int y = 0; y < r->H; ++y)
      for (int x = 0; x < r->W; ++x) {
        float sp[3] = {(float)x, (float)y, (float)(insofs * 2)}, w[3], w2[3], vp[3];
        matvec3(O, sp, w);
        matvec3(T, w, w2);
        matvec3(r->reconW2I, w2, vp);
        float val = orc_reg_tex3d(r, vp);
        if (val < 0) val = -1.0f;
        out[y * r->W + x] = val;
      }
  }
}

/* averageIf RC.cu:4459-4496 (added once, double accumulation; see header) */
static void average_if(const float *layers, const int *activelayers, int n, int W, int H, float *sum, int *count) {
  for (int slot = 0; slot < n; ++slot) {
    int sl = activelayers ? activelayers[slot] : slot;
    const float *p = layers + (size_t)sl * W * H;
    double acc = 0;
    int c = 0;
    for (int i = 0; i < W * H; ++i)
      if (p[i] > -1.0f) { ++c; acc += p[i]; }
    sum[slot] += (float)acc;
    count[slot] += c;
  }
}

/* evaluateCostsMultipleSlices RC.cu:4150-4221 */
void orc_reg_evaluate_costs(orc_reg *r, int active_slices, int level, float blurring, int writeoffset,
                            int writestep, int writenum) {
  const int a = active_slices, s = r->slices, W = r->W, H = r->H;
  if (a == 0) return;
  float *F = r->temp_float;
  int *I = r->temp_int;
  memset(F, 0, sizeof(float) * 6 * a);                                        /* :4159 */
  memset(I, 0, sizeof(int) * 2 * a);                                          /* :4160 */
  average_if(r->resampled_float, r->active, a, W, H, F, I);                   /* :4164 */
  for (int insofs = -1; insofs <= 1; ++insofs) {
    generate_slices(r, a, insofs);                                            /* :4188 */
    orc_reg_blur_stack(r->reg, r->tmp, W, H, a, blurring);                    /* :4192 */
    if (r->reg_dbg)
      memcpy(r->reg_dbg + (size_t)(insofs + 1) * s * W * H, r->reg, sizeof(float) * (size_t)a * W * H);
    average_if(r->reg, NULL, a, W, H, F + a, I + a);                          /* :4195, no reset */
    for (int i = 2 * s; i < 5 * s; ++i) F[i] = 0;                             /* :4200, `slices` not `active_slices` */
    /* computeNCCAndReduce RC.cu:4498-4550 */
    const int lv = level + 1;
    for (int slot = 0; slot < a; ++slot) {
      const float *A = r->resampled_float + (size_t)r->active[slot] * W * H;
      const float *B = r->reg + (size_t)slot * W * H;
      float avg_a = F[slot], avg_b = F[a + slot];
      if (avg_a != 0) avg_a /= I[slot];
      if (avg_b != 0) avg_b /= I[a + slot];
      double m0 = 0, m1 = 0, m2 = 0;
      for (int lin = 0; lin < W * H; ++lin) {
        float va = A[lin], vb = B[lin];
        if (va >= 0.0f && vb >= 0.0f && lin % lv == 0) {
          float sa = va - avg_a, sb = vb - avg_b;
          m0 += (double)(sa * sb);
          m1 += (double)(sa * sa);
          m2 += (double)(sb * sb);
        }
      }
      float *R = F + 3 * a + 3 * slot;
      R[0] = R[0] + (float)m0;
      R[1] = R[1] + (float)m1;
      R[2] = R[2] + (float)m2;
    }
    /* addNccValues RC.cu:4552-4563 */
    for (int slot = 0; slot < a; ++slot) {
      const float *R = F + 3 * a + 3 * slot;
      float norm = R[1] * R[2];
      float res = 0;
      if (norm > 0) res = R[0] / sqrtf(norm);
      F[2 * a + slot] += res;
    }
  }
  /* writeSimilarities RC.cu:4565-4575 */
  for (int slot = 0; slot < a; ++slot) {
    float res = F[2 * a + slot];
    int sl = r->active[slot];
    for (int i = 0; i < writenum; ++i) r->similarities[(size_t)writeoffset * s + (size_t)s * writestep * i + sl] = res;
  }
}

static float f_sin(float x) { return (float)sin((double)x); }
static float f_cos(float x) { return (float)cos((double)x); }
static float f_asin(float x) { return (float)asin((double)x); }
static float f_atan2(float y, float x) { return (float)atan2((double)y, (double)x); }

static void euler_of(const float *m, float p_rot[3]) {      /* RC.cu:4254-4269 = 4352-4367 */
  const float TOL = 0.000001f;
  float tmp = f_asin(-1.0f * m[2]);
  if (fabsf(f_cos(tmp)) > TOL) {
    p_rot[0] = f_atan2(m[6], m[10]);
    p_rot[1] = tmp;
    p_rot[2] = f_atan2(m[1], m[0]);
  } else {
    p_rot[0] = f_atan2(-m[2] * m[4], -m[2] * m[8]);
    p_rot[1] = tmp;
    p_rot[2] = 0;
  }
}

static void rot_of(const float p_rot[3], float *m) {        /* RC.cu:4274-4291 = 4373-4390 */
  float cosrx = f_cos(p_rot[0]), cosry = f_cos(p_rot[1]), cosrz = f_cos(p_rot[2]);
  float sinrx = f_sin(p_rot[0]), sinry = f_sin(p_rot[1]), sinrz = f_sin(p_rot[2]);
  m[0] = cosry * cosrz;
  m[1] = cosry * sinrz;
  m[2] = -sinry;
  m[4] = (sinrx * sinry * cosrz - cosrx * sinrz);
  m[5] = (sinrx * sinry * sinrz + cosrx * cosrz);
  m[6] = sinrx * cosry;
  m[8] = (cosrx * sinry * cosrz + sinrx * sinrz);
  m[9] = (cosrx * sinry * sinrz - sinrx * cosrz);
  m[10] = cosrx * cosry;
}

/* adjustSamplingMatrixForCentralDifferences RC.cu:4231-4293 */
void orc_reg_adjust(const float *in, float *out, int part, float step) {
  const float pi = 3.14159265358979323846f;
  memcpy(out, in, 16 * sizeof(float));
  if (part < 3) {
    out[4 * part + 3] = in[4 * part + 3] + step;
  } else {
    float p_rot[3];
    euler_of(in, p_rot);
    p_rot[part - 3] += step * pi / 180.0f;
    rot_of(p_rot, out);
  }
}

/* gradientStep RC.cu:4335-4391 */
void orc_reg_gradient_step(float *m, const float g[6], float step) {
  const float pi = 3.14159265358979323846f;
  for (int p = 0; p < 3; ++p) m[4 * p + 3] = m[4 * p + 3] + step * g[p];
  float p_rot[3];
  euler_of(m, p_rot);
  for (int p = 0; p < 3; ++p) p_rot[p] += g[p + 3] * step * pi / 180.0f;
  rot_of(p_rot, m);
}

/* checkImprovement RC.cu:4393-4457 (stable compaction) */
static int check_improvement(orc_reg *r, int *new_mask, int n, const int *mask, int cursim, int prev, float eps) {
  int c = 0;
  const int s = r->slices;
  for (int i = 0; i < n; ++i) {
    int sl = mask[i];
    if (r->similarities[(size_t)cursim * s + sl] > r->similarities[(size_t)prev * s + sl] + eps) new_mask[c++] = sl;
  }
  return c;
}

/* prepareSliceToVolumeReg RC.cu:3884-3900 */
void orc_reg_prepare(orc_reg *r, float recon_dim_x) {
  r->levels = 2; r->steps = 4; r->iterations = 20; r->epsilon = 0.0001f;
  r->blurring[0] = recon_dim_x / 2.0f;
  for (int i = 0; i < r->levels; ++i) r->length_of_steps[i] = (float)(0.1 * pow(2.0f, i));
  for (int i = 1; i < r->levels; ++i) r->blurring[i] = r->blurring[i - 1] * 2;
}

/* start of a level, RC.cu:4016-4026: targets = blurred copy of the resampled slices */
void orc_reg_begin_level(orc_reg *r, int level) {
  size_t n = (size_t)r->slices * r->W * r->H;
  memcpy(r->resampled_float, r->resampled, n * sizeof(float));
  orc_reg_blur_stack(r->resampled_float, r->tmp, r->W, r->H, r->slices, r->blurring[level]);
}

/* registerMultipleSlicesToVolume RC.cu:4001-4141; transf [slices][16] in/out.
 * counters (optional, int[4]): evaluate calls, line-search steps, outer iterations, slot-evaluations */
void orc_reg_register(orc_reg *r, float *transf, long long *counters) {
  const int s = r->slices;
  memcpy(r->matrices, transf, sizeof(float) * 16 * s);
  memcpy(r->matrices_orig, transf, sizeof(float) * 16 * s);
  long long n_eval = 0, n_ls = 0, n_it = 0, n_slot = 0;
  for (int level = r->levels - 1; level >= 0; --level) {
    float blur = r->blurring[level];
    float step = r->length_of_steps[level];
    orc_reg_begin_level(r, level);
    for (int st = 0; st < r->steps; ++st) {
      for (int i = 0; i < s; ++i) r->active[i] = i;                                   /* initActiveSlices */
      int active = s;
      for (int iter = 0; iter < r->iterations; ++iter) {
        ++n_it;
        orc_reg_evaluate_costs(r, active, level, blur, 0, 1, 3); ++n_eval; n_slot += active;
        for (int p = 0; p < 6; ++p) {
          for (int i = 0; i < active; ++i) {
            int sl = r->active[i];
            orc_reg_adjust(r->matrices_orig + 16 * (size_t)sl, r->matrices + 16 * (size_t)sl, p, step);
          }
          orc_reg_evaluate_costs(r, active, level, blur, 3, 0, 1); ++n_eval; n_slot += active;
          for (int i = 0; i < active; ++i) {
            int sl = r->active[i];
            orc_reg_adjust(r->matrices_orig + 16 * (size_t)sl, r->matrices + 16 * (size_t)sl, p, -step);
          }
          orc_reg_evaluate_costs(r, active, level, blur, 4, 0, 1); ++n_eval; n_slot += active;
          for (int i = 0; i < active; ++i) {                                          /* computeGradientCentralDiff */
            int sl = r->active[i];
            float dx = r->similarities[3 * (size_t)s + sl] - r->similarities[4 * (size_t)s + sl];
            r->gradient[(size_t)p * s + sl] = dx;
            if (p == 0) r->gradient[6 * (size_t)s + sl] = dx * dx;
            else r->gradient[6 * (size_t)s + sl] += dx * dx;
          }
        }
        for (int i = 0; i < active; ++i) {                                            /* normalizeGradient */
          int sl = r->active[i];
          float norm = r->gradient[6 * (size_t)s + sl];
          if (norm > 0) norm = 1.0f / sqrtf(norm);
          for (int j = 0; j < 6; ++j) r->gradient[(size_t)j * s + sl] *= norm;
        }
        int prev_active = active;
        memcpy(r->active_prev, r->active, sizeof(int) * active);
        do {
          for (int i = 0; i < active; ++i) {                                          /* copySimilarity 2 <- 0 */
            int sl = r->active[i];
            r->similarities[2 * (size_t)s + sl] = r->similarities[sl];
          }
          for (int i = 0; i < active; ++i) {
            int sl = r->active[i];
            float g[6];
            for (int j = 0; j < 6; ++j) g[j] = r->gradient[(size_t)j * s + sl];
            orc_reg_gradient_step(r->matrices + 16 * (size_t)sl, g, step);
          }
          orc_reg_evaluate_costs(r, active, level, blur, 0, 1, 1); ++n_eval; ++n_ls; n_slot += active;
          active = check_improvement(r, r->active2, active, r->active, 0, 2, r->epsilon);
          int *t = r->active; r->active = r->active2; r->active2 = t;
        } while (active > 0);
        for (int i = 0; i < prev_active; ++i) {                                       /* back track */
          int sl = r->active_prev[i];
          float g[6];
          for (int j = 0; j < 6; ++j) g[j] = r->gradient[(size_t)j * s + sl];
          orc_reg_gradient_step(r->matrices + 16 * (size_t)sl, g, -step);
        }
        memcpy(r->matrices_orig, r->matrices, sizeof(float) * 16 * s);
        active = check_improvement(r, r->active, prev_active, r->active_prev, 2, 1, r->epsilon);
        if (active == 0) break;
      }
      step /= 2.0f;
    }
  }
  memcpy(transf, r->matrices, sizeof(float) * 16 * s);
  if (counters) { counters[0] = n_eval; counters[1] = n_ls; counters[2] = n_it; counters[3] = n_slot; }
}

/* ---- PVR patch-to-volume registration cost (second variant) ------------------------
 * computeCCpatch, R2/patchBased2D3DRegistration_gpu2.cu:130-190 (R2 = /root/reference/source/
 * reconstructionGPU2): raw-moment NCC in float of one (blurred) patch against the volume sampled with
 * the software trilinear `interp` of R2/include/interpFunctions.cuh:81-96 at the three through-plane
 * offsets z = -1, 0, 1 of the patch grid, every (level+1)-th pixel.  Sequential float sums, like the
 * single thread per patch of the reference (parallelPatchRegOptimization :199-). */
static float interp_sw(const float p[3], const float *data, int sx, int sy, int sz) {
  /* lower = max(floor, 0), upper = min(floor + 1, size - 1); v() reads 0 for an index that is out of
   * range as an unsigned number -- so a negative `upper` reads 0 while `lower` clamps to voxel 0 */
  int b[3] = {(int)floorf(p[0]), (int)floorf(p[1]), (int)floorf(p[2])};
  float f[3] = {p[0] - floorf(p[0]), p[1] - floorf(p[1]), p[2] - floorf(p[2])};
  int size[3] = {sx, sy, sz}, lo[3], up[3];
  for (int k = 0; k < 3; ++k) {
    lo[k] = b[k] > 0 ? b[k] : 0;
    up[k] = b[k] + 1 < size[k] - 1 ? b[k] + 1 : size[k] - 1;
  }
#define V(X, Y, Z) (((unsigned)(X) < (unsigned)sx && (unsigned)(Y) < (unsigned)sy && (unsigned)(Z) < (unsigned)sz) \
                        ? data[(size_t)(X) + (size_t)(Y) * sx + (size_t)(Z) * sx * sy] : 0.0f)
  return V(lo[0], lo[1], lo[2]) * (1 - f[0]) * (1 - f[1]) * (1 - f[2])
       + V(up[0], lo[1], lo[2]) * f[0] * (1 - f[1]) * (1 - f[2])
       + V(lo[0], up[1], lo[2]) * (1 - f[0]) * f[1] * (1 - f[2])
       + V(up[0], up[1], lo[2]) * f[0] * f[1] * (1 - f[2])
       + V(lo[0], lo[1], up[2]) * (1 - f[0]) * (1 - f[1]) * f[2]
       + V(up[0], lo[1], up[2]) * f[0] * (1 - f[1]) * f[2]
       + V(lo[0], up[1], up[2]) * (1 - f[0]) * f[1] * f[2]
       + V(up[0], up[1], up[2]) * f[0] * f[1] * f[2];
#undef V
}

static void matmul4f(const float *A, const float *B, float *C) {   /* Matrix4 * Matrix4, RVH:148-159 order */
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      C[4 * i + j] = A[4 * i] * B[j] + A[4 * i + 1] * B[4 + j] + A[4 * i + 2] * B[8 + j] + A[4 * i + 3] * B[12 + j];
}

float orc_cc_patch(const float *buffer, int px, int py, const float *RI2W, const float *Tmat, const float *reconW2I,
                   const float *vol, int vx, int vy, int vz, int level, float sums6[6]) {
  float M[16];
  matmul4f(Tmat, RI2W, M);                       /* Tmat * patch.RI2W * patchPos (:152) */
  float xy = 0, y_ = 0, x_ = 0, y2 = 0, x2 = 0;
  unsigned n = 0;
  const int z0 = (int)(-(3 + 0.5f)) / 2, z1 = (int)((3 + 0.5f) / 2 + 1);   /* psize.z = 3 (:208): z = -1, 0, 1 */
  for (int y = 0; y < py; y += level + 1)
    for (int x = 0; x < px; x += level + 1)
      for (int z = z0; z < z1; ++z) {
        float a = buffer[y * px + x];
        float pos[3] = {(float)x, (float)y, (float)z}, w[3], vp[3];
        matvec3(M, pos, w);
        matvec3(reconW2I, w, vp);
        float b = interp_sw(vp, vol, vx, vy, vz);
        if (a >= 0.0f && b >= 0.0f && a == a && b == b) {
          xy += a * b; x_ += a; y_ += b; x2 += a * a; y2 += b * b; n++;
        }
      }
  if (sums6) { sums6[0] = (float)n; sums6[1] = x_; sums6[2] = y_; sums6[3] = x2; sums6[4] = y2; sums6[5] = xy; }
  if (n > 0) return (xy - (x_ * y_) / n) / (sqrtf(x2 - x_ * x_ / n) * sqrtf(y2 - y_ * y_ / n));
  return 0.0f;
}

/* ===================== patch-to-volume registration of the patch-based path =====================
 * PatchBased2D3DRegistration_gpu2<T>::run (R2/patchBased2D3DRegistration_gpu2.cu:450-566) with parallelPatchRegOptimization
 * (:198-291), Matrix2Parameters / Parameters2Matrix (R2/include/matrix4.cuh:189-260) and patchBasedFilterGaussStack
 * (R2/GPUGauss/patchBasedGaussfilter.cu:88-335).  The similarity of an evaluation is computeCCpatch with the six moments
 * accumulated in double in the order of the device's workgroup (256 strided partial sums, the wave's shuffle-down tree, four
 * waves added left to right) so that both sides take the same optimiser decisions; orc_cc_patch above is the reference's
 * sequential float accumulation, and tests compare the two. */
static void pvr_m2p(const float *m, float p[6]) {
  const float TOL = 0.000001f;
  p[0] = m[3]; p[1] = m[7]; p[2] = m[11];
  const float tmp = f_asin(-1.0f * m[2]);
  if (fabsf(f_cos(tmp)) > TOL) {
    p[3] = f_atan2(m[6], m[10]);
    p[4] = tmp;
    p[5] = f_atan2(m[1], m[0]);
  } else {
    p[3] = f_atan2(-1.0f * m[2] * m[4], -1.0f * m[2] * m[8]);
    p[4] = tmp;
    p[5] = 0;
  }
  for (int k = 3; k < 6; ++k) p[k] = (float)((double)p[k] * (180.0 / 3.14159265358979323846));
}
static void pvr_p2m(const float p[6], float *m) {
  const double k = 3.14159265358979323846 / 180.0;
  const float cosrx = (float)cos((double)p[3] * k), cosry = (float)cos((double)p[4] * k), cosrz = (float)cos((double)p[5] * k);
  const float sinrx = (float)sin((double)p[3] * k), sinry = (float)sin((double)p[4] * k), sinrz = (float)sin((double)p[5] * k);
  m[0] = cosry * cosrz; m[1] = cosry * sinrz; m[2] = -sinry; m[3] = p[0];
  m[4] = (sinrx * sinry * cosrz - cosrx * sinrz); m[5] = (sinrx * sinry * sinrz + cosrx * cosrz); m[6] = sinrx * cosry; m[7] = p[1];
  m[8] = (cosrx * sinry * cosrz + sinrx * sinrz); m[9] = (cosrx * sinry * sinrz - sinrx * cosrz); m[10] = cosrx * cosry; m[11] = p[2];
  m[12] = 0; m[13] = 0; m[14] = 0; m[15] = 1.0f;
}
void orc_pvr_params(const float *m, float p6[6], float *rebuilt16) { pvr_m2p(m, p6); pvr_p2m(p6, rebuilt16); }

/* GaussXKernel / GaussYKernel of the patch-based filter: a neighbour outside the patch contributes 0 (PGF.cu:107-111), -1 is
 * left alone, negative neighbours count as 0.  (The reference launches 32x32 thread blocks over each patch and lets the threads
 * beyond the patch width wrap into other rows: patch sizes that are not a multiple of 32 race there.) */
void orc_pvr_blur_patches(const float *in, float *out, int px, int py, int n, float sigma) {
  float half[40];
  memset(half, 0, sizeof(half));
  const int klen = orc_reg_gauss_kernel(sigma, half);
  int nh = (klen + 1) / 2, nh_y = nh;
  if (klen == 13) { half[nh] = half[nh - 1]; nh_y = nh + 1; }      /* GaussYKernel<14> for klength 13 (PGF.cu:311) */
  float *tmp = (float *)malloc(sizeof(float) * (size_t)px * py);
  for (int p = 0; p < n; ++p) {
    const float *src = in + (size_t)p * px * py;
    float *dst = out + (size_t)p * px * py;
    for (int y = 0; y < py; ++y)
      for (int x = 0; x < px; ++x) {
        float v = src[y * px + x];
        if (v != -1) {
          v = v * half[0];
          for (int i = 1; i < nh; ++i) {
            const float a = (x + i) < px ? max0(src[y * px + x + i]) : 0.0f;
            const float b = (x - i) >= 0 ? max0(src[y * px + x - i]) : 0.0f;
            v = v + half[i] * (a + b);
          }
        }
        tmp[y * px + x] = v;
      }
    for (int y = 0; y < py; ++y)
      for (int x = 0; x < px; ++x) {
        float v = tmp[y * px + x];
        if (v != -1) {
          v = v * half[0];
          for (int i = 1; i < nh_y; ++i) {
            const float a = (y + i) < py ? max0(tmp[(y + i) * px + x]) : 0.0f;
            const float b = (y - i) >= 0 ? max0(tmp[(y - i) * px + x]) : 0.0f;
            v = v + half[i] * (a + b);
          }
        }
        dst[y * px + x] = v;
      }
  }
  free(tmp);
}

/* computeCCpatch with the moments summed like the device's workgroup */
static float cc_patch_tree(const float *buf, int px, int py, const float *ri2w, const float *tmat, const float *reconW2I,
                           const float *vol, int vx, int vy, int vz, int level) {
  float M[16];
  matmul4f(tmat, ri2w, M);
  const int st = level + 1, nx = (px + st - 1) / st, ny = (py + st - 1) / st, total = nx * ny * 3;
  static double part[6][256];
  for (int k = 0; k < 6; ++k) for (int t = 0; t < 256; ++t) part[k][t] = 0;
  for (int t = 0; t < 256; ++t)
    for (int i = t; i < total; i += 256) {
      const int z = i % 3 - 1, r = i / 3;
      const int x = (r % nx) * st, y = (r / nx) * st;
      const float a = buf[y * px + x];
      float pos[3] = {(float)x, (float)y, (float)z}, w[3], vp[3];
      matvec3(M, pos, w);
      matvec3(reconW2I, w, vp);
      const float b = interp_sw(vp, vol, vx, vy, vz);
      if (a >= 0.0f && b >= 0.0f && a == a && b == b) {
        part[0][t] += 1.0; part[1][t] += (double)a; part[2][t] += (double)b;
        part[3][t] += (double)(a * a); part[4][t] += (double)(b * b); part[5][t] += (double)(a * b);
      }
    }
  double s[6];
  for (int k = 0; k < 6; ++k) {
    double wsum[4];
    for (int w = 0; w < 4; ++w) {
      double v[64];
      for (int l = 0; l < 64; ++l) v[l] = part[k][64 * w + l];
      for (int off = 32; off > 0; off >>= 1)
        for (int l = 0; l + off < 64; ++l) v[l] = v[l] + v[l + off];     /* lane l reads the old value of lane l + off: ascending l */
      wsum[w] = v[0];
    }
    s[k] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
  const float cnt = (float)s[0], x = (float)s[1], y = (float)s[2], x2 = (float)s[3], y2 = (float)s[4], xy = (float)s[5];
  return s[0] > 0 ? (xy - (x * y) / cnt) / (sqrtf(x2 - x * x / cnt) * sqrtf(y2 - y * y / cnt)) : 0.0f;
}

/* one parallelPatchRegOptimization run of one patch; returns the number of evaluations */
static int pvr_patch_opt(const float *buf, int px, int py, const float *ri2w, float *mat, const float *reconW2I, const float *vol,
                         int vx, int vy, int vz, int level, float step) {
  float params[6], dxt[6], cur[16];
  memcpy(cur, mat, sizeof(cur));
  pvr_m2p(cur, params);
  int n_eval = 0;
  for (int i = 0; i < 6; ++i) {
    const float pv = params[i];
    params[i] = pv + step; pvr_p2m(params, cur);
    const float s1 = cc_patch_tree(buf, px, py, ri2w, cur, reconW2I, vol, vx, vy, vz, level);
    params[i] = pv - step; pvr_p2m(params, cur);
    const float s2 = cc_patch_tree(buf, px, py, ri2w, cur, reconW2I, vol, vx, vy, vz, level);
    dxt[i] = s1 - s2;
    params[i] = pv; pvr_p2m(params, cur);
    n_eval += 2;
  }
  float norm = 0;
  for (int i = 0; i < 6; ++i) norm += dxt[i] * dxt[i];
  norm = sqrtf(norm);
  for (int i = 0; i < 6; ++i) dxt[i] = norm > 0.0f ? dxt[i] / norm : 0.0f;
  float similarity = cc_patch_tree(buf, px, py, ri2w, cur, reconW2I, vol, vx, vy, vz, level), new_similarity;
  int count = 0;
  n_eval += 1;
  do {
    new_similarity = similarity;
    for (int i = 0; i < 6; ++i) params[i] = params[i] + step * dxt[i];
    pvr_p2m(params, cur);
    similarity = cc_patch_tree(buf, px, py, ri2w, cur, reconW2I, vol, vx, vy, vz, level);
    count++;
    n_eval += 1;
  } while (similarity > new_similarity + 0.0001f && count < 50);
  for (int i = 0; i < 6; ++i) params[i] = params[i] - step * dxt[i];
  pvr_p2m(params, cur);
  memcpy(mat, cur, sizeof(cur));
  return n_eval;
}

static void invert4d(const double *a, double *out) {
  double w[4][8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) { w[i][j] = a[4 * i + j]; w[i][4 + j] = (i == j) ? 1.0 : 0.0; }
  for (int c = 0; c < 4; ++c) {
    int p = c;
    for (int r = c + 1; r < 4; ++r) if (fabs(w[r][c]) > fabs(w[p][c])) p = r;
    for (int j = 0; j < 8; ++j) { const double t = w[c][j]; w[c][j] = w[p][j]; w[p][j] = t; }
    const double d = w[c][c];
    for (int j = 0; j < 8; ++j) w[c][j] /= d;
    for (int r = 0; r < 4; ++r)
      if (r != c) { const double f = w[r][c]; for (int j = 0; j < 8; ++j) w[r][j] -= f * w[c][j]; }
  }
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[4 * i + j] = w[i][4 + j];
}

/* levels, steps, iterations: 3, 4, 20 in the reference (its run() also iterates an undefined fourth level, see the engine) */
void orc_pvr_register_patches(const float *patches, int px, int py, int n, const float *RI2W, const float *Mo, const float *InvMo,
                              float *T, float *Tinv_out, const float *reconW2I, const float *vol, int vx, int vy, int vz,
                              float recon_dim_x, int levels, int steps, int iterations, long long *counters3) {
  float *M = (float *)malloc(sizeof(float) * 16 * (size_t)n);
  float *buf = (float *)malloc(sizeof(float) * (size_t)n * px * py);
  long long evals = 0, launches = 0;
  for (int i = 0; i < n; ++i) matmul4f(T + 16 * (size_t)i, Mo + 16 * (size_t)i, M + 16 * (size_t)i);
  for (int level = levels - 1; level >= 0; --level) {
    const float sigma = (recon_dim_x / 2.0f) * (float)(1 << level);
    orc_pvr_blur_patches(patches, buf, px, py, n, sigma);
    float step = 2.0f * (float)(1 << level);
    for (int st = 0; st < steps; ++st) {
      for (int it = 0; it < iterations; ++it) {
        for (int i = 0; i < n; ++i)
          evals += pvr_patch_opt(buf + (size_t)i * px * py, px, py, RI2W + 16 * (size_t)i, M + 16 * (size_t)i, reconW2I, vol, vx, vy,
                                 vz, level, step);
        ++launches;
      }
      step /= 2.0f;
    }
  }
  for (int i = 0; i < n; ++i) {
    matmul4f(M + 16 * (size_t)i, InvMo + 16 * (size_t)i, T + 16 * (size_t)i);
    double a[16], inv[16];
    for (int k = 0; k < 16; ++k) a[k] = T[16 * (size_t)i + k];
    invert4d(a, inv);
    for (int k = 0; k < 16; ++k) Tinv_out[16 * (size_t)i + k] = (float)inv[k];
  }
  if (counters3) { counters3[0] = launches; counters3[1] = evals; counters3[2] = n; }
  free(M);
  free(buf);
}
