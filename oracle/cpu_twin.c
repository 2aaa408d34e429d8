/*
 * cpu_twin.c -- a plain-C restatement of the REFERENCE'S CPU RECONSTRUCTION PATH (the `--useCPU` branch of
 * reconstruction.cc:930-1140): irtkReconstruction::{CoeffInit, InitializeEMValues, GaussianReconstruction, SimulateSlices,
 * InitializeRobustStatistics, EStep, Scale, Superresolution + AdaptiveRegularization, MStep, MaskVolume} of
 * source/reconstructionGPU2/irtkReconstructionGPU.cc ("RG.cc"), each function citing the lines it follows.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Used only by bench.py's `cpu_baseline` leg (timing: "the reference irtkReconstruction CPU path
 * timed on the same box's host cores") and by tests/ (a quality cross-check of final volumes).  It PINS NOTHING:
 *   * the reference's own CPU path cannot be compiled in this image (IRTK needs GSL, boost and TBB headers the image lacks, RG.cc
 *     includes the CUDA class header) and building it against stand-ins for those is excluded by this repository's rules -- so this
 *     is a `port`, like oracle/svr_oracle.c, of a DIFFERENT algorithm of the reference;
 *   * that algorithm is not the GPU path's: a Gaussian PSF sampled on a res / quality_factor grid inside a 2 x voxel box and splatted
 *     trilinearly into an explicit coefficient list per slice pixel (`_volcoeffs`), all arithmetic in double (irtkRealPixel is double,
 *     IRTKSimple2/image++/include/irtkVoxel.h:9) -- against sinc^2 x Gauss taps on the 16^3 voxel lattice in float (README.md:117-119,
 *     142-144).  The two agree in image quality, not in bits.
 *
 * Differences from the reference that do not change what is computed: the three point transformations of RG.cc:2457-2459 /
 * 2506-2512 (slice.ImageToWorld, _transformations[i].Transform, _reconstructed.WorldToImage: three double 4 x 4 products) are composed
 * into one double matrix per slice from the float matrices the engine receives; TBB's parallel_for / parallel_reduce over slices are
 * pthreads over contiguous slice ranges (per-thread addon / confidence maps added in thread order, like parallel_reduce's join); the
 * coefficient lists are one flat array per slice instead of vector<vector<vector<POINT3D>>>.  Bias correction is off (the command
 * line's default, reconstruction.cc:121,202): b = 0 everywhere, exp(-b) = 1.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int32_t v; float value; } coeff;        /* POINT3D (RG.h): voxel (here its linear index) and weight */

typedef struct twin {
  int vx, vy, vz;
  double res;
  double *mask, *recon, *volw, *cmap;                  /* _mask, _reconstructed, _volume_weights, _confidence_map */
  int ns, sx, sy;
  double *slices, *weights, *sim, *simw;               /* [ns][sy][sx]: _slices (-1 = padding), _weights, _simulated_slices, _simulated_weights */
  unsigned char *siminside;                            /* _simulated_inside */
  int *sizes_x, *sizes_y;
  double *M;                                           /* [ns][16]: reconW2I * T * sliceI2W */
  double *dim;                                         /* [ns][3] slice voxel size */
  coeff **vc;                                          /* _volcoeffs: per slice, flat */
  uint64_t **vc_off;                                   /* per slice [sy * sx + 1] offsets into vc[s] */
  unsigned char *slice_inside;                         /* _slice_inside_cpu */
  double *scale, *slice_weight;                        /* _scale_cpu, _slice_weight_cpu */
  int *small, n_small;                                 /* _small_slices */
  int *excluded, n_excluded;                           /* _force_excluded */
  double quality_factor, step;
  double sigma, mix, m, mean_s, mean_s2, sigma_s, sigma_s2, mix_s;
  double delta, lambda, alpha, min_intensity, max_intensity, average_volume_weight;
  int adaptive, threads;
} twin;

static void mul4(const double *a, const double *b, double *c) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double t = 0;
      for (int k = 0; k < 4; ++k) t += a[4 * i + k] * b[4 * k + j];
      c[4 * i + j] = t;
    }
}

/* ---- a parallel_for over slices ------------------------------------------------------------------------------------------ */
typedef struct { twin *t; int lo, hi, id; void (*fn)(twin *, int, int, int, void *); void *arg; } job;
static void *run_job(void *p) { job *j = (job *)p; j->fn(j->t, j->lo, j->hi, j->id, j->arg); return NULL; }
/* contiguous slice ranges balanced by the pixels with data (TBB's ranges are work-stolen; the results do not depend on the cut) */
static void parallel_slices(twin *t, void (*fn)(twin *, int, int, int, void *), void *arg) {
  int nt = t->threads < 1 ? 1 : t->threads;
  if (nt > t->ns) nt = t->ns;
  double *cum = (double *)calloc((size_t)t->ns + 1, sizeof(double));
  for (int s = 0; s < t->ns; ++s) {
    long c = 0;
    const double *sl = t->slices + (size_t)s * t->sx * t->sy;
    for (int i = 0; i < t->sx * t->sy; ++i) c += sl[i] != -1;
    cum[s + 1] = cum[s] + (double)c + 1.0;
  }
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nt);
  job *jobs = (job *)malloc(sizeof(job) * nt);
  int at = 0;
  for (int k = 0; k < nt; ++k) {
    const int lo = at;
    if (k == nt - 1) at = t->ns;
    else { const double want = cum[t->ns] * (k + 1) / nt; while (at < t->ns && cum[at] < want) ++at; }
    jobs[k].t = t; jobs[k].lo = lo; jobs[k].hi = at; jobs[k].id = k; jobs[k].fn = fn; jobs[k].arg = arg;
    pthread_create(&th[k], NULL, run_job, &jobs[k]);
  }
  for (int k = 0; k < nt; ++k) pthread_join(th[k], NULL);
  free(th); free(jobs); free(cum);
}

twin *twin_create(int vx, int vy, int vz, double res, const float *mask, int ns, int sx, int sy, const float *slices, const int *sizes_x,
                  const int *sizes_y, const float *slice_i2w, const float *slice_t, const float *recon_w2i, const float *slice_dim,
                  double min_intensity, double max_intensity, int threads) {
  twin *t = (twin *)calloc(1, sizeof(twin));
  const size_t nv = (size_t)vx * vy * vz, np = (size_t)ns * sx * sy;
  t->vx = vx; t->vy = vy; t->vz = vz; t->res = res; t->ns = ns; t->sx = sx; t->sy = sy; t->threads = threads;
  t->mask = (double *)malloc(nv * sizeof(double)); t->recon = (double *)calloc(nv, sizeof(double));
  t->volw = (double *)calloc(nv, sizeof(double)); t->cmap = (double *)calloc(nv, sizeof(double));
  for (size_t i = 0; i < nv; ++i) t->mask[i] = mask[i];
  t->slices = (double *)malloc(np * sizeof(double)); t->weights = (double *)calloc(np, sizeof(double));
  t->sim = (double *)calloc(np, sizeof(double)); t->simw = (double *)calloc(np, sizeof(double)); t->siminside = (unsigned char *)calloc(np, 1);
  for (size_t i = 0; i < np; ++i) t->slices[i] = slices[i];
  t->sizes_x = (int *)malloc(ns * sizeof(int)); t->sizes_y = (int *)malloc(ns * sizeof(int));
  memcpy(t->sizes_x, sizes_x, ns * sizeof(int)); memcpy(t->sizes_y, sizes_y, ns * sizeof(int));
  t->M = (double *)malloc((size_t)ns * 16 * sizeof(double)); t->dim = (double *)malloc((size_t)ns * 3 * sizeof(double));
  double w2i[16];
  for (int k = 0; k < 16; ++k) w2i[k] = recon_w2i[k];
  for (int s = 0; s < ns; ++s) {
    double a[16], b[16], tmp[16];
    for (int k = 0; k < 16; ++k) { a[k] = slice_i2w[16 * (size_t)s + k]; b[k] = slice_t[16 * (size_t)s + k]; }
    mul4(b, a, tmp); mul4(w2i, tmp, t->M + 16 * (size_t)s);
    for (int k = 0; k < 3; ++k) t->dim[3 * (size_t)s + k] = slice_dim[3 * (size_t)s + k];
  }
  t->vc = (coeff **)calloc(ns, sizeof(coeff *)); t->vc_off = (uint64_t **)calloc(ns, sizeof(uint64_t *));
  t->slice_inside = (unsigned char *)calloc(ns, 1);
  t->scale = (double *)malloc(ns * sizeof(double)); t->slice_weight = (double *)malloc(ns * sizeof(double));
  t->small = (int *)malloc(ns * sizeof(int)); t->excluded = NULL;
  /* irtkReconstruction::irtkReconstruction RG.cc:159-221; SetSmoothingParameters RG.h:605-612 is called by the driver */
  t->quality_factor = 2; t->step = 0.0001; t->sigma_s = 0.025; t->mix_s = 0.9; t->mix = 0.9; t->delta = 1; t->lambda = 0.1;
  t->alpha = (0.05 / t->lambda) * t->delta * t->delta; t->adaptive = 0;
  t->min_intensity = min_intensity; t->max_intensity = max_intensity;     /* InitializeEM RG.cc:2887-2902: the range of the slices' intensities */
  return t;
}

void twin_destroy(twin *t) {
  if (!t) return;
  for (int s = 0; s < t->ns; ++s) { free(t->vc[s]); free(t->vc_off[s]); }
  free(t->vc); free(t->vc_off); free(t->mask); free(t->recon); free(t->volw); free(t->cmap); free(t->slices); free(t->weights); free(t->sim);
  free(t->simw); free(t->siminside); free(t->sizes_x); free(t->sizes_y); free(t->M); free(t->dim); free(t->slice_inside); free(t->scale);
  free(t->slice_weight); free(t->small); free(t->excluded); free(t);
}

void twin_set_smoothing_parameters(twin *t, double delta, double lambda) {          /* RG.h:605-612 */
  t->delta = delta; t->lambda = lambda * delta * delta; t->alpha = 0.05 / lambda;
  if (t->alpha > 1) t->alpha = 1;
}
void twin_set_force_excluded(twin *t, const int *idx, int n) {
  free(t->excluded);
  t->excluded = (int *)malloc((n > 0 ? n : 1) * sizeof(int)); memcpy(t->excluded, idx, n * sizeof(int)); t->n_excluded = n;
}

/* ---- ParallelCoeffInit::operator() RG.cc:2312-2605 ------------------------------------------------------------------------ */
static void coeff_init_range(twin *t, int lo, int hi, int id, void *arg) {
  (void)id; (void)arg;
  const int VX = t->vx, VY = t->vy, VZ = t->vz;
  const double res = t->res;                                                  /* :2323-2325 */
  for (int s = lo; s < hi; ++s) {
    const double *slice = t->slices + (size_t)s * t->sx * t->sy;
    const double *M = t->M + 16 * (size_t)s;
    const double dx = t->dim[3 * (size_t)s], dy = t->dim[3 * (size_t)s + 1], dz = t->dim[3 * (size_t)s + 2];   /* :2344-2345 */
    const double sigmax = 1.2 * dx / 2.3548, sigmay = 1.2 * dy / 2.3548, sigmaz = dz / 2.3548;                /* :2348-2350 */
    const double size = res / t->quality_factor;                                                               /* :2376 */
    const int xDim = (int)round(2 * dx / size), yDim = (int)round(2 * dy / size), zDim = (int)round(2 * dz / size);   /* :2381-2383 */
    /* the PSF image has default attributes (origin 0, identity axes): ImageToWorld(i) = (i - (n - 1) / 2) * size, and its centre maps to 0 (:2396-2400) */
    double *PSF = (double *)malloc((size_t)xDim * yDim * zDim * sizeof(double));
    double sum = 0;
    for (int i = 0; i < xDim; ++i)                                                                             /* :2405-2421 */
      for (int j = 0; j < yDim; ++j)
        for (int k = 0; k < zDim; ++k) {
          const double x = (i - 0.5 * (xDim - 1)) * size, y = (j - 0.5 * (yDim - 1)) * size, z = (k - 0.5 * (zDim - 1)) * size;
          const double v = exp(-x * x / (2 * sigmax * sigmax) - y * y / (2 * sigmay * sigmay) - z * z / (2 * sigmaz * sigmaz));
          PSF[((size_t)i * yDim + j) * zDim + k] = v;
          sum += v;
        }
    for (size_t q = 0; q < (size_t)xDim * yDim * zDim; ++q) PSF[q] /= sum;
    const int dim = (int)(floor(ceil(sqrt((double)(xDim * xDim + yDim * yDim + zDim * zDim)) * size / res) / 2)) * 2 + 1 + 2;   /* :2430-2431 */
    const int centre = (dim - 1) / 2;                                                                          /* :2442 */
    double *tPSF = (double *)malloc((size_t)dim * dim * dim * sizeof(double));
    const int gx = t->sizes_x[s], gy = t->sizes_y[s];
    uint64_t *off = (uint64_t *)calloc((size_t)t->sx * t->sy + 1, sizeof(uint64_t));
    size_t cap = 1 << 16, n = 0;
    coeff *out = (coeff *)malloc(cap * sizeof(coeff));
    int slice_inside = 0;
    /* the list order of the reference is pixel (i, j) with i = x OUTER (:2450-2451); the flat array here is indexed [j * sx + i], so the
     * lists are built in (j, i) order -- every list's own order (ii, jj, kk) is the reference's, which is what the sums depend on */
    for (int j = 0; j < t->sy; ++j)
      for (int i = 0; i < t->sx; ++i) {
        off[(size_t)j * t->sx + i] = n;
        if (i >= gx || j >= gy || slice[(size_t)j * t->sx + i] == -1) continue;                                /* :2452 */
        double x = M[0] * i + M[1] * j + M[3], y = M[4] * i + M[5] * j + M[7], z = M[8] * i + M[9] * j + M[11];   /* :2454-2459 (z = 0) */
        const int tx = (int)round(x), ty = (int)round(y), tz = (int)round(z);                                  /* :2460-2462 */
        memset(tPSF, 0, (size_t)dim * dim * dim * sizeof(double));                                             /* :2465-2468 */
        for (int ii = 0; ii < xDim; ++ii)                                                                      /* :2471-2580 */
          for (int jj = 0; jj < yDim; ++jj)
            for (int kk = 0; kk < zDim; ++kk) {
              /* PSF world coordinates around its centre, then slice image coordinates centred over the pixel (:2480-2503) */
              double px = (ii - 0.5 * (xDim - 1)) * size / dx + i, py = (jj - 0.5 * (yDim - 1)) * size / dy + j, pz = (kk - 0.5 * (zDim - 1)) * size / dz;
              x = M[0] * px + M[1] * py + M[2] * pz + M[3];                                                    /* :2506-2512 */
              y = M[4] * px + M[5] * py + M[6] * pz + M[7];
              z = M[8] * px + M[9] * py + M[10] * pz + M[11];
              const int nx = (int)floor(x), ny = (int)floor(y), nz = (int)floor(z);                            /* :2520-2522 */
              double wsum = 0;
              int inside = 0;
              for (int l = nx; l <= nx + 1; ++l)                                                               /* :2530-2542 */
                if (l >= 0 && l < VX)
                  for (int m = ny; m <= ny + 1; ++m)
                    if (m >= 0 && m < VY)
                      for (int q = nz; q <= nz + 1; ++q)
                        if (q >= 0 && q < VZ) {
                          wsum += (1 - fabs(l - x)) * (1 - fabs(m - y)) * (1 - fabs(q - z));
                          if (t->mask[(size_t)l + (size_t)m * VX + (size_t)q * VX * VY] == 1) { inside = 1; slice_inside = 1; }
                        }
              if (wsum <= 0 || !inside) continue;                                                              /* :2544-2545 */
              const double pv = PSF[((size_t)ii * yDim + jj) * zDim + kk];
              for (int l = nx; l <= nx + 1; ++l)                                                               /* :2547-2579 */
                if (l >= 0 && l < VX)
                  for (int m = ny; m <= ny + 1; ++m)
                    if (m >= 0 && m < VY)
                      for (int q = nz; q <= nz + 1; ++q)
                        if (q >= 0 && q < VZ) {
                          const double weight = (1 - fabs(l - x)) * (1 - fabs(m - y)) * (1 - fabs(q - z));
                          const int aa = l - tx + centre, bb = m - ty + centre, cc = q - tz + centre;
                          if (aa < 0 || aa >= dim || bb < 0 || bb >= dim || cc < 0 || cc >= dim) {
                            fprintf(stderr, "cpu_twin: error while trying to populate tPSF (%d %d %d)\n", aa, bb, cc);   /* the reference exits here (:2566-2575) */
                            exit(1);
                          }
                          tPSF[((size_t)aa * dim + bb) * dim + cc] += pv * weight / wsum;
                        }
            }
        for (int ii = 0; ii < dim; ++ii)                                                                       /* :2583-2592 */
          for (int jj = 0; jj < dim; ++jj)
            for (int kk = 0; kk < dim; ++kk) {
              const double v = tPSF[((size_t)ii * dim + jj) * dim + kk];
              if (v > 0) {
                if (n == cap) { cap *= 2; out = (coeff *)realloc(out, cap * sizeof(coeff)); }
                out[n].v = (int32_t)((ii + tx - centre) + (size_t)(jj + ty - centre) * VX + (size_t)(kk + tz - centre) * VX * VY);
                out[n].value = (float)v;
                ++n;
              }
            }
      }
    off[(size_t)t->sx * t->sy] = n;
    free(t->vc[s]); free(t->vc_off[s]);
    t->vc[s] = (coeff *)realloc(out, (n ? n : 1) * sizeof(coeff));
    t->vc_off[s] = off;
    t->slice_inside[s] = (unsigned char)slice_inside;                                                          /* :2601 */
    free(PSF); free(tPSF);
  }
}

/* irtkReconstruction::CoeffInit RG.cc:2617-2673 */
void twin_coeff_init(twin *t) {
  parallel_slices(t, coeff_init_range, NULL);
  const size_t nv = (size_t)t->vx * t->vy * t->vz;
  memset(t->volw, 0, nv * sizeof(double));                                                                     /* :2636-2637 */
  for (int s = 0; s < t->ns; ++s) {                                                                            /* :2641-2650 */
    const uint64_t n = t->vc_off[s][(size_t)t->sx * t->sy];
    for (uint64_t k = 0; k < n; ++k) t->volw[t->vc[s][k].v] += t->vc[s][k].value;
  }
  double sum = 0;
  long num = 0;
  for (size_t i = 0; i < nv; ++i) if (t->mask[i] == 1) { sum += t->volw[i]; ++num; }                           /* :2655-2667 */
  t->average_volume_weight = num ? sum / num : 0;
}
uint64_t twin_coefficients(const twin *t) {
  uint64_t n = 0;
  for (int s = 0; s < t->ns; ++s) if (t->vc_off[s]) n += t->vc_off[s][(size_t)t->sx * t->sy];
  return n;
}
/* slice pixels with s != -1 and a non-empty coefficient list: BASELINE.md section 3's `Vact` of the CPU path */
uint64_t twin_active_pixels(const twin *t) {
  uint64_t n = 0;
  for (int s = 0; s < t->ns; ++s)
    for (size_t p = 0; p < (size_t)t->sx * t->sy; ++p)
      n += t->slices[(size_t)s * t->sx * t->sy + p] != -1 && t->vc_off[s][p + 1] > t->vc_off[s][p];
  return n;
}

/* irtkReconstruction::InitializeEMValues RG.cc:2955-2985 */
void twin_initialize_em_values(twin *t) {
  const size_t np = (size_t)t->ns * t->sx * t->sy;
  for (size_t i = 0; i < np; ++i) t->weights[i] = t->slices[i] != -1 ? 1 : 0;
  for (int s = 0; s < t->ns; ++s) { t->slice_weight[s] = 1; t->scale[s] = 1; }
}

/* irtkReconstruction::GaussianReconstruction RG.cc:2765-2860 (serial in the reference too) */
static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }
void twin_gaussian_reconstruction(twin *t) {
  const size_t nv = (size_t)t->vx * t->vy * t->vz, n2 = (size_t)t->sx * t->sy;
  memset(t->recon, 0, nv * sizeof(double));                                                                    /* :2780 */
  int *voxel_num = (int *)malloc(t->ns * sizeof(int));
  for (int s = 0; s < t->ns; ++s) {
    const double scale = t->scale[s];
    int cnt = 0;
    for (size_t p = 0; p < n2; ++p) {
      double v = t->slices[(size_t)s * n2 + p];
      if (v == -1) continue;
      v *= scale;                                                                                              /* :2800 (exp(-b) = 1) */
      const uint64_t a = t->vc_off[s][p], b = t->vc_off[s][p + 1];
      if (b > a) ++cnt;                                                                                        /* :2813-2814 */
      for (uint64_t k = a; k < b; ++k) t->recon[t->vc[s][k].v] += t->vc[s][k].value * v;                      /* :2818-2821 */
    }
    voxel_num[s] = cnt;
  }
  for (size_t i = 0; i < nv; ++i) t->recon[i] = t->volw[i] != 0 ? t->recon[i] / t->volw[i] : 0;               /* :2833, irtkGenericImage.cc:801-826 */
  int *tmp = (int *)malloc(t->ns * sizeof(int));
  memcpy(tmp, voxel_num, t->ns * sizeof(int));
  qsort(tmp, t->ns, sizeof(int), cmp_int);                                                                     /* :2842-2848 */
  int mi = (int)round(t->ns * 0.5);
  if (mi >= t->ns) mi = t->ns - 1;
  const int median = tmp[mi];
  t->n_small = 0;
  for (int s = 0; s < t->ns; ++s) if (voxel_num[s] < 0.1 * median) t->small[t->n_small++] = s;                /* :2851-2854 */
  free(tmp); free(voxel_num);
}

/* ParallelSimulateSlices RG.cc:1090-1143 */
static void simulate_range(twin *t, int lo, int hi, int id, void *arg) {
  (void)id; (void)arg;
  const size_t n2 = (size_t)t->sx * t->sy;
  for (int s = lo; s < hi; ++s) {
    t->slice_inside[s] = 0;
    for (size_t p = 0; p < n2; ++p) {
      const size_t i = (size_t)s * n2 + p;
      t->sim[i] = 0; t->simw[i] = 0; t->siminside[i] = 0;
      if (t->slices[i] == -1) continue;
      double weight = 0, acc = 0;
      for (uint64_t k = t->vc_off[s][p]; k < t->vc_off[s][p + 1]; ++k) {
        const coeff c = t->vc[s][k];
        acc += c.value * t->recon[c.v];
        weight += c.value;
        if (t->mask[c.v] == 1) { t->siminside[i] = 1; t->slice_inside[s] = 1; }
      }
      if (weight > 0) { t->sim[i] = acc / weight; t->simw[i] = weight; }
    }
  }
}
void twin_simulate_slices(twin *t) { parallel_slices(t, simulate_range, NULL); }

/* irtkReconstruction::InitializeRobustStatistics RG.cc:3022-3074 */
void twin_initialize_robust_statistics(twin *t) {
  const size_t n2 = (size_t)t->sx * t->sy;
  double sigma = 0;
  long num = 0;
  for (int s = 0; s < t->ns; ++s) {
    for (size_t p = 0; p < n2; ++p) {
      const size_t i = (size_t)s * n2 + p;
      if (t->slices[i] == -1) continue;
      if (t->siminside[i] == 1 && t->simw[i] > 0.99) { const double e = t->slices[i] - t->sim[i]; sigma += e * e; ++num; }
    }
    if (!t->slice_inside[s]) t->slice_weight[s] = 0;
  }
  for (int k = 0; k < t->n_excluded; ++k) t->slice_weight[t->excluded[k]] = 0;
  t->sigma = sigma / num;
  t->sigma_s = 0.025; t->mix = 0.9; t->mix_s = 0.9;
  t->m = 1 / (2.1 * t->max_intensity - 1.9 * t->min_intensity);
}

static double G(const twin *t, double x, double s) { return t->step * exp(-x * x / (2 * s)) / (sqrt(6.28 * s)); }   /* RG.h:529-532 */
static double Mu(const twin *t, double m) { return m * t->step; }                                                  /* RG.h:534-537 */

/* ParallelEStep RG.cc:3076-3160 */
static void estep_range(twin *t, int lo, int hi, int id, void *arg) {
  (void)id;
  double *pot = (double *)arg;
  const size_t n2 = (size_t)t->sx * t->sy;
  for (int s = lo; s < hi; ++s) {
    const double scale = t->scale[s];
    double num = 0, sp = 0;
    for (size_t p = 0; p < n2; ++p) {
      const size_t i = (size_t)s * n2 + p;
      t->weights[i] = 0;                                                                                       /* :3088 */
      if (t->slices[i] == -1) continue;
      double v = t->slices[i] * scale;
      const int n = t->vc_off[s][p + 1] > t->vc_off[s][p];
      if (n && t->simw[i] > 0) {
        v -= t->sim[i];
        const double g = G(t, v, t->sigma), m = Mu(t, t->m);
        const double weight = g * t->mix / (g * t->mix + m * (1 - t->mix));
        t->weights[i] = weight;
        if (t->simw[i] > 0.99) { sp += (1.0 - weight) * (1.0 - weight); ++num; }
      }
    }
    pot[s] = num > 0 ? sqrt(sp / num) : -1;
  }
}
/* irtkReconstruction::EStep RG.cc:3442-3695 */
void twin_estep(twin *t) {
  const int ns = t->ns;
  double *pot = (double *)calloc(ns, sizeof(double));
  parallel_slices(t, estep_range, pot);
  for (int k = 0; k < t->n_excluded; ++k) pot[t->excluded[k]] = -1;
  for (int k = 0; k < t->n_small; ++k) pot[t->small[k]] = -1;
  for (int s = 0; s < ns; ++s) if (t->scale[s] < 0.2 || t->scale[s] > 5) pot[s] = -1;
  double sum = 0, den = 0, sum2 = 0, den2 = 0, maxs = 0, mins = 1;
  for (int s = 0; s < ns; ++s)
    if (pot[s] >= 0) {
      sum += pot[s] * t->slice_weight[s]; den += t->slice_weight[s];
      sum2 += pot[s] * (1 - t->slice_weight[s]); den2 += (1 - t->slice_weight[s]);
      if (pot[s] > maxs) maxs = pot[s];
      if (pot[s] < mins) mins = pot[s];
    }
  t->mean_s = den > 0 ? sum / den : mins;
  t->mean_s2 = den2 > 0 ? sum2 / den2 : (maxs + t->mean_s) / 2;
  sum = den = sum2 = den2 = 0;
  for (int s = 0; s < ns; ++s)
    if (pot[s] >= 0) {
      sum += (pot[s] - t->mean_s) * (pot[s] - t->mean_s) * t->slice_weight[s]; den += t->slice_weight[s];
      sum2 += (pot[s] - t->mean_s2) * (pot[s] - t->mean_s2) * (1 - t->slice_weight[s]); den2 += (1 - t->slice_weight[s]);
    }
  const double floor_ = t->step * t->step / 6.28;
  if (sum > 0 && den > 0) { t->sigma_s = sum / den; if (t->sigma_s < floor_) t->sigma_s = floor_; }
  else t->sigma_s = 0.025;
  if (sum2 > 0 && den2 > 0) { t->sigma_s2 = sum2 / den2; if (t->sigma_s2 < floor_) t->sigma_s2 = floor_; }
  else { t->sigma_s2 = (t->mean_s2 - t->mean_s) * (t->mean_s2 - t->mean_s) / 4; if (t->sigma_s2 < floor_) t->sigma_s2 = floor_; }
  for (int s = 0; s < ns; ++s) {
    if (pot[s] == -1) { t->slice_weight[s] = 0; continue; }
    if (den <= 0 || t->mean_s2 <= t->mean_s) { t->slice_weight[s] = 1; continue; }
    const double gs1 = pot[s] < t->mean_s2 ? G(t, pot[s] - t->mean_s, t->sigma_s) : 0;
    const double gs2 = pot[s] > t->mean_s ? G(t, pot[s] - t->mean_s2, t->sigma_s2) : 0;
    const double likelihood = gs1 * t->mix_s + gs2 * (1 - t->mix_s);
    if (likelihood > 0) t->slice_weight[s] = gs1 * t->mix_s / likelihood;
    else {
      if (pot[s] <= t->mean_s) t->slice_weight[s] = 1;
      if (pot[s] >= t->mean_s2) t->slice_weight[s] = 0;
      if (pot[s] < t->mean_s2 && pot[s] > t->mean_s) t->slice_weight[s] = 1;
    }
  }
  sum = 0;
  int num = 0;
  for (int s = 0; s < ns; ++s) if (pot[s] >= 0) { sum += t->slice_weight[s]; ++num; }
  t->mix_s = num > 0 ? sum / num : 0.9;
  free(pot);
}

/* ParallelScale RG.cc:3697-3749 */
static void scale_range(twin *t, int lo, int hi, int id, void *arg) {
  (void)id; (void)arg;
  const size_t n2 = (size_t)t->sx * t->sy;
  for (int s = lo; s < hi; ++s) {
    double num = 0, den = 0;
    for (size_t p = 0; p < n2; ++p) {
      const size_t i = (size_t)s * n2 + p;
      if (t->slices[i] == -1 || !(t->simw[i] > 0.99)) continue;
      num += t->weights[i] * t->slices[i] * t->sim[i];
      den += t->weights[i] * t->slices[i] * t->slices[i];
    }
    t->scale[s] = den > 0 ? num / den : 1;
  }
}
void twin_scale(twin *t) { parallel_slices(t, scale_range, NULL); }

/* ParallelSuperresolution RG.cc:3940-4022: parallel_reduce -- every thread its own addon / confidence map, joined in thread order */
typedef struct { double **addon, **cmap; } sr_arg;
static void sr_range(twin *t, int lo, int hi, int id, void *arg) {
  sr_arg *a = (sr_arg *)arg;
  const size_t nv = (size_t)t->vx * t->vy * t->vz, n2 = (size_t)t->sx * t->sy;
  double *addon = a->addon[id] = (double *)calloc(nv, sizeof(double));
  double *cmap = a->cmap[id] = (double *)calloc(nv, sizeof(double));
  for (int s = lo; s < hi; ++s) {
    const double scale = t->scale[s], sw = t->slice_weight[s];
    for (size_t p = 0; p < n2; ++p) {
      const size_t i = (size_t)s * n2 + p;
      if (t->slices[i] == -1) continue;
      double v = t->slices[i] * scale;
      v = t->sim[i] > 0 ? v - t->sim[i] : 0;                                                                   /* :3969-3972 */
      const double w = t->weights[i];
      for (uint64_t k = t->vc_off[s][p]; k < t->vc_off[s][p + 1]; ++k) {
        const coeff c = t->vc[s][k];
        addon[c.v] += c.value * v * w * sw;
        cmap[c.v] += c.value * w * sw;
      }
    }
  }
}
/* ParallelAdaptiveRegularization1 / 2 + AdaptiveRegularization RG.cc:4265-4428 (x planes over the threads) */
static const int DIRS[13][3] = {{1, 0, -1}, {0, 1, -1}, {1, 1, -1}, {1, -1, -1}, {1, 0, 0}, {0, 1, 0}, {1, 1, 0}, {1, -1, 0}, {1, 0, 1}, {0, 1, 1},
                                {1, 1, 1}, {1, -1, 1}, {0, 0, 1}};                                              /* RG.cc:188-204 */
typedef struct { twin *t; const double *original, *original2; double *b; int lo, hi, pass; } reg_job;
static void *reg_run(void *p) {
  reg_job *j = (reg_job *)p;
  twin *t = j->t;
  const int dx = t->vx, dy = t->vy, dz = t->vz;
  const size_t nv = (size_t)dx * dy * dz;
#define AT(x, y, z) ((size_t)(x) + (size_t)(y) * dx + (size_t)(z) * dx * dy)
  if (j->pass == 1) {
    for (int i = j->lo; i < j->hi; ++i) {                                                                      /* :4281-4306: direction i */
      double f = 0;
      for (int k = 0; k < 3; ++k) f += fabs((double)DIRS[i][k]);
      f = 1 / f;
      for (int x = 0; x < dx; ++x)
        for (int y = 0; y < dy; ++y)
          for (int z = 0; z < dz; ++z) {
            const int xx = x + DIRS[i][0], yy = y + DIRS[i][1], zz = z + DIRS[i][2];
            double v = 0;
            if (xx >= 0 && xx < dx && yy >= 0 && yy < dy && zz >= 0 && zz < dz && t->cmap[AT(x, y, z)] > 0 && t->cmap[AT(xx, yy, zz)] > 0) {
              const double diff = (j->original[AT(xx, yy, zz)] - j->original[AT(x, y, z)]) * sqrt(f) / t->delta;
              v = f / sqrt(1 + diff * diff);
            }
            j->b[(size_t)i * nv + AT(x, y, z)] = v;
          }
    }
    return NULL;
  }
  const double kk = t->alpha * t->lambda / (t->delta * t->delta);
  for (int x = j->lo; x < j->hi; ++x)                                                                          /* :4339-4388 */
    for (int y = 0; y < dy; ++y)
      for (int z = 0; z < dz; ++z) {
        double val = 0, valW = 0, sum = 0;
        for (int i = 0; i < 13; ++i) {
          const int xx = x + DIRS[i][0], yy = y + DIRS[i][1], zz = z + DIRS[i][2];
          if (xx >= 0 && xx < dx && yy >= 0 && yy < dy && zz >= 0 && zz < dz) {
            const double bb = j->b[(size_t)i * nv + AT(x, y, z)];
            val += bb * j->original2[AT(xx, yy, zz)] * t->cmap[AT(xx, yy, zz)];
            valW += bb * t->cmap[AT(xx, yy, zz)];
            sum += bb;
          }
        }
        for (int i = 0; i < 13; ++i) {
          const int xx = x - DIRS[i][0], yy = y - DIRS[i][1], zz = z - DIRS[i][2];
          if (xx >= 0 && xx < dx && yy >= 0 && yy < dy && zz >= 0 && zz < dz) {
            const double bb = j->b[(size_t)i * nv + AT(xx, yy, zz)];
            val += bb * j->original2[AT(xx, yy, zz)] * t->cmap[AT(xx, yy, zz)];
            valW += bb * t->cmap[AT(xx, yy, zz)];
            sum += bb;
          }
        }
        val -= sum * j->original2[AT(x, y, z)] * t->cmap[AT(x, y, z)];
        valW -= sum * t->cmap[AT(x, y, z)];
        val = j->original2[AT(x, y, z)] * t->cmap[AT(x, y, z)] + kk * val;
        valW = t->cmap[AT(x, y, z)] + kk * valW;
        t->recon[AT(x, y, z)] = valW > 0 ? val / valW : 0;
      }
#undef AT
  return NULL;
}
static void reg_pass(twin *t, int pass, int n, const double *original, const double *original2, double *b) {
  int nt = t->threads < 1 ? 1 : t->threads;
  if (nt > n) nt = n;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nt);
  reg_job *jobs = (reg_job *)malloc(sizeof(reg_job) * nt);
  for (int k = 0; k < nt; ++k) {
    jobs[k].t = t; jobs[k].original = original; jobs[k].original2 = original2; jobs[k].b = b; jobs[k].pass = pass;
    jobs[k].lo = (int)((long)n * k / nt); jobs[k].hi = (int)((long)n * (k + 1) / nt);
    pthread_create(&th[k], NULL, reg_run, &jobs[k]);
  }
  for (int k = 0; k < nt; ++k) pthread_join(th[k], NULL);
  free(th); free(jobs);
}

/* irtkReconstruction::Superresolution RG.cc:4055-4119 */
void twin_superresolution(twin *t, int iter) {
  (void)iter;
  const size_t nv = (size_t)t->vx * t->vy * t->vz;
  double *original = (double *)malloc(nv * sizeof(double));
  memcpy(original, t->recon, nv * sizeof(double));                                                             /* :4064 */
  int nt = t->threads < 1 ? 1 : t->threads;
  if (nt > t->ns) nt = t->ns;
  sr_arg a;
  a.addon = (double **)calloc(nt, sizeof(double *)); a.cmap = (double **)calloc(nt, sizeof(double *));
  parallel_slices(t, sr_range, &a);
  double *addon = a.addon[0];
  memcpy(t->cmap, a.cmap[0], nv * sizeof(double));
  for (int k = 1; k < nt; ++k) {                                                                               /* join, :3992-3995 */
    if (!a.addon[k]) continue;
    for (size_t i = 0; i < nv; ++i) { addon[i] += a.addon[k][i]; t->cmap[i] += a.cmap[k][i]; }
    free(a.addon[k]); free(a.cmap[k]);
  }
  free(a.cmap[0]);
  if (!t->adaptive)                                                                                            /* :4080-4091 */
    for (size_t i = 0; i < nv; ++i)
      if (t->cmap[i] > 0) { addon[i] /= t->cmap[i]; t->cmap[i] = 1; }
  for (size_t i = 0; i < nv; ++i) {                                                                            /* :4093-4103 */
    t->recon[i] += addon[i] * t->alpha;
    if (t->recon[i] < t->min_intensity * 0.9) t->recon[i] = t->min_intensity * 0.9;
    if (t->recon[i] > t->max_intensity * 1.1) t->recon[i] = t->max_intensity * 1.1;
  }
  free(addon); free(a.addon); free(a.cmap);
  /* AdaptiveRegularization RG.cc:4394-4428 */
  double *b = (double *)malloc(13 * nv * sizeof(double));
  reg_pass(t, 1, 13, original, NULL, b);
  double *original2 = (double *)malloc(nv * sizeof(double));
  memcpy(original2, t->recon, nv * sizeof(double));
  reg_pass(t, 2, t->vx, NULL, original2, b);
  free(b); free(original); free(original2);
}

/* ParallelMStep + irtkReconstruction::MStep RG.cc:4121-4263: parallel_reduce; the root body starts at (max, min) = (FLT limits), split
 * bodies at 0 -- with one body (the serial fallback) the limits; here every thread starts like the root and the joins take min / max */
typedef struct { double *sigma, *mix, *num, *min, *max; } ms_arg;
static void mstep_range(twin *t, int lo, int hi, int id, void *arg) {
  ms_arg *a = (ms_arg *)arg;
  const size_t n2 = (size_t)t->sx * t->sy;
  double sigma = 0, mix = 0, num = 0, mn = 1.7976931348623157e308, mx = 2.2250738585072014e-308;               /* voxel_limits<double> */
  for (int s = lo; s < hi; ++s) {
    const double scale = t->scale[s];
    for (size_t p = 0; p < n2; ++p) {
      const size_t i = (size_t)s * n2 + p;
      if (t->slices[i] == -1 || !(t->simw[i] > 0.99)) continue;
      const double e = t->slices[i] * scale - t->sim[i];
      sigma += e * e * t->weights[i];
      mix += t->weights[i];
      if (e < mn) mn = e;
      if (e > mx) mx = e;
      ++num;
    }
  }
  a->sigma[id] = sigma; a->mix[id] = mix; a->num[id] = num; a->min[id] = mn; a->max[id] = mx;
}
int twin_mstep(twin *t, int iter) {
  int nt = t->threads < 1 ? 1 : t->threads;
  if (nt > t->ns) nt = t->ns;
  double *buf = (double *)calloc((size_t)5 * nt, sizeof(double));
  ms_arg a = {buf, buf + nt, buf + 2 * nt, buf + 3 * nt, buf + 4 * nt};
  for (int k = 0; k < nt; ++k) { a.min[k] = 1.7976931348623157e308; a.max[k] = 2.2250738585072014e-308; }
  parallel_slices(t, mstep_range, &a);
  double sigma = 0, mix = 0, num = 0, mn = a.min[0], mx = a.max[0];
  for (int k = 0; k < nt; ++k) { sigma += a.sigma[k]; mix += a.mix[k]; num += a.num[k]; if (a.min[k] < mn) mn = a.min[k]; if (a.max[k] > mx) mx = a.max[k]; }
  free(buf);
  if (!(mix > 0)) return 1;                                                                                    /* the reference exits (:4246-4249) */
  t->sigma = sigma / mix;
  if (t->sigma < t->step * t->step / 6.28) t->sigma = t->step * t->step / 6.28;
  if (iter > 1) t->mix = mix / num;
  t->m = 1 / (mx - mn);
  return 0;
}

/* irtkReconstruction::MaskVolume RG.cc:5325-5335 */
void twin_mask_volume(twin *t) {
  const size_t nv = (size_t)t->vx * t->vy * t->vz;
  for (size_t i = 0; i < nv; ++i) if (t->mask[i] == 0) t->recon[i] = -1;
}

/* irtkReconstruction::RestoreSliceIntensities RG.cc:1003-1024 (stack_index[ns], factors per stack) + ScaleVolume RG.cc:1034-1079 */
void twin_restore_and_scale_volume(twin *t, const float *stack_factor, const int *stack_index) {
  const size_t n2 = (size_t)t->sx * t->sy, nv = (size_t)t->vx * t->vy * t->vz;
  for (int s = 0; s < t->ns; ++s) {
    const double f = stack_factor[stack_index[s]];
    for (size_t p = 0; p < n2; ++p) if (t->slices[(size_t)s * n2 + p] > 0) t->slices[(size_t)s * n2 + p] /= f;
  }
  double num = 0, den = 0;
  for (int s = 0; s < t->ns; ++s)
    for (size_t p = 0; p < n2; ++p) {
      const size_t i = (size_t)s * n2 + p;
      if (t->slices[i] == -1 || !(t->simw[i] > 0.99)) continue;
      num += t->weights[i] * t->slice_weight[s] * t->slices[i] * t->sim[i];
      den += t->weights[i] * t->slice_weight[s] * t->sim[i] * t->sim[i];
    }
  const double scale = num / den;
  for (size_t i = 0; i < nv; ++i) if (t->recon[i] > 0) t->recon[i] *= scale;
}

void twin_get_volume(const twin *t, float *out) {
  const size_t nv = (size_t)t->vx * t->vy * t->vz;
  for (size_t i = 0; i < nv; ++i) out[i] = (float)t->recon[i];
}
void twin_get_state(const twin *t, double *scale, double *slice_weight, double scalars8[8]) {
  if (scale) memcpy(scale, t->scale, t->ns * sizeof(double));
  if (slice_weight) memcpy(slice_weight, t->slice_weight, t->ns * sizeof(double));
  if (scalars8) { scalars8[0] = t->sigma; scalars8[1] = t->mix; scalars8[2] = t->m; scalars8[3] = t->mean_s; scalars8[4] = t->mean_s2;
                  scalars8[5] = t->sigma_s; scalars8[6] = t->sigma_s2; scalars8[7] = t->mix_s; }
}
