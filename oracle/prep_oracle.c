/*
 * prep_oracle.c -- CPU restatement of the host functions that feed the hot path:
 *
 *   orc_match_stack_intensities   irtkReconstruction::MatchStackIntensitiesWithMasking
 *                                 (source/reconstructionGPU2/irtkReconstructionGPU.cc:1375-1493)
 *   orc_generate_2d_patches       PatchBasedObject<T>::generate2DPatches
 *                                 (source/reconstructionGPU2/include/patchBasedObject.cuh:174-342)
 *   orc_segment_slic              runStackSLIC<T>::segmentSLIC: the SLICO superpixel labels of a stack, slice by slice
 *                                 (source/reconstructionGPU2/runStackSLIC.cpp:55-151, 291-537, 665-840)
 *   orc_generate_2d_superpixel_patches   PatchBasedObject<T>::generate2DSuperpixelPatches + dilatePatch
 *                                 (patchBasedObject.cuh:347-367, 433-802)
 *
 * TEST INFRASTRUCTURE ONLY (like svr_oracle.c): nothing in fetalreconstruction_amd/ or include/ may include, link or call
 * this file.  It exists so that the C++ command lines (csrc/svr_prep.h, csrc/pvr_cli.cpp) and their Python mirrors are checked
 * against a third, independent statement of the reference's loops instead of only against each other.
 *
 * PARITY UNPINNED, as for the compute path: the reference ships no fixtures for these functions and IRTK does not build here.
 * The image geometry follows IRTKSimple2/image++/src/irtkBaseImage.cc:79-147 (GetImageToWorldMatrix / GetWorldToImageMatrix,
 * matrix products in the reference's order), GetRegion follows irtkGenericImage.cc:570-640, round() irtkCommon.h:85-88.
 * Everything is double like the reference's host code; the loops keep the reference's nesting (x outermost in the intensity
 * matching, so the double sums are added in its order).
 *
 * One deliberate deviation, switchable: generate2DPatches truncates `slice(xx, yy, 0)` coordinates that are integers in exact
 * arithmetic but come out of two double matrix products as 19.999999999999996 or 20.000000000000004 (patchBasedObject.cuh:
 * 262-277); which pixel a patch column reads is then rounding noise.  snap = 1 treats a coordinate within 1e-6 of an integer
 * as that integer (what both command lines do); snap = 0 is the literal truncation.
 */
#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int nx, ny, nz;
  double dx, dy, dz;
  double xaxis[3], yaxis[3], zaxis[3], origin[3];
} orc_attr;                      /* = svr_image_attr of include/svr_host.h, field for field */

/* ---- 4x4 helpers (row-major) ------------------------------------------------------------------------------------------ */
static void mm(const double *A, const double *B, double *C) {
  double t[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j];   /* irtkMatrix::operator*: k innermost */
      t[4 * i + j] = s;
    }
  memcpy(C, t, sizeof(t));
}
static void ident(double *M) { memset(M, 0, 16 * sizeof(double)); M[0] = M[5] = M[10] = M[15] = 1.0; }

/* irtkBaseImage::GetImageToWorldMatrix, irtkBaseImage.cc:79-113: translate2 * (rot * (scale * translate1)) */
static void image_to_world(const orc_attr *a, double *M) {
  double t1[16], sc[16], rot[16], t2[16], tmp[16];
  ident(t1);
  t1[3] = -(a->nx - 1) / 2.0; t1[7] = -(a->ny - 1) / 2.0; t1[11] = -(a->nz - 1) / 2.0;
  memset(sc, 0, sizeof(sc));
  sc[0] = a->dx; sc[5] = a->dy; sc[10] = a->dz; sc[15] = 1.0;
  memset(rot, 0, sizeof(rot));
  for (int i = 0; i < 3; ++i) { rot[4 * i + 0] = a->xaxis[i]; rot[4 * i + 1] = a->yaxis[i]; rot[4 * i + 2] = a->zaxis[i]; }
  rot[15] = 1.0;
  ident(t2);
  t2[3] = a->origin[0]; t2[7] = a->origin[1]; t2[11] = a->origin[2];
  mm(sc, t1, tmp);
  mm(rot, tmp, tmp);
  mm(t2, tmp, M);
}
/* irtkBaseImage::GetWorldToImageMatrix, irtkBaseImage.cc:115-147: translate2 * (scale * (rot * translate1)) */
static void world_to_image(const orc_attr *a, double *M) {
  double t1[16], rot[16], sc[16], t2[16], tmp[16];
  ident(t1);
  t1[3] = -a->origin[0]; t1[7] = -a->origin[1]; t1[11] = -a->origin[2];
  memset(rot, 0, sizeof(rot));
  for (int j = 0; j < 3; ++j) { rot[0 + j] = a->xaxis[j]; rot[4 + j] = a->yaxis[j]; rot[8 + j] = a->zaxis[j]; }
  rot[15] = 1.0;
  memset(sc, 0, sizeof(sc));
  sc[0] = 1.0 / a->dx; sc[5] = 1.0 / a->dy; sc[10] = 1.0 / a->dz; sc[15] = 1.0;
  ident(t2);
  t2[3] = (a->nx - 1) / 2.0; t2[7] = (a->ny - 1) / 2.0; t2[11] = (a->nz - 1) / 2.0;
  mm(rot, t1, tmp);
  mm(sc, tmp, tmp);
  mm(t2, tmp, M);
}
static void apply(const double *M, double *x, double *y, double *z) {
  const double a = M[0] * *x + M[1] * *y + M[2] * *z + M[3];
  const double b = M[4] * *x + M[5] * *y + M[6] * *z + M[7];
  const double c = M[8] * *x + M[9] * *y + M[10] * *z + M[11];
  *x = a; *y = b; *z = c;
}
static double irtk_round(double x) { return x > 0 ? (double)(int)(x + 0.5) : (double)(int)(x - 0.5); }   /* irtkCommon.h:85-88 */

/* ---- MatchStackIntensitiesWithMasking, RG.cc:1375-1493 ---------------------------------------------------------------- */
/* stacks[s]: double [nz][ny][nx] of attrs[s], rescaled IN PLACE (voxels > 0 only); transforms: row-major 4x4 per stack
 * (irtkRigidTransformation::Transform = the matrix applied to the world position); mask on mask_attr, compared with == 1;
 * factors_out[s] = (float)factor as _stack_factor stores it.  Returns 0, or 1 + s if stack s has no overlap with the ROI. */
int orc_match_stack_intensities(int n_stacks, const orc_attr *attrs, double *const *stacks, const double *transforms,
                                const orc_attr *mask_attr, const double *mask, double average_value, int together,
                                float *factors_out, double *averages_out) {
  double m_w2i[16];
  world_to_image(mask_attr, m_w2i);
  double *avg = (double *)malloc(sizeof(double) * (size_t)n_stacks);
  for (int ind = 0; ind < n_stacks; ++ind) {
    const orc_attr *a = &attrs[ind];
    double i2w[16];
    image_to_world(a, i2w);
    double sum = 0, num = 0;
    for (int i = 0; i < a->nx; i++)                          /* the reference's nesting: x, y, z (RG.cc:1398-1400) */
      for (int j = 0; j < a->ny; j++)
        for (int k = 0; k < a->nz; k++) {
          double x = i, y = j, z = k;
          apply(i2w, &x, &y, &z);                            /* stacks[ind].ImageToWorld */
          apply(transforms + 16 * (size_t)ind, &x, &y, &z);  /* stack_transformations[ind].Transform */
          apply(m_w2i, &x, &y, &z);                          /* _mask.WorldToImage */
          x = irtk_round(x); y = irtk_round(y); z = irtk_round(z);
          if ((x >= 0) && (x < mask_attr->nx) && (y >= 0) && (y < mask_attr->ny) && (z >= 0) && (z < mask_attr->nz)) {
            if (mask[((size_t)z * mask_attr->ny + (size_t)y) * mask_attr->nx + (size_t)x] == 1) {
              sum += stacks[ind][((size_t)k * a->ny + j) * a->nx + i];
              num++;
            }
          }
        }
    if (num > 0) avg[ind] = sum / num;
    else { free(avg); return 1 + ind; }                      /* "Stack .. has no overlap with ROI", exit(1) */
  }
  double global_average = 0;
  if (together) {
    for (int i = 0; i < n_stacks; ++i) global_average += avg[i];
    global_average /= n_stacks;
  }
  for (int ind = 0; ind < n_stacks; ++ind) {
    const double factor = together ? average_value / global_average : average_value / avg[ind];
    factors_out[ind] = (float)factor;
    if (averages_out) averages_out[ind] = avg[ind];
    const size_t n = (size_t)attrs[ind].nx * attrs[ind].ny * attrs[ind].nz;
    for (size_t i = 0; i < n; ++i)
      if (stacks[ind][i] > 0) stacks[ind][i] *= factor;
  }
  free(avg);
  return 0;
}

/* ---- generate2DPatches, patchBasedObject.cuh:174-342 (T = float) ------------------------------------------------------- */
/* stack: float [nz][ny][nx] of `attr`; thickness = m_thickness (the slice's z size becomes 2 * thickness, :201); mask on
 * mask_attr (> 0 = inside).  pbb / stride in pixels; full_slices: patch = slice, stride = size + 1 (:183-189).
 * Outputs for up to `cap` patches: data [n][py][px] (zero-initialised images, :227), i2w / w2i [n][16] (float, toMatrix4),
 * origins [n][3].  *n_out = patches kept (setCount > pbb.x * pbb.y / 3, :318), *total_pixels += setCount of the kept ones.
 * Returns 0, or 1 when more than `cap` patches would be kept. */
int orc_generate_2d_patches(const orc_attr *attr, const float *stack, double thickness, const orc_attr *mask_attr, const float *mask,
                            int pbbx, int pbby, int stridex, int stridey, int full_slices, int snap, int cap, float *data, float *i2w,
                            float *w2i, double *origins, int *n_out, long *total_pixels) {
  if (full_slices) { pbbx = attr->nx; pbby = attr->ny; stridex = attr->nx + 1; stridey = attr->ny + 1; }
  double s_i2w[16], m_w2i[16];
  image_to_world(attr, s_i2w);
  world_to_image(mask_attr, m_w2i);
  int n = 0;
  for (int z = 0; z < attr->nz; z++) {
    /* slice = GetRegion(0, 0, z, nx, ny, z + 1): origin shifted so that its voxel (0,0,0) is the stack's (0,0,z)
     * (irtkGenericImage.cc:570-640), then PutPixelSize(dx, dy, 2 * thickness) */
    orc_attr sa = *attr;
    sa.nz = 1;
    sa.origin[0] = sa.origin[1] = sa.origin[2] = 0;
    {
      double x1 = 0, y1 = 0, z1 = z, x2 = 0, y2 = 0, z2 = 0, t[16];
      apply(s_i2w, &x1, &y1, &z1);
      image_to_world(&sa, t);
      apply(t, &x2, &y2, &z2);
      sa.origin[0] = x1 - x2; sa.origin[1] = y1 - y2; sa.origin[2] = z1 - z2;
    }
    sa.dz = thickness * 2;
    double sl_i2w[16], sl_w2i[16];
    image_to_world(&sa, sl_i2w);
    world_to_image(&sa, sl_w2i);
    const float *slice = stack + (size_t)z * attr->ny * attr->nx;
    for (int y = 0; y < attr->ny + pbby; y += stridey)
      for (int x = 0; x < attr->nx + pbbx; x += stridex) {
        orc_attr pa = sa;
        pa.nx = pbbx; pa.ny = pbby;
        pa.origin[0] = pa.origin[1] = pa.origin[2] = 0;
        double x1 = x, y1 = y, z1 = 0, x2 = 0, y2 = 0, z2 = 0, t[16];
        apply(sl_i2w, &x1, &y1, &z1);
        image_to_world(&pa, t);
        apply(t, &x2, &y2, &z2);
        pa.origin[0] = x1 - x2; pa.origin[1] = y1 - y2; pa.origin[2] = z1 - z2;
        double p_i2w[16], p_w2i[16];
        image_to_world(&pa, p_i2w);
        world_to_image(&pa, p_w2i);
        if (n >= cap) return 1;
        float *patch = data + (size_t)n * pbbx * pbby;
        memset(patch, 0, sizeof(float) * (size_t)pbbx * pbby);
        int setCount = 0;
        for (int j = 0; j < pbby; j++)
          for (int i = 0; i < pbbx; i++) {
            double xx = i, yy = j, zz = 0;
            apply(p_i2w, &xx, &yy, &zz);
            double xx1 = xx, yy1 = yy, zz1 = zz;
            apply(sl_w2i, &xx, &yy, &zz);
            apply(m_w2i, &xx1, &yy1, &zz1);
            if (snap) {
              if (fabs(xx - irtk_round(xx)) < 1e-6) xx = irtk_round(xx);
              if (fabs(yy - irtk_round(yy)) < 1e-6) yy = irtk_round(yy);
              if (fabs(xx1 - irtk_round(xx1)) < 1e-6) xx1 = irtk_round(xx1);
              if (fabs(yy1 - irtk_round(yy1)) < 1e-6) yy1 = irtk_round(yy1);
              if (fabs(zz1 - irtk_round(zz1)) < 1e-6) zz1 = irtk_round(zz1);
            }
            if (xx >= 0 && yy >= 0 && xx < sa.nx && yy < sa.ny) {
              if (xx1 >= 0 && yy1 >= 0 && zz1 >= 0 && xx1 < mask_attr->nx && yy1 < mask_attr->ny && zz1 < mask_attr->nz) {
                /* m_mask.Get(xx1, yy1, zz1), slice(xx, yy, 0): double -> int conversions truncate */
                if (mask[((size_t)(int)zz1 * mask_attr->ny + (size_t)(int)yy1) * mask_attr->nx + (size_t)(int)xx1] > 0) {
                  const float v = slice[(size_t)(int)yy * attr->nx + (size_t)(int)xx];
                  patch[(size_t)j * pbbx + i] = v;
                  if (v != 0 && v != -1) setCount++;
                }
              }
            }
          }
        if (setCount > 1.0f / 3.0f * pbby * pbbx) {           /* :318 (float arithmetic on the right-hand side) */
          for (int q = 0; q < 16; ++q) { i2w[16 * (size_t)n + q] = (float)p_i2w[q]; w2i[16 * (size_t)n + q] = (float)p_w2i[q]; }
          if (origins) { origins[3 * (size_t)n] = pa.origin[0]; origins[3 * (size_t)n + 1] = pa.origin[1]; origins[3 * (size_t)n + 2] = pa.origin[2]; }
          *total_pixels += setCount;
          ++n;
        }
      }
  }
  *n_out = n;
  return 0;
}

/* =======================================================================================================================
 * Superpixel patches (the second generator):
 *   orc_segment_slic                      runStackSLIC<T>::segmentSLIC with rgbtolab, getLABXYSeeds, PerformSuperpixelSLICO,
 *                                         EnforceSuperpixelConnectivity   (source/reconstructionGPU2/runStackSLIC.cpp:55-151,
 *                                         291-537, 665-840; SLICO is Achanta et al.'s zero-parameter SLIC)
 *   orc_generate_2d_superpixel_patches    PatchBasedObject<T>::generate2DSuperpixelPatches + dilatePatch
 *                                         (source/reconstructionGPU2/include/patchBasedObject.cuh:347-367, 433-802)
 * T = float (the command line instantiates PatchBasedObject<float>).  What the reference leaves undefined is pinned the way
 * both products pin it: klabels of pixels no seed window reaches are -1 and take no part in the cluster statistics
 * (uninitialised memory there, :305, :372-373); the connectivity pass keeps its work lists as large as the image (10 * SUPSZ
 * entries there, written past for larger segments, :461-462); a stack whose minimum equals its maximum gets grey value 0
 * (0/0 there, :738-740).
 * ======================================================================================================================= */
/* rgbtolab (runSLIC_2D.c:55-110) for r = g = b: the input is an 8-bit grey level, so the conversion is a table of 256 entries.
 * Written in round 3 from the colour-space definitions the reference's lines implement, independently of the product's
 * csrc/svr_slic.h: sRGB companding (IEC 61966-2-1: linear below 0.04045, else ((C + 0.055) / 1.055)^2.4), the D65 sRGB -> XYZ
 * matrix rows as the reference types them (each row applied to the same linear grey value, summed left to right), the white
 * point (0.950456, 1, 1.088754), and CIE L*a*b* with epsilon = 0.008856, kappa = 903.3. */
static void slic_lab(const int *grey, int sz, double *lv, double *av, double *bv) {
  static const double M[3][3] = {{0.4124564, 0.3575761, 0.1804375}, {0.2126729, 0.7151522, 0.0721750}, {0.0193339, 0.1191920, 0.9503041}};
  static const double white[3] = {0.950456, 1.0, 1.088754};
  double tab[256][3];
  for (int g = 0; g < 256; ++g) {
    const double srgb = g / 255.0;
    double lin;
    if (srgb <= 0.04045) lin = srgb / 12.92;
    else lin = pow((srgb + 0.055) / 1.055, 2.4);
    double f[3];
    for (int c = 0; c < 3; ++c) {
      const double xyz = lin * M[c][0] + lin * M[c][1] + lin * M[c][2];
      const double rel = xyz / white[c];
      f[c] = rel > 0.008856 ? pow(rel, 1.0 / 3.0) : (903.3 * rel + 16.0) / 116.0;
    }
    tab[g][0] = 116.0 * f[1] - 16.0;
    tab[g][1] = 500.0 * (f[0] - f[1]);
    tab[g][2] = 200.0 * (f[1] - f[2]);
  }
  for (int i = 0; i < sz; ++i) {
    const int g = grey[i] < 0 ? 0 : (grey[i] > 255 ? 255 : grey[i]);
    lv[i] = tab[g][0]; av[i] = tab[g][1]; bv[i] = tab[g][2];
  }
}

/* getLABXYSeeds (runSLIC_2D.c:111-151), square grid: along each axis the strips are STEP wide, their number is the extent over
 * STEP rounded to nearest (one fewer if that overshoots), and the pixels left over are spread over the strips -- strip i starts
 * i * STEP + STEP / 2 + trunc(i * leftover / strips).  The two axes are independent: positions per axis first, then the grid. */
static int slic_axis_seeds(int STEP, int extent, int *pos) {
  int strips = (int)(0.5 + (double)extent / (double)STEP);
  if (extent - STEP * strips < 0) strips--;
  const int left = extent - STEP * strips;
  const double per = (double)left / (double)strips;
  for (int i = 0; i < strips; ++i) pos[i] = i * STEP + STEP / 2 + (int)(i * per);
  return strips;
}
static int slic_grid_seeds(int STEP, int width, int height, int *seed) {
  int *sx = (int *)malloc((size_t)(width + 2) * sizeof(int)), *sy = (int *)malloc((size_t)(height + 2) * sizeof(int));
  const int nx = slic_axis_seeds(STEP, width, sx), ny = slic_axis_seeds(STEP, height, sy);
  int n = 0;
  for (int j = 0; j < ny; ++j)
    for (int i = 0; i < nx; ++i) seed[n++] = sy[j] * width + sx[i];
  free(sx); free(sy);
  return n;
}

static void slic_slico(const double *lv, const double *av, const double *bv, const int *seed, int numk, int width, int height,
                       int STEP, int *klabels) {                                               /* PerformSuperpixelSLICO :291-437 */
  const int sz = width * height;
  double *kx = malloc(sizeof(double) * numk), *ky = malloc(sizeof(double) * numk), *kl = malloc(sizeof(double) * numk),
         *ka = malloc(sizeof(double) * numk), *kb = malloc(sizeof(double) * numk), *maxlab = malloc(sizeof(double) * numk),
         *sum = calloc((size_t)6 * numk, sizeof(double)), *distvec = malloc(sizeof(double) * sz), *distlab = malloc(sizeof(double) * sz);
  for (int k = 0; k < numk; k++) {
    kx[k] = seed[k] % width; ky[k] = seed[k] / width;                                          /* :760-767 */
    kl[k] = lv[seed[k]]; ka[k] = av[seed[k]]; kb[k] = bv[seed[k]];
    maxlab[k] = 10.0 * 10.0;
  }
  for (int i = 0; i < sz; i++) { distlab[i] = 1.7976931348623157e308; klabels[i] = -1; }
  const double invxywt = 1.0 / (STEP * STEP);
  for (int itr = 0; itr < 10; itr++) {
    for (int i = 0; i < sz; i++) distvec[i] = 1.7976931348623157e308;
    for (int n = 0; n < numk; n++) {
      int x1 = kx[n] - STEP, y1 = ky[n] - STEP, x2 = kx[n] + STEP, y2 = ky[n] + STEP;          /* double -> int: truncation */
      if (x1 < 0) x1 = 0;
      if (y1 < 0) y1 = 0;
      if (x2 > width) x2 = width;
      if (y2 > height) y2 = height;
      for (int y = y1; y < y2; y++)
        for (int x = x1; x < x2; x++) {
          const int i = y * width + x;
          const double l = lv[i], a = av[i], b = bv[i];
          distlab[i] = (l - kl[n]) * (l - kl[n]) + (a - ka[n]) * (a - ka[n]) + (b - kb[n]) * (b - kb[n]);
          const double distxy = (x - kx[n]) * (x - kx[n]) + (y - ky[n]) * (y - ky[n]);
          const double dist = distlab[i] / maxlab[n] + distxy * invxywt;
          if (dist < distvec[i]) { distvec[i] = dist; klabels[i] = n; }
        }
    }
    if (itr == 0) for (int n = 0; n < numk; n++) maxlab[n] = 1.0;
    for (int i = 0; i < sz; i++)
      if (klabels[i] >= 0 && maxlab[klabels[i]] < distlab[i]) maxlab[klabels[i]] = distlab[i];
    memset(sum, 0, sizeof(double) * 6 * numk);
    int ind = 0;
    for (int r = 0; r < height; r++)
      for (int c = 0; c < width; c++, ind++)
        if (klabels[ind] >= 0) {
          double *s = sum + 6 * klabels[ind];
          s[0] += lv[ind]; s[1] += av[ind]; s[2] += bv[ind]; s[3] += c; s[4] += r; s[5] += 1.0;
        }
    for (int k = 0; k < numk; k++) {
      double *s = sum + 6 * k;
      if (s[5] <= 0) s[5] = 1;
      const double inv = 1.0 / s[5];
      kl[k] = s[0] * inv; ka[k] = s[1] * inv; kb[k] = s[2] * inv; kx[k] = s[3] * inv; ky[k] = s[4] * inv;
    }
  }
  free(kx); free(ky); free(kl); free(ka); free(kb); free(maxlab); free(sum); free(distvec); free(distlab);
}

static int slic_connectivity(const int *labels, int width, int height, int numSuperpixels, int *nlabels) {   /* :440-537 */
  const int dx4[4] = {-1, 0, 1, 0}, dy4[4] = {0, -1, 0, 1};
  const int sz = width * height, SUPSZ = sz / numSuperpixels;
  int *xvec = malloc(sizeof(int) * sz), *yvec = malloc(sizeof(int) * sz);
  for (int i = 0; i < sz; i++) nlabels[i] = -1;
  int oindex = 0, adjlabel = 0, label = 0;
  for (int j = 0; j < height; j++)
    for (int k = 0; k < width; k++, oindex++) {
      if (nlabels[oindex] >= 0) continue;
      nlabels[oindex] = label;
      xvec[0] = k; yvec[0] = j;
      for (int n = 0; n < 4; n++) {
        const int x = xvec[0] + dx4[n], y = yvec[0] + dy4[n];
        if (x >= 0 && x < width && y >= 0 && y < height && nlabels[y * width + x] >= 0) adjlabel = nlabels[y * width + x];
      }
      int count = 1;
      for (int c = 0; c < count; c++)
        for (int n = 0; n < 4; n++) {
          const int x = xvec[c] + dx4[n], y = yvec[c] + dy4[n];
          if (x >= 0 && x < width && y >= 0 && y < height) {
            const int ni = y * width + x;
            if (nlabels[ni] < 0 && labels[oindex] == labels[ni]) { xvec[count] = x; yvec[count] = y; nlabels[ni] = label; count++; }
          }
        }
      if (count <= SUPSZ >> 2) {
        for (int c = 0; c < count; c++) nlabels[yvec[c] * width + xvec[c]] = adjlabel;
        label--;
      }
      label++;
    }
  free(xvec); free(yvec);
  return label;
}

/* stack [nz][ny][nx] -> labels [nz][ny][nx] (the label image stack_spx as float, :811-819).  The slice goes into SLIC's buffer
 * with x outermost (:731-749), so SLIC sees a ny-wide, nx-high image: the transposed slice. */
int orc_segment_slic(const float *stack, int nx, int ny, int nz, int spx0, int spx1, float *labels_out) {
  const int width = ny, height = nx, sz = width * height;                                      /* :704-706 */
  float vmin = stack[0], vmax = stack[0];
  for (size_t i = 0; i < (size_t)sz * nz; i++) { if (stack[i] < vmin) vmin = stack[i]; if (stack[i] > vmax) vmax = stack[i]; }
  const int numSuperpixels = (int)(sz / (spx0 * spx1));                                        /* :710 */
  if (numSuperpixels < 1) return 1;
  int *grey = malloc(sizeof(int) * sz), *kl = malloc(sizeof(int) * sz), *cl = malloc(sizeof(int) * sz), *seed = malloc(sizeof(int) * sz);
  double *lv = malloc(sizeof(double) * sz), *av = malloc(sizeof(double) * sz), *bv = malloc(sizeof(double) * sz);
  for (int z = 0; z < nz; z++) {
    const float *sl = stack + (size_t)z * nx * ny;
    int p = 0;
    for (int x = 0; x < nx; ++x)
      for (int y = 0; y < ny; ++y, ++p)                                                        /* `(int) 255 * (v - min) / (max - min)`: float, truncated */
        grey[p] = vmax > vmin ? (int)(((float)255 * (sl[(size_t)y * nx + x] - vmin)) / (vmax - vmin)) : 0;
    slic_lab(grey, sz, lv, av, bv);
    const int step = (int)(sqrt((double)sz / (double)numSuperpixels) + 0.5);                   /* :755 */
    const int numseeds = slic_grid_seeds(step, width, height, seed);
    slic_slico(lv, av, bv, seed, numseeds, width, height, step, kl);
    slic_connectivity(kl, width, height, numSuperpixels, cl);
    p = 0;
    for (int x = 0; x < nx; ++x)
      for (int y = 0; y < ny; ++y, ++p) labels_out[((size_t)z * ny + y) * nx + x] = (float)cl[p];
  }
  free(grey); free(kl); free(cl); free(seed); free(lv); free(av); free(bv);
  return 0;
}

/* attributes of GetRegion(i1, j1, k1, i2, j2, k2) of an image (irtkGenericImage.cc:570-611) */
static void region_attr(const orc_attr *src, int i1, int j1, int k1, int i2, int j2, int k2, orc_attr *out) {
  double s_i2w[16], t[16];
  *out = *src;
  out->nx = i2 - i1; out->ny = j2 - j1; out->nz = k2 - k1;
  out->origin[0] = out->origin[1] = out->origin[2] = 0;
  double x1 = i1, y1 = j1, z1 = k1, x2 = 0, y2 = 0, z2 = 0;
  image_to_world(src, s_i2w);
  apply(s_i2w, &x1, &y1, &z1);
  image_to_world(out, t);
  apply(t, &x2, &y2, &z2);
  out->origin[0] = x1 - x2; out->origin[1] = y1 - y2; out->origin[2] = z1 - z2;
}

static void dilate_once(float *p, int nx, int ny) {                                            /* dilatePatch :347-367 */
  for (int j = 0; j < ny; j++)
    for (int i = 0; i < nx; i++)
      if (p[j * nx + i] == 1) {
        if (i > 0 && p[j * nx + i - 1] == 0) p[j * nx + i - 1] = 2;
        if (j > 0 && p[(j - 1) * nx + i] == 0) p[(j - 1) * nx + i] = 2;
        if (i + 1 < nx && p[j * nx + i + 1] == 0) p[j * nx + i + 1] = 2;
        if (j + 1 < ny && p[(j + 1) * nx + i] == 0) p[(j + 1) * nx + i] = 2;
      }
  for (int k = 0; k < nx * ny; k++) if (p[k] == 2) p[k] = 1;
}

/* stack / labels [nz][ny][nx] of `attr` (labels from orc_segment_slic); outputs for up to `cap` patches: data [n][py][px]
 * (-1 outside the dilated superpixel), spxmask [n][4096] ('1' / 0, 64 wide), i2w / w2i [n][16], origins [n][3].
 * pxy_out = {px, py} (64 x 64 clamped to the slice, :455-482).  Returns 0, or 1 when more than `cap` patches are kept. */
int orc_generate_2d_superpixel_patches(const orc_attr *attr, const float *stack, const float *labels, double thickness,
                                       const orc_attr *mask_attr, const float *mask, int spx_x, int spx_y, int extend_percent, int cap,
                                       float *data, unsigned char *spxmask, float *i2w, float *w2i, double *origins, int *n_out,
                                       long *total_pixels, int *pxy_out) {
  const float dilateRatio = (float)extend_percent / 100.f;                                     /* :449 */
  int pbx = 64, pby = 64;
  if (pbx > attr->nx) pbx = attr->nx;
  if (pby > attr->ny) pby = attr->ny;
  pxy_out[0] = pbx; pxy_out[1] = pby;
  double m_w2i[16];
  world_to_image(mask_attr, m_w2i);
  float *patch = malloc(sizeof(float) * (size_t)pbx * pby);
  int n = 0;
  for (int z = 0; z < attr->nz; z++) {
    orc_attr sa;
    region_attr(attr, 0, 0, z, attr->nx, attr->ny, z + 1, &sa);                                /* slice / spx_slice :495-499 */
    sa.dz = thickness * 2;
    double sl_w2i[16];
    world_to_image(&sa, sl_w2i);
    const float *slice = stack + (size_t)z * attr->nx * attr->ny, *lab = labels + (size_t)z * attr->nx * attr->ny;
    float minL = lab[0], maxL = lab[0];
    for (int k = 0; k < attr->nx * attr->ny; k++) { if (lab[k] < minL) minL = lab[k]; if (lab[k] > maxL) maxL = lab[k]; }
    for (int idx = (int)minL; idx < (int)maxL; idx++) {                                        /* :509: the largest label is never cut out */
      int xMin = 2147483647, yMin = 2147483647, xMax = -2147483647 - 1, yMax = -2147483647 - 1, exists = 0;
      for (int yi = 0; yi < attr->ny; yi++)
        for (int xi = 0; xi < attr->nx; xi++)
          if ((int)lab[yi * attr->nx + xi] == idx) {
            if (xi < xMin) xMin = xi;
            if (xi > xMax) xMax = xi;
            if (yi < yMin) yMin = yi;
            if (yi > yMax) yMax = yi;
            exists = 1;
          }
      if (!exists) continue;
      const unsigned sizex = (unsigned)(xMax - xMin), sizey = (unsigned)(yMax - yMin);
      const int diter = (int)(sizex > sizey ? dilateRatio * sizex : dilateRatio * sizey);      /* :541-546, float product */
      const unsigned ex = (unsigned)round(((float)pbx - (float)sizex) / 2.), ey = (unsigned)round(((float)pby - (float)sizey) / 2.);
      if ((int)(xMin - ex) < 0) { xMax = pbx; xMin = 0; }                                      /* :552-579 */
      else if ((int)(xMax + ex) > attr->nx) { xMax = attr->nx; xMin = xMax - pbx; }
      else { xMin -= ex; xMax = xMin + pbx; }
      if ((int)(yMin - ey) < 0) { yMax = pby; yMin = 0; }
      else if ((int)(yMax + ey) > attr->ny) { yMax = attr->ny; yMin = yMax - pby; }
      else { yMin -= ey; yMax = yMin + pby; }
      orc_attr pa;
      region_attr(attr, xMin, yMin, z, xMax, yMax, z + 1, &pa);                                /* :602-603 */
      pa.dz = thickness * 2;
      double p_i2w[16], p_w2i[16];
      image_to_world(&pa, p_i2w);
      world_to_image(&pa, p_w2i);
      int setCount = 0;
      for (int j = 0; j < pby; j++)
        for (int i = 0; i < pbx; i++) {                                                        /* :621-664 */
          double xx = i, yy = j, zz = 0;
          apply(p_i2w, &xx, &yy, &zz);
          double xx1 = xx, yy1 = yy, zz1 = zz;
          apply(sl_w2i, &xx, &yy, &zz);
          apply(m_w2i, &xx1, &yy1, &zz1);
          xx = irtk_round(xx); yy = irtk_round(yy);
          xx1 = irtk_round(xx1); yy1 = irtk_round(yy1); zz1 = irtk_round(zz1);
          float v = 0;
          if (xx >= 0 && yy >= 0 && xx < sa.nx && yy < sa.ny)
            if (xx1 >= 0 && yy1 >= 0 && zz1 >= 0 && xx1 < mask_attr->nx && yy1 < mask_attr->ny && zz1 < mask_attr->nz)
              if (mask[((size_t)(int)zz1 * mask_attr->ny + (size_t)(int)yy1) * mask_attr->nx + (size_t)(int)xx1] > 0)
                v = lab[(int)yy * attr->nx + (int)xx] == idx ? 1 : 0;
          patch[j * pbx + i] = v;
          if (v > 0) setCount++;
        }
      if (setCount < 2) continue;                                                              /* :667-668 */
      if (setCount < 1.0f / 4.0f * spx_y * spx_x) continue;
      for (int it = 0; it < diter; it++) dilate_once(patch, pbx, pby);
      for (int j = 0; j < pby; j++)
        for (int i = 0; i < pbx; i++) {                                                        /* :684-726 */
          if (patch[j * pbx + i] == 0) { patch[j * pbx + i] = -1; continue; }
          double xx = i, yy = j, zz = 0;
          apply(p_i2w, &xx, &yy, &zz);
          double xx1 = xx, yy1 = yy, zz1 = zz;
          apply(sl_w2i, &xx, &yy, &zz);
          apply(m_w2i, &xx1, &yy1, &zz1);
          xx = irtk_round(xx); yy = irtk_round(yy);
          xx1 = irtk_round(xx1); yy1 = irtk_round(yy1); zz1 = irtk_round(zz1);
          if (xx >= 0 && yy >= 0 && xx < sa.nx && yy < sa.ny)
            if (xx1 >= 0 && yy1 >= 0 && zz1 >= 0 && xx1 < mask_attr->nx && yy1 < mask_attr->ny && zz1 < mask_attr->nz) {
              if (mask[((size_t)(int)zz1 * mask_attr->ny + (size_t)(int)yy1) * mask_attr->nx + (size_t)(int)xx1] > 0)
                patch[j * pbx + i] = patch[j * pbx + i] == 1 ? slice[(int)yy * attr->nx + (int)xx] : -1;
              else
                patch[j * pbx + i] = -1;
            }
        }
      if (n >= cap) { free(patch); return 1; }
      unsigned char *mk = spxmask + (size_t)n * 4096;
      memset(mk, 0, 4096);
      for (int j = 0; j < pby; j++)
        for (int i = 0; i < pbx; i++)
          if (patch[j * pbx + i] != -1) { mk[i + 64 * j] = '1'; *total_pixels += 1; }          /* :728-737 */
      memcpy(data + (size_t)n * pbx * pby, patch, sizeof(float) * (size_t)pbx * pby);
      for (int q = 0; q < 16; ++q) { i2w[16 * (size_t)n + q] = (float)p_i2w[q]; w2i[16 * (size_t)n + q] = (float)p_w2i[q]; }
      origins[3 * (size_t)n] = pa.origin[0]; origins[3 * (size_t)n + 1] = pa.origin[1]; origins[3 * (size_t)n + 2] = pa.origin[2];
      ++n;
    }
  }
  free(patch);
  *n_out = n;
  return 0;
}

/* =====================================================================================================================
 * The rest of the pre-processing chain of reconstruction.cc:386-815, restated from the reference's own loops (round 3):
 * CreateTemplate (RG.cc:648-694 + irtkResampling<>::Initialize, irtkResampling.cc:74-130), SetMask (RG.cc:750-803: Gaussian
 * blurring, threshold, nearest-neighbour transformation onto the template), TransformMask (RG.cc:805-821), CropImage
 * (RG.cc:5205-5306 + GetRegion) and MaskSlices (RG.cc:1940-1988).  Written from the reference, not from csrc/svr_prep.h.
 * ===================================================================================================================== */

/* CreateTemplate: the stack's grid with two more planes (about the same origin = image centre), resampled to d x d x d:
 * dims int(n * old / d), at least 1 (then the old voxel size is kept), same axes and origin.  Returns d. */
double orc_create_template(const orc_attr *stack, double resolution, orc_attr *out) {
  double d;
  orc_attr a = *stack;
  a.nz += 2;                                             /* attr._z += 2 */
  if (resolution <= 0) {
    if ((a.dx <= a.dy) && (a.dx <= a.dz)) d = a.dx;
    else if (a.dy <= a.dz) d = a.dy;
    else d = a.dz;
  } else {
    d = resolution;
  }
  int new_x = (int)(a.nx * a.dx / d), new_y = (int)(a.ny * a.dy / d), new_z = (int)(a.nz * a.dz / d);
  double sx = d, sy = d, sz = d;
  if (new_x < 1) { new_x = 1; sx = a.dx; }
  if (new_y < 1) { new_y = 1; sy = a.dy; }
  if (new_z < 1) { new_z = 1; sz = a.dz; }
  *out = a;
  out->nx = new_x; out->ny = new_y; out->nz = new_z;
  out->dx = sx; out->dy = sy; out->dz = sz;
  return d;
}

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
/* irtkGaussianBlurring<irtkRealPixel>(sigma).Run() (irtkGaussianBlurring.cc:40-125): three 1-D passes (x, y, z -- the z pass
 * only if the image has more than one plane), each irtkConvolution_1D with normalisation (irtkConvolution_1D.cc:42-90: the
 * kernel is cut at the image border and the result divided by the sum of the weights that took part).  The kernel is
 * irtkScalarGaussian(sigma / voxel, 1, 1) sampled at the integer offsets -h..h, h = round(4 sigma / voxel), through
 * irtkScalarFunctionToImage (values below FLT_MIN become 0); its constant norm cancels in the normalisation but is kept. */
void orc_gaussian_blur(const orc_attr *a, double *data, double sigma) {
  const int n[3] = {a->nx, a->ny, a->nz};
  const double vs[3] = {a->dx, a->dy, a->dz};
  const size_t total = (size_t)a->nx * a->ny * a->nz;
  double *buf = (double *)malloc(total * sizeof(double));
  for (int axis = 0; axis < 3; ++axis) {
    if (axis == 2 && a->nz == 1) continue;               /* `if (this->_output->GetX() != 1)` after the two flips */
    const double s = sigma / vs[axis];
    const int h = (int)irtk_round(4 * sigma / vs[axis]);
    const int kn = 2 * h + 1;
    double *k = (double *)malloc(kn * sizeof(double));
    const double norm = 1.0 / (sqrt(2.0 * M_PI) * s * sqrt(2.0 * M_PI) * 1.0 * sqrt(2.0 * M_PI) * 1.0);
    for (int i = 0; i < kn; ++i) {
      const double x = (double)i - (kn - 1) / 2.0;       /* ImageToWorld of the kernel image: unit voxels about the origin */
      double v = norm * exp(-(x * x) / (2.0 * s * s) - 0.0 / (2.0 * 1.0 * 1.0) - 0.0 / (2.0 * 1.0 * 1.0));
      if (fabs(v) < FLT_MIN) v = 0;
      k[i] = v;
    }
    const size_t stride = axis == 0 ? 1 : (axis == 1 ? (size_t)a->nx : (size_t)a->nx * a->ny);
    for (int z = 0; z < a->nz; ++z)
      for (int y = 0; y < a->ny; ++y)
        for (int x = 0; x < a->nx; ++x) {
          const int p = axis == 0 ? x : (axis == 1 ? y : z);
          const size_t base = ((size_t)z * a->ny + y) * a->nx + x;
          const int x1 = p - kn / 2, x2 = p + kn / 2;
          double val = 0, sum = 0;
          for (int q = x1; q <= x2; ++q)
            if (q >= 0 && q < n[axis]) {
              val += k[q - x1] * data[base + (ptrdiff_t)(q - p) * (ptrdiff_t)stride];
              sum += k[q - x1];
            }
          buf[base] = sum > 0 ? val / sum : 0;
        }
    memcpy(data, buf, total * sizeof(double));
    free(k);
  }
  free(buf);
}

/* irtkImageTransformation::Run (irtkImageTransformation.cc:200-324) with the nearest-neighbour interpolator: every target
 * voxel above the target padding value is taken to world coordinates, through the transformation, into the source grid
 * (three separate applications); inside the source's field of view (-0.5 < x < n - 0.5, strictly) it takes the nearest
 * source voxel, everywhere else the source padding value.  target: in/out. */
void orc_transform_nn(const orc_attr *src_attr, const double *src, const orc_attr *tgt_attr, double *tgt, const double *T,
                      double target_padding, double source_padding) {
  double t_i2w[16], s_w2i[16];
  image_to_world(tgt_attr, t_i2w);
  world_to_image(src_attr, s_w2i);
  for (int k = 0; k < tgt_attr->nz; ++k)
    for (int j = 0; j < tgt_attr->ny; ++j)
      for (int i = 0; i < tgt_attr->nx; ++i) {
        double *o = &tgt[((size_t)k * tgt_attr->ny + j) * tgt_attr->nx + i];
        if (*o > target_padding) {
          double x = i, y = j, z = k;
          apply(t_i2w, &x, &y, &z);
          apply(T, &x, &y, &z);
          apply(s_w2i, &x, &y, &z);
          if ((x > -0.5) && (x < src_attr->nx - 0.5) && (y > -0.5) && (y < src_attr->ny - 0.5) && (z > -0.5) && (z < src_attr->nz - 0.5)) {
            const int a = (int)irtk_round(x), b = (int)irtk_round(y), c = (int)irtk_round(z);
            *o = src[((size_t)c * src_attr->ny + b) * src_attr->nx + a];
          } else {
            *o = source_padding;
          }
        } else {
          *o = source_padding;
        }
      }
}

/* SetMask (RG.cc:750-803).  mask: in/out like the reference's (it is blurred and binarised in place when sigma > 0);
 * out: the mask on the template grid (the template is an all-zero image: every voxel is above the target padding -1).
 * mask == NULL: ones. */
void orc_set_mask(const orc_attr *tmpl, const orc_attr *mask_attr, double *mask, double sigma, double threshold, double *out) {
  const size_t nt = (size_t)tmpl->nx * tmpl->ny * tmpl->nz;
  if (!mask) { for (size_t i = 0; i < nt; ++i) out[i] = 1; return; }
  if (sigma > 0) {
    orc_gaussian_blur(mask_attr, mask, sigma);
    const size_t nm = (size_t)mask_attr->nx * mask_attr->ny * mask_attr->nz;
    for (size_t i = 0; i < nm; ++i) mask[i] = mask[i] > threshold ? 1 : 0;
  }
  double I[16];
  ident(I);
  for (size_t i = 0; i < nt; ++i) out[i] = 0;            /* _mask = _reconstructed: the zero template */
  orc_transform_nn(mask_attr, mask, tmpl, out, I, -1, 0);
}

/* TransformMask (RG.cc:805-821): `m = image` -- the target starts as a COPY OF THE IMAGE, so a voxel of the image at or below
 * the target padding -1 is not looked up but set to 0 -- then the mask is resampled onto it.  image_as_target: in/out. */
void orc_transform_mask(const orc_attr *image_attr, double *image_as_target, const orc_attr *mask_attr, const double *mask, const double *T) {
  orc_transform_nn(mask_attr, mask, image_attr, image_as_target, T, -1, 0);
}

/* CropImage (RG.cc:5205-5306): the bounding box of mask > 0, found plane by plane from either end of every axis;
 * bounds6 = {x1, y1, z1, x2, y2, z2} (inclusive; the region cut is [x1, x2 + 1) ...).  A mask without a voxel > 0 leaves
 * x2 = y2 = z2 = -1 and x1 = nx ... like the reference's loops.  region: the attributes of the cut image. */
void orc_crop_bounds(const orc_attr *a, const double *mask, int bounds6[6], orc_attr *region) {
  int i, j, k, sum;
#define M_(i, j, k) mask[((size_t)(k) * a->ny + (j)) * a->nx + (i)]
  for (k = a->nz - 1; k >= 0; k--) { sum = 0; for (j = a->ny - 1; j >= 0; j--) for (i = a->nx - 1; i >= 0; i--) if (M_(i, j, k) > 0) sum++; if (sum > 0) break; }
  const int z2 = k;
  for (k = 0; k <= a->nz - 1; k++) { sum = 0; for (j = a->ny - 1; j >= 0; j--) for (i = a->nx - 1; i >= 0; i--) if (M_(i, j, k) > 0) sum++; if (sum > 0) break; }
  const int z1 = k;
  for (j = a->ny - 1; j >= 0; j--) { sum = 0; for (k = a->nz - 1; k >= 0; k--) for (i = a->nx - 1; i >= 0; i--) if (M_(i, j, k) > 0) sum++; if (sum > 0) break; }
  const int y2 = j;
  for (j = 0; j <= a->ny - 1; j++) { sum = 0; for (k = a->nz - 1; k >= 0; k--) for (i = a->nx - 1; i >= 0; i--) if (M_(i, j, k) > 0) sum++; if (sum > 0) break; }
  const int y1 = j;
  for (i = a->nx - 1; i >= 0; i--) { sum = 0; for (k = a->nz - 1; k >= 0; k--) for (j = a->ny - 1; j >= 0; j--) if (M_(i, j, k) > 0) sum++; if (sum > 0) break; }
  const int x2 = i;
  for (i = 0; i <= a->nx - 1; i++) { sum = 0; for (k = a->nz - 1; k >= 0; k--) for (j = a->ny - 1; j >= 0; j--) if (M_(i, j, k) > 0) sum++; if (sum > 0) break; }
  const int x1 = i;
#undef M_
  bounds6[0] = x1; bounds6[1] = y1; bounds6[2] = z1; bounds6[3] = x2; bounds6[4] = y2; bounds6[5] = z2;
  if (region && x2 >= x1 && y2 >= y1 && z2 >= z1) region_attr(a, x1, y1, z1, x2 + 1, y2 + 1, z2 + 1, region);
}

/* MaskSlices (RG.cc:1940-1988) for one slice: values below 0.01 are padding (-1); a pixel whose position -- slice
 * ImageToWorld, the slice's transformation, the mask's WorldToImage, rounded -- is outside the mask's grid or on a zero of
 * the mask is -1.  slice: in/out [ny][nx]. */
void orc_mask_slice(const orc_attr *slice_attr, double *slice, const double *T, const orc_attr *mask_attr, const double *mask) {
  double s_i2w[16], m_w2i[16];
  image_to_world(slice_attr, s_i2w);
  world_to_image(mask_attr, m_w2i);
  for (int i = 0; i < slice_attr->nx; i++)
    for (int j = 0; j < slice_attr->ny; j++) {
      double *v = &slice[(size_t)j * slice_attr->nx + i];
      if (*v < 0.01) *v = -1;
      double x = i, y = j, z = 0;
      apply(s_i2w, &x, &y, &z);
      apply(T, &x, &y, &z);
      apply(m_w2i, &x, &y, &z);
      x = irtk_round(x); y = irtk_round(y); z = irtk_round(z);
      if ((x >= 0) && (x < mask_attr->nx) && (y >= 0) && (y < mask_attr->ny) && (z >= 0) && (z < mask_attr->nz)) {
        if (mask[((size_t)(int)z * mask_attr->ny + (int)y) * mask_attr->nx + (int)x] == 0) *v = -1;
      } else {
        *v = -1;
      }
    }
}

/* attributes of image.GetRegion(i1, j1, k1, i2, j2, k2) (irtkGenericImage.cc:570-611), e.g. of one slice of a stack */
void orc_get_region_attr(const orc_attr *src, int i1, int j1, int k1, int i2, int j2, int k2, orc_attr *out) {
  region_attr(src, i1, j1, k1, i2, j2, k2, out);
}

/* ---- packages: SplitImage / SplitImageEvenOdd / SplitImageEvenOddHalf / HalfImage and the slice assignment of
 * PackageToVolume (RG.cc:4980-5192) -------------------------------------------------------------------------------------
 * A package is carried as its attributes and the list of the STACK slices its planes hold (plane k of a package made by
 * SplitImage(image, packages) from its package l is plane k * packages + l of `image`, RG.cc:5013; HalfImage's GetRegion keeps
 * planes [0, nz/2) and [nz/2, nz), RG.cc:5083-5086).  Voxel data follow their planes and are not carried here. */
typedef struct { orc_attr a; int n; int src[1024]; } orc_pack;

/* SplitImage RG.cc:4980-5038: package l takes planes l, l + packages, ...; its dz is packages * dz; the origin is moved so that its
 * voxel (0,0,0) sits where voxel (0,0,l) of the image sits */
static int split_image(const orc_pack *img, int packages, orc_pack *out) {
  const int pkg_z = img->a.nz / packages;
  const double pkg_dz = img->a.dz * packages;
  for (int l = 0; l < packages; l++) {
    orc_pack *p = &out[l];
    p->a = img->a;
    p->a.nz = (pkg_z * packages + l) < img->a.nz ? pkg_z + 1 : pkg_z;
    p->a.dz = pkg_dz;
    p->n = p->a.nz;
    for (int k = 0; k < p->a.nz; k++) p->src[k] = img->src[k * packages + l];
    double i2w[16], s2w[16];
    double x = 0, y = 0, z = l, sx = 0, sy = 0, sz = 0;
    image_to_world(&img->a, i2w);
    apply(i2w, &x, &y, &z);
    image_to_world(&p->a, s2w);                      /* (the stack still carries the image's origin: PutOrigin(ox, oy, oz)) */
    apply(s2w, &sx, &sy, &sz);
    p->a.origin[0] += x - sx; p->a.origin[1] += y - sy; p->a.origin[2] += z - sz;
  }
  return packages;
}
/* HalfImage RG.cc:5072-5091 */
static int half_image(const orc_pack *img, orc_pack *out) {
  if (img->a.nz >= 4) {
    const int h = img->a.nz / 2;
    region_attr(&img->a, 0, 0, 0, img->a.nx, img->a.ny, h, &out[0].a);
    out[0].n = h;
    for (int k = 0; k < h; k++) out[0].src[k] = img->src[k];
    region_attr(&img->a, 0, 0, h, img->a.nx, img->a.ny, img->a.nz, &out[1].a);
    out[1].n = img->a.nz - h;
    for (int k = h; k < img->a.nz; k++) out[1].src[k - h] = img->src[k];
    return 2;
  }
  out[0] = *img;
  return 1;
}
static int split_even_odd(const orc_pack *img, int packages, orc_pack *out) {          /* RG.cc:5040-5058 */
  orc_pack *packs = (orc_pack *)malloc(sizeof(orc_pack) * (size_t)packages);
  const int n = split_image(img, packages, packs);
  int m = 0;
  for (int i = 0; i < n; i++) m += split_image(&packs[i], 2, out + m);
  free(packs);
  return m;
}
static int split_even_odd_half(const orc_pack *img, int packages, orc_pack *out, int iter) {   /* RG.cc:5060-5079 */
  orc_pack *packs = (orc_pack *)malloc(sizeof(orc_pack) * 1024);
  const int n = iter > 1 ? split_even_odd_half(img, packages, packs, iter - 1) : split_even_odd(img, packages, packs);
  int m = 0;
  for (int i = 0; i < n; i++) m += half_image(&packs[i], out + m);
  free(packs);
  return m;
}
/* The packages of one stack and, as PackageToVolume finds them (RG.cc:5131-5138, 5167-5172: ImageToWorld of the package's plane,
 * WorldToImage of the stack, round), the stack slice every package plane is assigned to.
 * out: n_packages; pack_nz[p]; slice_of[p * max_nz + k] = assigned slice (from the geometry), plane_src[...] = the stack plane whose
 * voxels the package plane holds (from the splitting); pack_attr[p].  Returns the number of packages, or -1 (more than max_packs
 * packages, more than max_nz planes, or more than 1024 slices). */
int orc_split_packages(const orc_attr *stack, int packages, int evenodd, int half, int half_iter, int max_packs, int max_nz, int *pack_nz,
                       int *slice_of, int *plane_src, orc_attr *pack_attr) {
  if (stack->nz > 1024 || packages < 1) return -1;
  orc_pack *img = (orc_pack *)malloc(sizeof(orc_pack)), *out = (orc_pack *)malloc(sizeof(orc_pack) * 1024);
  img->a = *stack;
  img->n = stack->nz;
  for (int k = 0; k < stack->nz; k++) img->src[k] = k;
  int n;
  if (evenodd) n = half ? split_even_odd_half(img, packages, out, half_iter) : split_even_odd(img, packages, out);
  else n = split_image(img, packages, out);
  int rc = n;
  if (n > max_packs) rc = -1;
  double s_w2i[16];
  world_to_image(stack, s_w2i);
  for (int p = 0; p < n && rc >= 0; p++) {
    if (out[p].a.nz > max_nz) { rc = -1; break; }
    pack_nz[p] = out[p].a.nz;
    pack_attr[p] = out[p].a;
    double p_i2w[16];
    image_to_world(&out[p].a, p_i2w);
    for (int k = 0; k < out[p].a.nz; k++) {
      double x = 0, y = 0, z = k;
      apply(p_i2w, &x, &y, &z);
      apply(s_w2i, &x, &y, &z);
      slice_of[p * max_nz + k] = (int)irtk_round(z);
      plane_src[p * max_nz + k] = out[p].src[k];
    }
  }
  free(img); free(out);
  return rc;
}
