/*
 * prep_oracle.c -- CPU restatement of two host functions that feed the hot path:
 *
 *   orc_match_stack_intensities   irtkReconstruction::MatchStackIntensitiesWithMasking
 *                                 (source/reconstructionGPU2/irtkReconstructionGPU.cc:1375-1493)
 *   orc_generate_2d_patches       PatchBasedObject<T>::generate2DPatches
 *                                 (source/reconstructionGPU2/include/patchBasedObject.cuh:174-342)
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
#include <math.h>
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
