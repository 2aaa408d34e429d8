/* nifti_ref_driver.c -- TEST INFRASTRUCTURE ONLY.  A thin flat-argument harness over the reference's own vendored
 * NIfTI-1 library (source/IRTKSimple2/nifti/niftilib/nifti1_io.c + znzlib/znzlib.c, compiled where they lie by
 * oracle/Makefile into oracle/_ref/libnifti_ref.so; nothing is copied).  tests/test_nifti.py uses it to pin
 * csrc/svr_io.cpp: files written by the engine library must parse in the reference's library with the header
 * IRTK would have produced, and files written by the reference's library (any datatype, scaled, gzip) must read
 * back identically through svr_nifti_read.  The calls below are the ones irtkNIFTIHeader / irtkFileNIFTIToImage /
 * irtkImageToFileNIFTI make (image++/include/irtkNIFTI.h:84-160, irtkFileNIFTIToImage.cc:168-345). */
#include <stdlib.h>
#include <string.h>

#include "nifti1_io.h"

/* header + (optionally) raw voxel bytes of a file, through nifti_image_read */
int ref_nifti_read(const char *path, int dims[5], float pixdim[4], int codes[4], float qto[16], float sto[16],
                   float scl[2], void *data_or_null, long capacity_bytes) {
  nifti_image *nim = nifti_image_read(path, data_or_null != NULL);
  if (!nim) return 1;
  dims[0] = nim->ndim; dims[1] = nim->nx; dims[2] = nim->ny; dims[3] = nim->nz; dims[4] = nim->nt;
  pixdim[0] = nim->dx; pixdim[1] = nim->dy; pixdim[2] = nim->dz; pixdim[3] = nim->qfac;
  codes[0] = nim->qform_code; codes[1] = nim->sform_code; codes[2] = nim->datatype; codes[3] = nim->nbyper;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) { qto[4 * i + j] = nim->qto_xyz.m[i][j]; sto[4 * i + j] = nim->sto_xyz.m[i][j]; }
  scl[0] = nim->scl_slope; scl[1] = nim->scl_inter;
  int rc = 0;
  if (data_or_null) {
    const long need = (long)nim->nvox * nim->nbyper;
    if (need > capacity_bytes) rc = 2;
    else memcpy(data_or_null, nim->data, (size_t)need);
  }
  nifti_image_free(nim);
  return rc;
}

/* a single-file NIfTI-1 through nifti_image_write; qto/sto are row-major 4x4 or NULL */
int ref_nifti_write(const char *path, const int dims3[3], int nt, const float pix3[3], int datatype, const float *qto,
                    const float *sto, float slope, float inter, const void *data) {
  nifti_image *nim = nifti_simple_init_nim();
  if (!nim) return 1;
  int nbyper = 0, swap = 0;
  nifti_datatype_sizes(datatype, &nbyper, &swap);
  nim->datatype = datatype;
  nim->nbyper = nbyper;
  nim->nifti_type = 1;
  nim->ndim = nt > 1 ? 4 : 3;
  nim->nx = dims3[0]; nim->ny = dims3[1]; nim->nz = dims3[2]; nim->nt = nt > 1 ? nt : 1;
  nim->nu = nim->nv = nim->nw = 1;
  nim->dim[0] = nim->ndim; nim->dim[1] = nim->nx; nim->dim[2] = nim->ny; nim->dim[3] = nim->nz; nim->dim[4] = nim->nt;
  nim->dim[5] = nim->dim[6] = nim->dim[7] = 1;
  nim->dx = pix3[0]; nim->dy = pix3[1]; nim->dz = pix3[2]; nim->dt = 1;
  nim->pixdim[1] = pix3[0]; nim->pixdim[2] = pix3[1]; nim->pixdim[3] = pix3[2]; nim->pixdim[4] = 1;
  nim->nvox = (size_t)nim->nx * nim->ny * nim->nz * nim->nt;
  nim->qform_code = 0;
  nim->sform_code = 0;
  if (qto) {
    mat44 m;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m.m[i][j] = qto[4 * i + j];
    nim->qform_code = 1;
    nim->qto_xyz = m;
    nifti_mat44_to_quatern(m, &nim->quatern_b, &nim->quatern_c, &nim->quatern_d, &nim->qoffset_x, &nim->qoffset_y,
                           &nim->qoffset_z, &nim->dx, &nim->dy, &nim->dz, &nim->qfac);
    nim->pixdim[0] = nim->qfac; nim->pixdim[1] = nim->dx; nim->pixdim[2] = nim->dy; nim->pixdim[3] = nim->dz;
  }
  if (sto) {
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) nim->sto_xyz.m[i][j] = sto[4 * i + j];
    nim->sform_code = 1;
  }
  nim->xyz_units = NIFTI_UNITS_MM;
  nim->time_units = NIFTI_UNITS_MSEC;
  nim->scl_slope = slope;
  nim->scl_inter = inter;
  if (nifti_set_filenames(nim, path, 0, 1)) { nifti_image_free(nim); return 2; }
  nim->data = malloc((size_t)nim->nvox * nbyper);
  memcpy(nim->data, data, (size_t)nim->nvox * nbyper);
  nifti_image_write(nim);
  nifti_image_free(nim);
  return 0;
}

/* the quaternion round trip on its own */
void ref_quatern_round_trip(const float in16[16], float out16[16], float q[10]) {
  mat44 m;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m.m[i][j] = in16[4 * i + j];
  nifti_mat44_to_quatern(m, &q[0], &q[1], &q[2], &q[3], &q[4], &q[5], &q[6], &q[7], &q[8], &q[9]);
  mat44 r = nifti_quatern_to_mat44(q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9]);
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out16[4 * i + j] = r.m[i][j];
}
