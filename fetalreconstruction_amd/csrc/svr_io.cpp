// svr_io.cpp -- NIfTI-1 (.nii / .nii.gz) reader and writer with the reference's image conventions
// (first item of the pre-processing chain).  Plain host C++ over zlib.
//
// What the reference does (IRTKSimple2/image++/src/irtkFileNIFTIToImage.cc:168-345,
// irtkImageToFileNIFTI.cc:65-145, include/irtkNIFTI.h:84-160, through niftilib):
//   read : voxel counts / |pixdim|; the qform (if qform_code > 0) else the sform else a default
//          radiological matrix D (:257-289); axes = columns of D / voxel size (:303-307);
//          origin = D * (centre voxel) (:309-325, R = D M D^-1 reduces to that); scl_slope / scl_inter;
//          byte swapping; data in file order (x fastest).
//   write: single-file NIfTI-1 ("n+1", vox_offset 352), qform_code 1 from the image-to-world matrix,
//          sform_code 0, units mm / ms, slope 1, inter 0.
// The header layout and the quaternion <-> matrix relations are the NIfTI-1 standard's (nifti1.h
// documentation); nothing of niftilib is used.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sched.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <string>
#include <exception>
#include <vector>

#include "../../include/svr_host.h"

namespace {

#pragma pack(push, 1)
struct Nifti1Header {                 // 348 bytes, NIfTI-1 standard
  int32_t sizeof_hdr;
  char data_type[10], db_name[18];
  int32_t extents;
  int16_t session_error;
  char regular, dim_info;
  int16_t dim[8];
  float intent_p1, intent_p2, intent_p3;
  int16_t intent_code, datatype, bitpix, slice_start;
  float pixdim[8], vox_offset, scl_slope, scl_inter;
  int16_t slice_end;
  char slice_code, xyzt_units;
  float cal_max, cal_min, slice_duration, toffset;
  int32_t glmax, glmin;
  char descrip[80], aux_file[24];
  int16_t qform_code, sform_code;
  float quatern_b, quatern_c, quatern_d, qoffset_x, qoffset_y, qoffset_z;
  float srow_x[4], srow_y[4], srow_z[4];
  char intent_name[16], magic[4];
};
#pragma pack(pop)
static_assert(sizeof(Nifti1Header) == 348, "NIfTI-1 header is 348 bytes");

template <class T> T bswap(T v) {
  unsigned char *p = reinterpret_cast<unsigned char *>(&v);
  for (size_t i = 0; i < sizeof(T) / 2; ++i) std::swap(p[i], p[sizeof(T) - 1 - i]);
  return v;
}
void swap_header(Nifti1Header &h) {
  h.sizeof_hdr = bswap(h.sizeof_hdr); h.extents = bswap(h.extents); h.session_error = bswap(h.session_error);
  for (int i = 0; i < 8; ++i) { h.dim[i] = bswap(h.dim[i]); h.pixdim[i] = bswap(h.pixdim[i]); }
  h.intent_p1 = bswap(h.intent_p1); h.intent_p2 = bswap(h.intent_p2); h.intent_p3 = bswap(h.intent_p3);
  h.intent_code = bswap(h.intent_code); h.datatype = bswap(h.datatype); h.bitpix = bswap(h.bitpix);
  h.slice_start = bswap(h.slice_start); h.vox_offset = bswap(h.vox_offset); h.scl_slope = bswap(h.scl_slope);
  h.scl_inter = bswap(h.scl_inter); h.slice_end = bswap(h.slice_end); h.cal_max = bswap(h.cal_max);
  h.cal_min = bswap(h.cal_min); h.slice_duration = bswap(h.slice_duration); h.toffset = bswap(h.toffset);
  h.glmax = bswap(h.glmax); h.glmin = bswap(h.glmin); h.qform_code = bswap(h.qform_code);
  h.sform_code = bswap(h.sform_code); h.quatern_b = bswap(h.quatern_b); h.quatern_c = bswap(h.quatern_c);
  h.quatern_d = bswap(h.quatern_d); h.qoffset_x = bswap(h.qoffset_x); h.qoffset_y = bswap(h.qoffset_y);
  h.qoffset_z = bswap(h.qoffset_z);
  for (int i = 0; i < 4; ++i) { h.srow_x[i] = bswap(h.srow_x[i]); h.srow_y[i] = bswap(h.srow_y[i]); h.srow_z[i] = bswap(h.srow_z[i]); }
}

// NIfTI-1 standard, "quaternion representation of rotation matrix": R from (b, c, d), a = sqrt(1 - b^2 - c^2 - d^2);
// columns scaled by pixdim, the third also by qfac = pixdim[0] (-1 or +1)
void qform_matrix(const Nifti1Header &h, double D[16]) {
  double b = h.quatern_b, c = h.quatern_c, d = h.quatern_d;
  double a = 1.0 - (b * b + c * c + d * d);
  if (a < 1e-7) {                       // special case of the standard: a 180 degree rotation
    a = 1.0 / sqrt(b * b + c * c + d * d);
    b *= a; c *= a; d *= a; a = 0.0;
  } else {
    a = sqrt(a);
  }
  const double xd = h.pixdim[1] > 0 ? h.pixdim[1] : 1.0, yd = h.pixdim[2] > 0 ? h.pixdim[2] : 1.0;
  double zd = h.pixdim[3] > 0 ? h.pixdim[3] : 1.0;
  if (h.pixdim[0] < 0.0) zd = -zd;
  D[0] = (a * a + b * b - c * c - d * d) * xd; D[1] = 2.0 * (b * c - a * d) * yd; D[2] = 2.0 * (b * d + a * c) * zd;
  D[4] = 2.0 * (b * c + a * d) * xd; D[5] = (a * a + c * c - b * b - d * d) * yd; D[6] = 2.0 * (c * d - a * b) * zd;
  D[8] = 2.0 * (b * d - a * c) * xd; D[9] = 2.0 * (c * d + a * b) * yd; D[10] = (a * a + d * d - c * c - b * b) * zd;
  D[3] = h.qoffset_x; D[7] = h.qoffset_y; D[11] = h.qoffset_z;
  D[12] = D[13] = D[14] = 0.0; D[15] = 1.0;
}

// inverse relation: unit quaternion (a >= 0) of a proper rotation matrix R (row-major 3x3)
void rotation_to_quatern(const double R[9], double &b, double &c, double &d) {
  double a = R[0] + R[4] + R[8] + 1.0;
  if (a > 0.5) {
    a = 0.5 * sqrt(a);
    b = 0.25 * (R[7] - R[5]) / a; c = 0.25 * (R[2] - R[6]) / a; d = 0.25 * (R[3] - R[1]) / a;
  } else {
    const double xd = 1.0 + R[0] - (R[4] + R[8]), yd = 1.0 + R[4] - (R[0] + R[8]), zd = 1.0 + R[8] - (R[0] + R[4]);
    if (xd > 1.0) {
      b = 0.5 * sqrt(xd); c = 0.25 * (R[1] + R[3]) / b; d = 0.25 * (R[2] + R[6]) / b; a = 0.25 * (R[7] - R[5]) / b;
    } else if (yd > 1.0) {
      c = 0.5 * sqrt(yd); b = 0.25 * (R[1] + R[3]) / c; d = 0.25 * (R[5] + R[7]) / c; a = 0.25 * (R[2] - R[6]) / c;
    } else {
      d = 0.5 * sqrt(zd); b = 0.25 * (R[2] + R[6]) / d; c = 0.25 * (R[5] + R[7]) / d; a = 0.25 * (R[3] - R[1]) / d;
    }
    if (a < 0.0) { b = -b; c = -c; d = -d; }
  }
}

void image_to_world(const svr_image_attr &a, double M[16]) {     // irtkBaseImage.cc:79-111
  const double c[3] = {(a.nx - 1) / 2.0, (a.ny - 1) / 2.0, (a.nz - 1) / 2.0};
  for (int i = 0; i < 3; ++i) {
    M[4 * i + 0] = a.xaxis[i] * a.dx; M[4 * i + 1] = a.yaxis[i] * a.dy; M[4 * i + 2] = a.zaxis[i] * a.dz;
    M[4 * i + 3] = a.origin[i] - (M[4 * i + 0] * c[0] + M[4 * i + 1] * c[1] + M[4 * i + 2] * c[2]);
  }
  M[12] = M[13] = M[14] = 0.0; M[15] = 1.0;
}

int set_err(char *err, const std::string &m) {
  if (err) { strncpy(err, m.c_str(), 255); err[255] = 0; }
  return SVR_E_ARG;
}

}  // namespace

extern "C" {

int svr_nifti_read(const char *path, svr_image_attr *attr, int *nt, float **data, char err[256]) {
  if (!path || !attr || !data) return SVR_E_ARG;
  *data = nullptr;
  gzFile f = gzopen(path, "rb");                 // reads plain and gzip files alike
  if (!f) return set_err(err, std::string("cannot open ") + path);
  Nifti1Header h;
  if (gzread(f, &h, sizeof(h)) != (int)sizeof(h)) { gzclose(f); return set_err(err, "short NIfTI header"); }
  bool swapped = false;
  if (h.sizeof_hdr != 348) {
    swap_header(h);
    swapped = true;
    if (h.sizeof_hdr != 348) { gzclose(f); return set_err(err, "not a NIfTI-1 file (sizeof_hdr != 348)"); }
  }
  if (!(h.magic[0] == 'n' && h.magic[1] == '+' && h.magic[2] == '1')) {
    gzclose(f);
    return set_err(err, "not a single-file NIfTI-1 image (magic != n+1)");
  }
  if (h.dim[0] < 1 || h.dim[0] > 5 || (h.dim[0] == 5 && h.dim[4] != 1)) { gzclose(f); return set_err(err, "unsupported number of dimensions"); }
  svr_image_attr a;
  memset(&a, 0, sizeof(a));
  a.nx = h.dim[1]; a.ny = h.dim[0] >= 2 ? h.dim[2] : 1; a.nz = h.dim[0] >= 3 ? h.dim[3] : 1;
  a.dx = fabs(h.pixdim[1]); a.dy = fabs(h.pixdim[2]); a.dz = fabs(h.pixdim[3]);     // :239-244
  const int t = h.dim[0] == 4 ? h.dim[4] : (h.dim[0] == 5 ? h.dim[5] : 1);
  if (a.nx < 1 || a.ny < 1 || a.nz < 1 || t < 1 || !(a.dx > 0) || !(a.dy > 0) || !(a.dz > 0)) {
    gzclose(f);
    return set_err(err, "bad dimensions or voxel sizes");
  }
  double D[16];
  if (h.qform_code > 0) {
    qform_matrix(h, D);
  } else if (h.sform_code > 0) {
    for (int j = 0; j < 4; ++j) { D[j] = h.srow_x[j]; D[4 + j] = h.srow_y[j]; D[8 + j] = h.srow_z[j]; }
    D[12] = D[13] = D[14] = 0; D[15] = 1;
  } else {                                                                           // :262-289
    memset(D, 0, sizeof(D));
    D[0] = -a.dx; D[5] = a.dy; D[10] = a.dz; D[15] = 1;
    D[3] = a.dx * (a.nx - 1) / 2.0; D[7] = -a.dy * (a.ny - 1) / 2.0; D[11] = -a.dz * (a.nz - 1) / 2.0;
  }
  const double c[3] = {(a.nx - 1) / 2.0, (a.ny - 1) / 2.0, (a.nz - 1) / 2.0};
  for (int i = 0; i < 3; ++i) {                                                      // :303-325
    a.xaxis[i] = D[4 * i + 0] / a.dx; a.yaxis[i] = D[4 * i + 1] / a.dy; a.zaxis[i] = D[4 * i + 2] / a.dz;
    a.origin[i] = D[4 * i + 0] * c[0] + D[4 * i + 1] * c[1] + D[4 * i + 2] * c[2] + D[4 * i + 3];
  }
  int bytes = 0;
  switch (h.datatype) {
    case 2: case 256: bytes = 1; break;          // uint8, int8
    case 4: case 512: bytes = 2; break;          // int16, uint16
    case 8: case 768: case 16: bytes = 4; break; // int32, uint32, float32
    case 64: bytes = 8; break;                   // float64
    default: gzclose(f); return set_err(err, "unsupported NIfTI datatype " + std::to_string(h.datatype));
  }
  // a header is 16-bit counts times each other: bound the product before anything is allocated (no exception may cross
  // the C boundary, and a corrupt header must not ask for terabytes)
  const double n_d = (double)a.nx * a.ny * a.nz * t;
  if (!(n_d * bytes <= 64.0 * 1024 * 1024 * 1024)) { gzclose(f); return set_err(err, "image larger than 64 GiB: corrupt header?"); }
  const size_t n = (size_t)a.nx * a.ny * a.nz * t;
  const long off = (long)h.vox_offset >= 348 ? (long)h.vox_offset : 352;
  if (gzseek(f, off, SEEK_SET) < 0) { gzclose(f); return set_err(err, "seek to vox_offset failed"); }
  std::vector<unsigned char> raw;
  try {
    raw.resize(n * bytes);
  } catch (const std::exception &) {
    gzclose(f);
    return set_err(err, "out of memory");
  }
  size_t got = 0;
  while (got < raw.size()) {
    const int r = gzread(f, raw.data() + got, (unsigned)std::min<size_t>(raw.size() - got, 1u << 30));
    if (r <= 0) break;
    got += (size_t)r;
  }
  gzclose(f);
  if (got != raw.size()) return set_err(err, "short NIfTI data");
  float *out = (float *)malloc(n * sizeof(float));
  if (!out) return set_err(err, "out of memory");
  const double slope = h.scl_slope != 0 ? h.scl_slope : 1.0, inter = h.scl_inter;    // :225-231
  for (size_t i = 0; i < n; ++i) {
    const unsigned char *p = raw.data() + i * bytes;
    double v = 0;
#define RD(T) ([&] { T x; memcpy(&x, p, sizeof(T)); return swapped ? bswap(x) : x; }())
    switch (h.datatype) {
      case 2: v = *p; break;
      case 256: v = (signed char)*p; break;
      case 4: v = RD(int16_t); break;
      case 512: v = RD(uint16_t); break;
      case 8: v = RD(int32_t); break;
      case 768: v = RD(uint32_t); break;
      case 16: v = RD(float); break;
      case 64: v = RD(double); break;
    }
#undef RD
    out[i] = (float)(v * slope + inter);
  }
  *attr = a;
  if (nt) *nt = t;
  *data = out;
  return SVR_OK;
}

int svr_nifti_write(const char *path, const svr_image_attr *attr, const float *data, char err[256]) {
  if (!path || !attr || !data) return SVR_E_ARG;
  const svr_image_attr &a = *attr;
  if (a.nx < 1 || a.ny < 1 || a.nz < 1 || a.nx > 32767 || a.ny > 32767 || a.nz > 32767)
    return set_err(err, "NIfTI-1 dimensions are 16-bit: 1..32767 per axis");
  Nifti1Header h;
  memset(&h, 0, sizeof(h));
  h.sizeof_hdr = 348;
  h.regular = 'r';
  h.dim[0] = 3; h.dim[1] = (int16_t)a.nx; h.dim[2] = (int16_t)a.ny; h.dim[3] = (int16_t)a.nz;
  h.dim[4] = h.dim[5] = h.dim[6] = h.dim[7] = 1;
  h.datatype = 16; h.bitpix = 32;
  h.vox_offset = 352;
  h.scl_slope = 1; h.scl_inter = 0;
  h.xyzt_units = 2 | 16;                              // mm, msec (irtkNIFTI.h:148-149)
  double M[16];
  image_to_world(a, M);
  // qform from the image-to-world matrix (irtkNIFTI.h:126-142): column lengths = voxel sizes, qfac = sign of the
  // determinant, quaternion of the remaining proper rotation
  double R[9], len[3];
  for (int j = 0; j < 3; ++j) {
    len[j] = sqrt(M[j] * M[j] + M[4 + j] * M[4 + j] + M[8 + j] * M[8 + j]);
    if (!(len[j] > 0)) return set_err(err, "degenerate image axes");
    for (int i = 0; i < 3; ++i) R[3 * i + j] = M[4 * i + j] / len[j];
  }
  const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
  double qfac = 1.0;
  if (det < 0) { qfac = -1.0; R[2] = -R[2]; R[5] = -R[5]; R[8] = -R[8]; }
  double b, c, d;
  rotation_to_quatern(R, b, c, d);
  h.qform_code = 1; h.sform_code = 0;
  h.pixdim[0] = (float)qfac; h.pixdim[1] = (float)len[0]; h.pixdim[2] = (float)len[1]; h.pixdim[3] = (float)len[2];
  h.pixdim[4] = 1;
  h.quatern_b = (float)b; h.quatern_c = (float)c; h.quatern_d = (float)d;
  h.qoffset_x = (float)M[3]; h.qoffset_y = (float)M[7]; h.qoffset_z = (float)M[11];
  memcpy(h.magic, "n+1", 4);
  const size_t n = (size_t)a.nx * a.ny * a.nz;
  const char pad[4] = {0, 0, 0, 0};
  const size_t len_path = strlen(path);
  const bool gz = len_path > 3 && !strcmp(path + len_path - 3, ".gz");
  if (gz && n * sizeof(float) >= (size_t)16 << 20) {
    // a large volume (the 0.5 mm cases: 200 MB) is deflated by all host threads: the file is a sequence of gzip members of
    // 4 MiB of payload each (RFC 1952 section 2.2: zlib's gzread, Python's gzip and the NIfTI readers built on them read the
    // members as one stream).  One thread at zlib's default level writes 55 MB/s, i.e. 3.8 s of a 20 s command line.
    std::vector<unsigned char> payload(sizeof(h) + 4 + n * sizeof(float));
    memcpy(payload.data(), &h, sizeof(h));
    memcpy(payload.data() + sizeof(h), pad, 4);
    memcpy(payload.data() + sizeof(h) + 4, data, n * sizeof(float));
    const size_t piece = (size_t)4 << 20, pieces = (payload.size() + piece - 1) / piece;
    std::vector<std::vector<unsigned char>> out(pieces);
    std::atomic<size_t> next{0};
    std::atomic<bool> good{true};
    auto work = [&]() {
      for (size_t i = next.fetch_add(1); i < pieces; i = next.fetch_add(1)) {
        const size_t lo = i * piece, len = std::min(piece, payload.size() - lo);
        z_stream z;
        memset(&z, 0, sizeof(z));
        if (deflateInit2(&z, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) { good = false; return; }
        out[i].resize(deflateBound(&z, (uLong)len) + 64);
        z.next_in = payload.data() + lo; z.avail_in = (uInt)len;
        z.next_out = out[i].data(); z.avail_out = (uInt)out[i].size();
        const int rc = deflate(&z, Z_FINISH);
        out[i].resize(z.total_out);
        deflateEnd(&z);
        if (rc != Z_STREAM_END) { good = false; return; }
      }
    };
    const unsigned nt = std::max(1u, std::min<unsigned>({(unsigned)svr_host_threads(), 32u, (unsigned)pieces}));
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (!good) return set_err(err, "deflate failed");
    FILE *f = fopen(path, "wb");
    if (!f) return set_err(err, std::string("cannot create ") + path);
    bool ok = true;
    for (size_t i = 0; i < pieces && ok; ++i) ok = fwrite(out[i].data(), 1, out[i].size(), f) == out[i].size();
    ok = (fclose(f) == 0) && ok;
    return ok ? SVR_OK : set_err(err, "write failed");
  }
  if (gz) {
    gzFile f = gzopen(path, "wb");
    if (!f) return set_err(err, std::string("cannot create ") + path);
    bool ok = gzwrite(f, &h, sizeof(h)) == (int)sizeof(h) && gzwrite(f, pad, 4) == 4;
    size_t done = 0;
    while (ok && done < n * sizeof(float)) {
      const unsigned chunk = (unsigned)std::min<size_t>(n * sizeof(float) - done, 1u << 30);
      ok = gzwrite(f, reinterpret_cast<const char *>(data) + done, chunk) == (int)chunk;
      done += chunk;
    }
    ok = (gzclose(f) == Z_OK) && ok;
    return ok ? SVR_OK : set_err(err, "write failed");
  }
  FILE *f = fopen(path, "wb");
  if (!f) return set_err(err, std::string("cannot create ") + path);
  bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(pad, 1, 4, f) == 4 && fwrite(data, sizeof(float), n, f) == n;
  ok = (fclose(f) == 0) && ok;
  return ok ? SVR_OK : set_err(err, "write failed");
}

int svr_host_threads(void) {
  static const int cached = [] {
    if (const char *e = getenv("SVR_HOST_THREADS")) return std::max(1, atoi(e));
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n < 1) n = 1;
    double quota = -1, period = -1;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                       // cgroup v2: "<quota|max> <period>"
      char q[64] = {0};
      if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0) quota = atof(q);
      fclose(f);
    } else {
      if (FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lf", &quota) != 1) quota = -1; fclose(fq); }
      if (FILE *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lf", &period) != 1) period = -1; fclose(fp); }
    }
    if (quota > 0 && period > 0) n = std::min(n, std::max(1, (int)(quota / period + 0.5)));
    return n;
  }();
  return cached;
}

// IRTK rigid `dof` files (irtkRigidTransformation::Read / Write, IRTKSimple2/packages/transformation/src/
// irtkRigidTransformation.cc:392-451): big-endian (irtkCifstream.cc:18-22), through zlib: uint32 magic 815007,
// uint32 type 2 (rigid), uint32 dofs 6, then tx ty tz rx ry rz as doubles (mm, degrees).  matrix16 = the
// homogeneous matrix of irtkRigidTransformation::UpdateMatrix (irtkRigidTransformation.cc:26-53), row-major.
int svr_dof_read(const char *path, double params6[6], double matrix16[16], char err[256]) {
  if (!path || !params6) return SVR_E_ARG;
  gzFile f = gzopen(path, "rb");
  if (!f) return set_err(err, std::string("cannot open ") + path);
  unsigned char buf[12 + 48];
  const int got = gzread(f, buf, sizeof(buf));
  gzclose(f);
  if (got != (int)sizeof(buf)) return set_err(err, "short dof file");
  auto be32 = [&](int o) { return ((uint32_t)buf[o] << 24) | ((uint32_t)buf[o + 1] << 16) | ((uint32_t)buf[o + 2] << 8) | buf[o + 3]; };
  if (be32(0) != 815007u) return set_err(err, "not an IRTK transformation file (magic != 815007)");
  if (be32(4) != 2u || be32(8) != 6u) return set_err(err, "only rigid transformations (type 2, 6 dofs) are supported");
  for (int i = 0; i < 6; ++i) {
    unsigned char t[8];
    for (int k = 0; k < 8; ++k) t[k] = buf[12 + 8 * i + 7 - k];
    memcpy(&params6[i], t, 8);
  }
  if (matrix16) {
    const double k = M_PI / 180.0;
    const double cx = cos(params6[3] * k), cy = cos(params6[4] * k), cz = cos(params6[5] * k);
    const double sx = sin(params6[3] * k), sy = sin(params6[4] * k), sz = sin(params6[5] * k);
    const double m[16] = {cy * cz, cy * sz, -sy, params6[0],
                          sx * sy * cz - cx * sz, sx * sy * sz + cx * cz, sx * cy, params6[1],
                          cx * sy * cz + sx * sz, cx * sy * sz - sx * cz, cx * cy, params6[2],
                          0, 0, 0, 1};
    memcpy(matrix16, m, sizeof(m));
  }
  return SVR_OK;
}

int svr_dof_write(const char *path, const double params6[6], char err[256]) {
  if (!path || !params6) return SVR_E_ARG;
  unsigned char buf[12 + 48];
  const uint32_t head[3] = {815007u, 2u, 6u};
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 4; ++k) buf[4 * i + k] = (unsigned char)(head[i] >> (24 - 8 * k));
  for (int i = 0; i < 6; ++i) {
    unsigned char t[8];
    memcpy(t, &params6[i], 8);
    for (int k = 0; k < 8; ++k) buf[12 + 8 * i + k] = t[7 - k];
  }
  FILE *f = fopen(path, "wb");
  if (!f) return set_err(err, std::string("cannot create ") + path);
  const bool ok = fwrite(buf, 1, sizeof(buf), f) == sizeof(buf);
  return (fclose(f) == 0 && ok) ? SVR_OK : set_err(err, "write failed");
}

void svr_free(void *p) { free(p); }

}  // extern "C"
