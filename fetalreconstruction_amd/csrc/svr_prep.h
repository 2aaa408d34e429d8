// svr_prep.h -- host-side pre-processing shared by the two command lines (csrc/svr_cli.cpp, csrc/pvr_cli.cpp):
// image geometry (irtkBaseImage), NIfTI reading, and the irtkReconstruction pre-processing steps in C++
// (irtkReconstructionGPU.cc = "RG.cc"; the line ranges are on each function).  The Python mirror is
// fetalreconstruction_amd/preprocess.py, which the tests compare these with.  Included by one translation unit
// per program, hence the unnamed namespace.
#ifndef SVR_PREP_H
#define SVR_PREP_H
#include <dirent.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <fstream>
#include <chrono>
#include <atomic>
#include <functional>
#include <mutex>
#include <unistd.h>
#include <string>
#include <thread>
#include <vector>

#include "../../include/svr_host.h"

namespace {

struct M4 { double m[16]; };
M4 ident() { M4 c; for (int i = 0; i < 16; ++i) c.m[i] = (i % 5 == 0) ? 1.0 : 0.0; return c; }
M4 mul(const M4 &a, const M4 &b) {
  M4 c;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double t = 0;
      for (int k = 0; k < 4; ++k) t += a.m[4 * i + k] * b.m[4 * k + j];
      c.m[4 * i + j] = t;
    }
  return c;
}
M4 image_to_world(const svr_image_attr &a) {          // irtkBaseImage.cc:79-111
  M4 t1 = ident(), sc = ident(), rot = ident(), t2 = ident();
  t1.m[3] = -(a.nx - 1) / 2.0; t1.m[7] = -(a.ny - 1) / 2.0; t1.m[11] = -(a.nz - 1) / 2.0;
  sc.m[0] = a.dx; sc.m[5] = a.dy; sc.m[10] = a.dz;
  for (int k = 0; k < 3; ++k) { rot.m[4 * k] = a.xaxis[k]; rot.m[4 * k + 1] = a.yaxis[k]; rot.m[4 * k + 2] = a.zaxis[k]; }
  for (int k = 0; k < 3; ++k) t2.m[4 * k + 3] = a.origin[k];
  return mul(t2, mul(rot, mul(sc, t1)));
}
M4 world_to_image(const svr_image_attr &a) {          // irtkBaseImage.cc:113-147
  M4 t1 = ident(), rot = ident(), sc = ident(), t2 = ident();
  for (int k = 0; k < 3; ++k) t1.m[4 * k + 3] = -a.origin[k];
  for (int k = 0; k < 3; ++k) { rot.m[k] = a.xaxis[k]; rot.m[4 + k] = a.yaxis[k]; rot.m[8 + k] = a.zaxis[k]; }
  sc.m[0] = 1.0 / a.dx; sc.m[5] = 1.0 / a.dy; sc.m[10] = 1.0 / a.dz;
  t2.m[3] = (a.nx - 1) / 2.0; t2.m[7] = (a.ny - 1) / 2.0; t2.m[11] = (a.nz - 1) / 2.0;
  return mul(t2, mul(sc, mul(rot, t1)));
}
M4 inverse_rigid_or_affine(const M4 &a) {             // Gauss-Jordan on the 4x4
  double w[4][8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) { w[i][j] = a.m[4 * i + j]; w[i][4 + j] = (i == j) ? 1.0 : 0.0; }
  for (int c = 0; c < 4; ++c) {
    int p = c;
    for (int r = c + 1; r < 4; ++r) if (fabs(w[r][c]) > fabs(w[p][c])) p = r;
    for (int j = 0; j < 8; ++j) std::swap(w[c][j], w[p][j]);
    const double d = w[c][c];
    for (int j = 0; j < 8; ++j) w[c][j] /= d;
    for (int r = 0; r < 4; ++r)
      if (r != c) { const double f = w[r][c]; for (int j = 0; j < 8; ++j) w[r][j] -= f * w[c][j]; }
  }
  M4 o;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) o.m[4 * i + j] = w[i][4 + j];
  return o;
}
double irtk_round(double x) { return x > 0 ? floor(x + 0.5) : ceil(x - 0.5); }   // irtkCommon.h:85-88

struct Image {
  svr_image_attr a;
  std::vector<double> d;                              // [z][y][x]
  double &at(int x, int y, int z) { return d[((size_t)z * a.ny + y) * a.nx + x]; }
  double at(int x, int y, int z) const { return d[((size_t)z * a.ny + y) * a.nx + x]; }
};

const char *prog_name = "SVRreconstructionGPU";
std::function<void()> before_exit;     // main thread only, e.g. wait for a context that another thread is still creating
// die() is reachable from the main thread, from pool workers (read_image, match_stack_intensities inside parallel_for) and
// from the per-device rank threads, possibly from several at once.  The first caller wins (the others wait on the mutex
// until the process is gone) and the process ends without running static destructors: peers of a failing rank sit in an
// RCCL collective or a barrier that will never complete, and the pool's destructor would join the very worker that called.
std::mutex die_mutex;
const std::thread::id main_thread_id = std::this_thread::get_id();
[[noreturn]] void die(const std::string &m) {
  std::lock_guard<std::mutex> first(die_mutex);
  fprintf(stderr, "%s: %s\n", prog_name, m.c_str());
  if (std::this_thread::get_id() == main_thread_id && before_exit) { auto f = before_exit; before_exit = nullptr; f(); }
  fflush(nullptr);
  _exit(1);
}

// SVR_CLI_TIMING=1: wall time of every stage of a command line on stderr (tools/run_cli_*.py read it)
struct StageClock {
  bool on = getenv("SVR_CLI_TIMING") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
  void mark(const char *what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[timing] %-34s %8.3f s  (at %7.3f s)\n", what, std::chrono::duration<double>(now - last).count(),
            std::chrono::duration<double>(now - t0).count());
    last = now;
  }
};

Image read_image(const std::string &path) {
  Image im;
  float *data = nullptr;
  int nt = 1;
  char err[256] = {0};
  if (svr_nifti_read(path.c_str(), &im.a, &nt, &data, err)) die(path + ": " + err);
  if (nt != 1) die(path + ": 3-D image expected");
  const size_t n = (size_t)im.a.nx * im.a.ny * im.a.nz;
  im.d.assign(data, data + n);
  svr_free(data);
  return im;
}

// CreateTemplate RG.cc:648-694 (+ irtkResampling::Initialize, irtkResampling.cc:74-130)
svr_image_attr create_template(const svr_image_attr &stack, double &resolution) {
  svr_image_attr a = stack;
  a.nz += 2;
  double d = resolution;
  if (resolution <= 0) d = (a.dx <= a.dy && a.dx <= a.dz) ? a.dx : (a.dy <= a.dz ? a.dy : a.dz);
  int n[3] = {(int)(a.nx * a.dx / d), (int)(a.ny * a.dy / d), (int)(a.nz * a.dz / d)};
  double s[3] = {d, d, d};
  const double old[3] = {a.dx, a.dy, a.dz};
  for (int k = 0; k < 3; ++k) if (n[k] < 1) { n[k] = 1; s[k] = old[k]; }
  a.nx = n[0]; a.ny = n[1]; a.nz = n[2]; a.dx = s[0]; a.dy = s[1]; a.dz = s[2];
  resolution = d;
  return a;
}

// irtkGaussianBlurring<irtkRealPixel>(sigma).Run() (irtkGaussianBlurring.cc:40-125, irtkConvolution_1D.cc:42-90)
// fn(i) for i in [0, n) on the host threads (svr_host_threads(), at most 32), items handed out one at a time; every item writes its own output,
// so the results do not depend on the thread count.  The grids of the fine cases have 10^7..10^8 voxels.
inline void parallel_for(int n, const std::function<void(int)> &fn) {
  const unsigned nt = std::max(1u, std::min<unsigned>({(unsigned)svr_host_threads(), 32u, (unsigned)std::max(n, 1)}));
  if (nt < 2 || n < 2) { for (int i = 0; i < n; ++i) fn(i); return; }
  std::atomic<int> next{0};
  auto work = [&]() { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i); };
  std::vector<std::thread> th;
  for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
  work();
  for (auto &t : th) t.join();
}

void gaussian_blur(Image &im, double sigma) {
  const svr_image_attr &a = im.a;
  const int n[3] = {a.nx, a.ny, a.nz};
  const double vs[3] = {a.dx, a.dy, a.dz};
  const size_t stride[3] = {1, (size_t)a.nx, (size_t)a.nx * a.ny};
  for (int axis = 0; axis < 3; ++axis) {
    if (axis == 2 && a.nz == 1) continue;
    const double s = sigma / vs[axis];
    const int half = (int)irtk_round(4 * sigma / vs[axis]);
    std::vector<double> k(2 * half + 1);
    for (int t = -half; t <= half; ++t) k[t + half] = exp(-(double)(t * t) / (2.0 * s * s));
    std::vector<double> out(im.d.size());
    parallel_for(a.nz, [&](int z) {
      for (int y = 0; y < a.ny; ++y)
        for (int x = 0; x < a.nx; ++x) {
          const int p[3] = {x, y, z};
          const size_t base = ((size_t)z * a.ny + y) * a.nx + x;
          double val = 0, sum = 0;
          for (int t = -half; t <= half; ++t) {
            const int q = p[axis] + t;
            if (q < 0 || q >= n[axis]) continue;
            val += k[t + half] * im.d[base + (ptrdiff_t)t * (ptrdiff_t)stride[axis]];
            sum += k[t + half];
          }
          out[base] = sum > 0 ? val / sum : 0.0;
        }
    });
    im.d.swap(out);
  }
}

// One point through a chain of matrices, ONE APPLICATION AT A TIME like the reference (ImageToWorld, Transform, WorldToImage are
// three calls there): composing the matrices first changes the last bits, and a coordinate that is x.5 in exact arithmetic
// then rounds to the other voxel (found by oracle/prep_oracle.c on the bundled mask's grid).
inline void apply_point(const M4 &m, double &x, double &y, double &z) {
  const double a = m.m[0] * x + m.m[1] * y + m.m[2] * z + m.m[3], b = m.m[4] * x + m.m[5] * y + m.m[6] * z + m.m[7],
               c = m.m[8] * x + m.m[9] * y + m.m[10] * z + m.m[11];
  x = a; y = b; z = c;
}

// irtkImageTransformation + nearest neighbour, target padding -1 on an all-zero target (RG.cc:782-793, 808-819)
Image transform_nn(const Image &src, const svr_image_attr &target, const M4 &t, double source_padding) {
  Image out;
  out.a = target;
  out.d.assign((size_t)target.nx * target.ny * target.nz, source_padding);
  const M4 t_i2w = image_to_world(target), s_w2i = world_to_image(src.a);
  parallel_for(target.nz, [&](int z) {
    for (int y = 0; y < target.ny; ++y)
      for (int x = 0; x < target.nx; ++x) {
        double q[3] = {(double)x, (double)y, (double)z};
        apply_point(t_i2w, q[0], q[1], q[2]); apply_point(t, q[0], q[1], q[2]); apply_point(s_w2i, q[0], q[1], q[2]);
        const long i = (long)irtk_round(q[0]), j = (long)irtk_round(q[1]), k = (long)irtk_round(q[2]);
        if (i >= 0 && i < src.a.nx && j >= 0 && j < src.a.ny && k >= 0 && k < src.a.nz)
          out.at(x, y, z) = src.at((int)i, (int)j, (int)k);
      }
  });
  return out;
}

// SetMask RG.cc:750-803
Image set_mask(const svr_image_attr &tmpl, const Image *mask, double sigma, double threshold = 0.5) {
  if (!mask) {
    Image o;
    o.a = tmpl;
    o.d.assign((size_t)tmpl.nx * tmpl.ny * tmpl.nz, 1.0);
    return o;
  }
  Image m = *mask;
  if (sigma > 0) {
    gaussian_blur(m, sigma);
    for (double &v : m.d) v = v > threshold ? 1.0 : 0.0;
  }
  return transform_nn(m, tmpl, ident(), 0.0);
}

// the regular entries of a directory, by name
std::vector<std::string> list_directory(const std::string &dir) {
  std::vector<std::string> out;
  DIR *d = opendir(dir.c_str());
  if (!d) die(dir + " does not exist");
  while (const dirent *e = readdir(d)) {
    const std::string name = e->d_name;
    if (name != "." && name != "..") out.push_back(name);
  }
  closedir(d);
  std::sort(out.begin(), out.end());
  return out;
}

// The origin irtkGenericImage::GetRegion gives the sub-image starting at voxel (i1, j1, k1) (irtkGenericImage.cc:570-611):
// the world position of that voxel minus the position of voxel (0, 0, 0) of the region grid with its origin at zero.  Kept
// literal, so that coordinates landing on x.5 after a later WorldToImage round the way the reference's do.
inline void region_origin(const svr_image_attr &src, int i1, int j1, int k1, svr_image_attr &region) {
  svr_image_attr z = region;
  z.origin[0] = z.origin[1] = z.origin[2] = 0;
  const M4 a = image_to_world(src), b = image_to_world(z);
  for (int k = 0; k < 3; ++k)
    region.origin[k] = (a.m[4 * k] * (double)i1 + a.m[4 * k + 1] * (double)j1 + a.m[4 * k + 2] * (double)k1 + a.m[4 * k + 3]) -
                       (b.m[4 * k] * 0.0 + b.m[4 * k + 1] * 0.0 + b.m[4 * k + 2] * 0.0 + b.m[4 * k + 3]);
}

// irtkGenericImage::GetRegion(i1, j1, k1, i2, j2, k2)
Image get_region(const Image &im, int x1, int y1, int z1, int x2, int y2, int z2) {
  Image o;
  o.a = im.a;
  o.a.nx = x2 - x1; o.a.ny = y2 - y1; o.a.nz = z2 - z1;
  region_origin(im.a, x1, y1, z1, o.a);
  o.d.resize((size_t)o.a.nx * o.a.ny * o.a.nz);
  for (int z = z1; z < z2; ++z)
    for (int y = y1; y < y2; ++y)
      for (int x = x1; x < x2; ++x) o.at(x - x1, y - y1, z - z1) = im.at(x, y, z);
  return o;
}

// CropImage RG.cc:5205-5306
Image crop_image(const Image &im, const Image &mask) {
  int lo[3] = {im.a.nx, im.a.ny, im.a.nz}, hi[3] = {-1, -1, -1};
  for (int z = 0; z < im.a.nz; ++z)
    for (int y = 0; y < im.a.ny; ++y)
      for (int x = 0; x < im.a.nx; ++x)
        if (mask.at(x, y, z) > 0) {
          const int p[3] = {x, y, z};
          for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); }
        }
  if (hi[0] < 0) die("CropImage: the mask does not overlap a stack");
  return get_region(im, lo[0], lo[1], lo[2], hi[0] + 1, hi[1] + 1, hi[2] + 1);
}

// MatchStackIntensitiesWithMasking RG.cc:1375-1493
std::vector<float> match_stack_intensities(std::vector<Image> &stacks, const std::vector<M4> &ts, const Image &mask,
                                           double average_value, bool together) {
  const M4 mw2i = world_to_image(mask.a);
  std::vector<double> avg(stacks.size());
  parallel_for((int)stacks.size(), [&](int s) {             // a stack per host thread; the sums of a stack keep the reference's order
    const Image &st = stacks[s];
    const M4 s_i2w = image_to_world(st.a);
    double sum = 0, num = 0;
    for (int z = 0; z < st.a.nz; ++z)
      for (int y = 0; y < st.a.ny; ++y)
        for (int x = 0; x < st.a.nx; ++x) {
          double qx = x, qy = y, qz = z;
          apply_point(s_i2w, qx, qy, qz); apply_point(ts[s], qx, qy, qz); apply_point(mw2i, qx, qy, qz);
          const long i = (long)irtk_round(qx), j = (long)irtk_round(qy), k = (long)irtk_round(qz);
          if (i >= 0 && i < mask.a.nx && j >= 0 && j < mask.a.ny && k >= 0 && k < mask.a.nz && mask.at((int)i, (int)j, (int)k) == 1) {
            sum += st.at(x, y, z);
            num += 1;
          }
        }
    if (!(num > 0)) die("a stack has no overlap with the ROI");
    avg[s] = sum / num;
  });
  double glob = 0;
  for (double v : avg) glob += v;
  glob /= (double)avg.size();
  std::vector<float> factors;
  for (size_t s = 0; s < stacks.size(); ++s) {
    const double f = average_value / (together ? glob : avg[s]);
    factors.push_back((float)f);
    for (double &v : stacks[s].d) if (v > 0) v *= f;
  }
  return factors;
}

M4 load_transformation(const std::string &spec) {
  if (spec == "id") return ident();
  M4 m;
  double p6[6];
  char err[256];
  if (svr_dof_read(spec.c_str(), p6, m.m, err) == 0) return m;          // IRTK rigid dof file
  std::ifstream f(spec.c_str());
  for (int i = 0; i < 16; ++i)
    if (!(f >> m.m[i])) die("transformation " + spec + ": expected 'id', an IRTK rigid dof file or a 4x4 text matrix");
  return m;
}

void to_f16(const M4 &m, float *out) { for (int i = 0; i < 16; ++i) out[i] = (float)m.m[i]; }

#define ENG(call) do { int rc_ = (call); if (rc_) die(std::string(#call) + ": " + svr_last_error(ctx)); } while (0)
#define HOST(call) do { int rc_ = (call); if (rc_) die(std::string(#call) + ": " + svrh_last_error(host)); } while (0)

}  // namespace

#endif  // SVR_PREP_H
