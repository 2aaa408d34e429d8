// irtk_reg.cpp -- the reference's default (CPU / IRTK) rigid registration schedule around the batched NCC cost of the
// engine (SURVEY 8a16 + 8f1 "the IRTK schedule" + 8f2 "stack-to-stack registration"), host C++.
//
// Restated, each function citing the lines it follows (IRTK = source/IRTKSimple2, RG.cc = source/reconstructionGPU2/
// irtkReconstructionGPU.cc):
//   irtkImageRigidRegistrationWithPadding::GuessParameterThickSlices / GuessParameterSliceToVolume
//                                             IRTK/packages/registration/src/irtkImageRigidRegistrationWithPadding.cc:110-205, 304-400
//   irtkImageRegistrationWithPadding::Initialize(level)      .../src/irtkImageRegistrationWithPadding.cc:28-334
//   irtkImageRegistration::Run, EvaluateGradient             .../src/irtkImageRegistration.cc:414-568
//   irtkGradientDescentOptimizer::Run, irtkOptimizer::Run    .../src/irtkGradientDescentOptimizer.cc:25-69, irtkOptimizer.cc:107-131
//   irtkGaussianBlurringWithPadding, irtkConvolutionWithPadding_1D, irtkScalarGaussian
//                                             IRTK/image++/src/irtkGaussianBlurringWithPadding.cc:35-121, irtkConvolutionWithPadding_1D.cc:38-125
//   irtkResamplingWithPadding                 IRTK/image++/src/irtkResamplingWithPadding.cc:202-443
//   irtkRigidTransformation::UpdateMatrix / Matrix2Parameters   IRTK/packages/transformation/src/irtkRigidTransformation.cc:26-53, 119-148
//   irtkReconstruction::StackRegistrations (+ ParallelStackRegistrations, ResetOrigin, InvertStackTransformations)   RG.cc:823-1001
//   irtkReconstruction::SliceToVolumeRegistration (+ ParallelSliceToVolumeRegistration)                               RG.cc:1991-2059, 2291-2303
// The similarity itself (irtkImageRigidRegistrationWithPadding::Evaluate + irtkCrossCorrelationSimilarityMetric) is the
// engine's svr_ncc_evaluate: exact integer moments, so the accept / reject decisions of the optimiser do not depend on
// where the cost is evaluated.  All targets that share a source are optimised in lock step: one optimiser step of every
// target (its similarity, its 12 finite differences, or one line-search probe) is ONE batched evaluate call.  A 3-D target
// (a stack) is handed to the engine as its z-planes, each with the start position the reference's iterator reaches by
// repeated addition (irtkHomogeneousTransformationIterator.h:96-199), and the six moments are added up.
#include <float.h>
#include <stdint.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <map>
#include <tuple>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "svr_prep.h"

namespace {

// Rows of an image, and the targets of a registration pass, are independent in every filter below: they are split over a
// pool of host threads that lives as long as the library (starting threads per call cost more than the 280 small pyramids
// of a slice-to-volume pass).  Results do not depend on the split.  A call from inside a pool task, or while another
// host thread (an in-process rank) owns the pool, runs inline.
class WorkPool {
 public:
  static WorkPool &get() { static WorkPool p; return p; }
  template <class F> void run(int n, int max_threads, F &fn) {
    // (a forked child inherits the object but not the threads: it works inline)
    if (n < 2 || max_threads < 2 || inside() || workers_->empty() || getpid() != pid_ || !owner_.try_lock()) { fn(0, n); return; }
    const int nt = std::min<int>({max_threads, (int)workers_->size() + 1, n});
    {
      std::lock_guard<std::mutex> g(m_);
      call_ = [](void *f, int a, int b) { (*static_cast<F *>(f))(a, b); };
      fn_ = &fn; n_ = n; chunk_ = std::max(1, n / (nt * 4)); next_ = 0; wanted_ = nt - 1; running_ = nt - 1; ++gen_;
    }
    cv_.notify_all();
    inside() = true;
    work();
    inside() = false;
    std::unique_lock<std::mutex> g(m_);
    done_.wait(g, [&] { return running_ == 0; });
    g.unlock();
    owner_.unlock();
  }
  int size() const { return (int)workers_->size() + 1; }

 private:
  WorkPool() {
    const int n = std::min(svr_host_threads(), 128);          // the CPUs this process may actually use (affinity, cgroup quota)
    pid_ = getpid();
    for (int i = 1; i < n; ++i) workers_->emplace_back([this] { loop(); });
  }
  ~WorkPool() {
    if (getpid() != pid_) return;                          // forked child: the thread objects name threads it never had (leaked on purpose)
    { std::lock_guard<std::mutex> g(m_); stop_ = true; }
    cv_.notify_all();
    for (auto &t : *workers_) t.join();
    delete workers_;
  }
  static bool &inside() { static thread_local bool in = false; return in; }
  void work() {
    for (;;) {
      const int a = next_.fetch_add(chunk_);
      if (a >= n_) break;
      call_(fn_, a, std::min(n_, a + chunk_));
    }
  }
  void loop() {
    inside() = true;
    uint64_t seen = 0;
    std::unique_lock<std::mutex> g(m_);
    for (;;) {
      cv_.wait(g, [&] { return stop_ || (gen_ != seen && wanted_ > 0); });
      if (stop_) return;
      seen = gen_;
      --wanted_;
      g.unlock();
      work();
      g.lock();
      if (--running_ == 0) done_.notify_one();
    }
  }
  std::vector<std::thread> *workers_ = new std::vector<std::thread>();
  std::mutex m_, owner_;
  std::condition_variable cv_, done_;
  void (*call_)(void *, int, int) = nullptr;
  void *fn_ = nullptr;
  int n_ = 0, chunk_ = 1, wanted_ = 0, running_ = 0;
  std::atomic<int> next_{0};
  uint64_t gen_ = 0;
  bool stop_ = false;
  pid_t pid_ = 0;
};
template <class F> void parallel_rows(int n, size_t work_per_row, F fn) {
  const int nt = (int)std::min<size_t>(128, (size_t)n * work_per_row / 50000 + 1);
  WorkPool::get().run(n, nt, fn);
}

template <class T> struct Vol {
  svr_image_attr a;
  std::vector<T> d;                                      // [z][y][x]
  T &at(int x, int y, int z) { return d[((size_t)z * a.ny + y) * a.nx + x]; }
  T at(int x, int y, int z) const { return d[((size_t)z * a.ny + y) * a.nx + x]; }
  size_t n() const { return (size_t)a.nx * a.ny * a.nz; }
};

template <class T> T put_as_double(double v);             // irtkGenericImage::PutAsDouble, irtkGenericImage.h:303-333
template <> short put_as_double<short>(double v) {
  if (v > 32767.0) v = 32767.0;
  if (v < -32768.0) v = -32768.0;
  return (short)v;                                        // static_cast: truncation
}
template <> double put_as_double<double>(double v) { return v; }

// irtkResamplingWithPadding<T>(rx, ry, rz, pad).Run(), RWP.cc:202-443
// the grid irtkResamplingWithPadding gives an image (RWP.cc:202-240)
svr_image_attr resampled_attr(const svr_image_attr &in, double rx, double ry, double rz) {
  svr_image_attr out = in;
  const double want[3] = {rx, ry, rz}, old[3] = {in.dx, in.dy, in.dz};
  const int n_old[3] = {in.nx, in.ny, in.nz};
  int n_new[3];
  double d_new[3];
  for (int k = 0; k < 3; ++k) {
    n_new[k] = (int)irtk_round(n_old[k] * old[k] / want[k]);
    d_new[k] = want[k];
    if (n_new[k] < 1) { n_new[k] = 1; d_new[k] = old[k]; }
  }
  out.nx = n_new[0]; out.ny = n_new[1]; out.nz = n_new[2];
  out.dx = d_new[0]; out.dy = d_new[1]; out.dz = d_new[2];
  return out;
}

template <class T> Vol<T> resample_with_padding(const Vol<T> &in, double rx, double ry, double rz, T pad) {
  Vol<T> out;
  out.a = resampled_attr(in.a, rx, ry, rz);
  out.d.assign(out.n(), pad);
  const M4 o_i2w = image_to_world(out.a), i_w2i = world_to_image(in.a);
  const int X = in.a.nx, Y = in.a.ny, Z = in.a.nz;
  parallel_rows(out.a.nz * out.a.ny, (size_t)out.a.nx * 40, [&](int r0, int r1) {
  for (int r = r0; r < r1; ++r) {
      const int k = r / out.a.ny, j = r % out.a.ny;
      for (int i = 0; i < out.a.nx; ++i) {
        // ImageToWorld then WorldToImage, two separate matrix applications like the reference
        const double wx = o_i2w.m[0] * i + o_i2w.m[1] * j + o_i2w.m[2] * k + o_i2w.m[3];
        const double wy = o_i2w.m[4] * i + o_i2w.m[5] * j + o_i2w.m[6] * k + o_i2w.m[7];
        const double wz = o_i2w.m[8] * i + o_i2w.m[9] * j + o_i2w.m[10] * k + o_i2w.m[11];
        const double x = i_w2i.m[0] * wx + i_w2i.m[1] * wy + i_w2i.m[2] * wz + i_w2i.m[3];
        const double y = i_w2i.m[4] * wx + i_w2i.m[5] * wy + i_w2i.m[6] * wz + i_w2i.m[7];
        const double z = i_w2i.m[8] * wx + i_w2i.m[9] * wy + i_w2i.m[10] * wz + i_w2i.m[11];
        const int u = (int)floor(x), v = (int)floor(y), w = (int)floor(z);
        const double dx = x - u, dy = y - v, dz = z - w;
        const double wt[8] = {(1 - dx) * (1 - dy) * (1 - dz), (1 - dx) * (1 - dy) * dz, (1 - dx) * dy * (1 - dz), (1 - dx) * dy * dz,
                              dx * (1 - dy) * (1 - dz),       dx * (1 - dy) * dz,       dx * dy * (1 - dz),       dx * dy * dz};
        static const int off[8][3] = {{0, 0, 0}, {0, 0, 1}, {0, 1, 0}, {0, 1, 1}, {1, 0, 0}, {1, 0, 1}, {1, 1, 0}, {1, 1, 1}};
        double val = 0, sum = 0;
        int npad = 8;
        for (int c = 0; c < 8; ++c) {
          const int a = u + off[c][0], b = v + off[c][1], e = w + off[c][2];
          if (a >= 0 && a < X && b >= 0 && b < Y && e >= 0 && e < Z) {
            const T g = in.at(a, b, e);
            if (g != pad) { npad--; val += g * wt[c]; sum += wt[c]; }
          } else {
            npad--;
          }
        }
        if (npad < 4 && sum > 0) out.at(i, j, k) = put_as_double<T>(val / sum);
      }
  }
  });
  return out;
}

// the sampled Gaussian of one pass of irtkGaussianBlurringWithPadding (GBWP.cc:60-120): sigma in mm, vs = the voxel size along the axis
std::vector<double> blur_kernel(double sigma, double vs) {
  const double s = sigma / vs;
  const int size = 2 * (int)irtk_round(4 * sigma / vs) + 1;
  std::vector<double> ker(size);                          // irtkScalarGaussian(s, 1, 1, 0, 0, 0) sampled at i - (size-1)/2
  const double norm = 1.0 / (sqrt(2.0 * M_PI) * s * sqrt(2.0 * M_PI) * 1 * sqrt(2.0 * M_PI) * 1);
  for (int i = 0; i < size; ++i) {
    const double x = i - (size - 1) / 2.0;
    ker[i] = norm * exp(-(x * x) / (2.0 * s * s) - 0.0 - 0.0);
    if (fabs(ker[i]) < FLT_MIN) ker[i] = 0;               // irtkScalarFunctionToImage.cc:89-92
  }
  return ker;
}

// irtkGaussianBlurringWithPadding<short>(sigma, pad).Run(): three 1-D passes, each rounded (truncated) back to short
void blur_with_padding(Vol<short> &im, double sigma, short pad) {
  const int n[3] = {im.a.nx, im.a.ny, im.a.nz};
  const double vs[3] = {im.a.dx, im.a.dy, im.a.dz};
  const ptrdiff_t stride[3] = {1, (ptrdiff_t)im.a.nx, (ptrdiff_t)im.a.nx * im.a.ny};
  for (int axis = 0; axis < 3; ++axis) {
    if (axis == 2 && im.a.nz == 1) continue;             // GBWP.cc:90
    const std::vector<double> ker = blur_kernel(sigma, vs[axis]);
    const int size = (int)ker.size(), half = size / 2;
    std::vector<short> out(im.d.size());
    parallel_rows(n[2] * n[1], (size_t)n[0] * size, [&](int r0, int r1) {
    for (int r = r0; r < r1; ++r) {
        const int z = r / n[1], y = r % n[1];
        for (int x = 0; x < n[0]; ++x) {
          const int p[3] = {x, y, z};
          const size_t base = ((size_t)z * n[1] + y) * n[0] + x;
          if (im.d[base] <= pad) { out[base] = pad; continue; }        // CWP_1D.cc:45
          double val = 0, sum = 0;
          for (int t = -half; t <= half; ++t) {
            const int q = p[axis] + t;
            if (q < 0 || q >= n[axis]) continue;
            const short g = im.d[base + t * stride[axis]];
            if (g > pad) { val += ker[t + half] * g; sum += ker[t + half]; }
          }
          out[base] = put_as_double<short>(sum > 0 ? val / sum : 0.0);
        }
    }
    });
    im.d.swap(out);
  }
}

// the eight-corner padding guess of GuessParameter*, IRRWP.cc:183-204
short guess_padding(const Vol<short> &v) {
  const int X = v.a.nx - 1, Y = v.a.ny - 1, Z = v.a.nz - 1;
  const short c = v.at(0, 0, 0);
  if (v.at(X, 0, 0) == c && v.at(0, Y, 0) == c && v.at(0, 0, Z) == c && v.at(X, Y, 0) == c && v.at(0, Y, Z) == c && v.at(X, 0, Z) == c &&
      v.at(X, Y, Z) == c)
    return c;
  return (short)-32768;                                    // MIN_GREY
}

struct Schedule {                                          // GuessParameterThickSlices / GuessParameterSliceToVolume
  int levels = 3, iterations[3], steps[3];
  double epsilon = 0.0001, t_blur[3], s_blur[3], t_res[3][3], s_res[3][3], length[3], delta[3];
};

Schedule guess_parameters(const svr_image_attr &t, const svr_image_attr &s, int slice_to_volume) {
  Schedule p;
  double size = t.dy < t.dx ? t.dy : t.dx;
  p.t_blur[0] = size / 2.0;
  p.t_res[0][0] = size; p.t_res[0][1] = size; p.t_res[0][2] = t.dz;
  for (int i = 1; i < 3; ++i) {
    p.t_blur[i] = p.t_blur[i - 1] * 2;
    p.t_res[i][0] = p.t_res[i - 1][0] * 2; p.t_res[i][1] = p.t_res[i - 1][1] * 2; p.t_res[i][2] = p.t_res[i - 1][2];
  }
  size = s.dy < s.dx ? s.dy : s.dx;
  if (slice_to_volume && s.dz < size) size = s.dz;         // IRRWP.cc:355-356
  p.s_blur[0] = size / 2.0;
  p.s_res[0][0] = size; p.s_res[0][1] = size; p.s_res[0][2] = slice_to_volume ? size : s.dz;
  for (int i = 1; i < 3; ++i) {
    p.s_blur[i] = p.s_blur[i - 1] * 2;
    p.s_res[i][0] = p.s_res[i - 1][0] * 2; p.s_res[i][1] = p.s_res[i - 1][1] * 2;
    p.s_res[i][2] = slice_to_volume ? p.s_res[i - 1][2] * 2 : p.s_res[i - 1][2];
  }
  for (int i = 0; i < 3; ++i) { p.iterations[i] = 20; p.steps[i] = 4; p.length[i] = 2 * pow(2.0, i); p.delta[i] = 0; }
  return p;
}

// irtkImageRegistrationWithPadding::Initialize(level) for one image: blur, resample, shift the range, pad with -1
int prepare_level(const Vol<short> &in, double blur, const double res[3], const double res0[3], int level, short pad, Vol<short> &out,
                  std::string &err) {
  out = in;
  if (blur > 0) blur_with_padding(out, blur, pad);
  const double temp = fabs(res0[0] - out.a.dx) + fabs(res0[1] - out.a.dy) + fabs(res0[2] - out.a.dz);
  if (level > 0 || temp > 0.000001) out = resample_with_padding<short>(out, res[0], res[1], res[2], pad);
  double vmax = -32768.0, vmin = 32767.0;
  for (short &v : out.d) {
    if (v > pad) { if (v > vmax) vmax = v; if (v < vmin) vmin = v; }
    else v = pad;
  }
  if (vmax - vmin > 32767.0) { err = "Initialize: dynamic range of an image is too large"; return 1; }
  for (short &v : out.d) v = v > pad ? (short)(v - (short)vmin) : (short)-1;
  return 0;
}

// irtkRigidTransformation::UpdateMatrix, RT.cc:26-53
M4 params_to_matrix(const double p[6]) {
  const double k = M_PI / 180.0;
  const double cosrx = cos(p[3] * k), cosry = cos(p[4] * k), cosrz = cos(p[5] * k);
  const double sinrx = sin(p[3] * k), sinry = sin(p[4] * k), sinrz = sin(p[5] * k);
  M4 m = ident();
  m.m[0] = cosry * cosrz; m.m[1] = cosry * sinrz; m.m[2] = -sinry; m.m[3] = p[0];
  m.m[4] = (sinrx * sinry * cosrz - cosrx * sinrz); m.m[5] = (sinrx * sinry * sinrz + cosrx * cosrz); m.m[6] = sinrx * cosry; m.m[7] = p[1];
  m.m[8] = (cosrx * sinry * cosrz + sinrx * sinrz); m.m[9] = (cosrx * sinry * sinrz - sinrx * cosrz); m.m[10] = cosrx * cosry; m.m[11] = p[2];
  return m;
}

// irtkRigidTransformation::Matrix2Parameters, RT.cc:119-148
void matrix_to_params(const M4 &m, double p[6]) {
  const double TOL = 0.000001;
  p[0] = m.m[3]; p[1] = m.m[7]; p[2] = m.m[11];
  const double tmp = asin(-1 * m.m[2]);
  if (fabs(cos(tmp)) > TOL) {
    p[3] = atan2(m.m[6], m.m[10]);
    p[4] = tmp;
    p[5] = atan2(m.m[1], m.m[0]);
  } else {
    p[3] = atan2(-1.0 * m.m[2] * m.m[4], -1.0 * m.m[2] * m.m[8]);
    p[4] = tmp;
    p[5] = 0;
  }
  p[3] *= 180.0 / M_PI; p[4] *= 180.0 / M_PI; p[5] *= 180.0 / M_PI;
}

struct Backend {                                           // the engine, or whatever the caller supplies
  svr_ctx *ctx;
  const svr_ncc_backend *be;
  int set_targets(int n, int tx, int ty, const int16_t *t) const {
    return be ? be->set_targets(be->user, n, tx, ty, t) : svr_ncc_set_targets(ctx, n, tx, ty, t);
  }
  int set_source(const uint32_t size[3], const int16_t *s) const {
    return be ? be->set_source(be->user, size, s) : svr_ncc_set_source(ctx, size, s);
  }
  int evaluate(int n, const int *idx, const double *m, int64_t *sums) const {
    return be ? be->evaluate(be->user, n, idx, m, sums, nullptr) : svr_ncc_evaluate(ctx, n, idx, m, sums, nullptr);
  }
};

struct Target {                                            // one registration: a target image and its transformation
  const Vol<short> *full;                                  // the unprocessed target
  Vol<short> lvl;                                          // this level's image
  int first_plane = 0;                                     // index of its plane 0 among the backend's targets
  int own_padding = 0;                                     // 1: `padding` replaces the call's target padding (GuessParameter's corner rule)
  short padding = 0;
  M4 matrix;                                               // irtkRigidTransformation::_matrix
  double p[6];                                             // ... and its parameters
  // optimiser state (irtkImageRegistration::Run + irtkGradientDescentOptimizer::Run)
  int phase = 0, step_i = 0, iter_j = 0, grad_k = 0;
  double step = 0, delta = 0, old_sim = 0, new_sim = 0, sim = 0, start[6], s1 = 0;
  float dx[6];
  bool done = false;
};

enum { PH_START = 0, PH_GRAD, PH_LINE };

double ncc_from_sums(const int64_t s[6]) {                 // irtkCrossCorrelationSimilarityMetric::Evaluate, CCSM.h:158-165
  const double n = (double)s[0], x = (double)s[1], y = (double)s[2], x2 = (double)s[3], y2 = (double)s[4], xy = (double)s[5];
  if (n > 0) return (xy - (x * y) / n) / (sqrt(x2 - x * x / n) * sqrt(y2 - y * y / n));
  return 0;
}

// One level of prepare_level on the device (csrc/svr_pyr.inc) for `n` images of one grid whose unprocessed voxels sit at
// `offset` of the upload slot: the attributes of the level come back in out[i], the voxels stay on the device as the NCC
// source (slot 0) or as the target planes first_plane, first_plane + nz, ...
int device_level(svr_ctx *ctx, int slot, size_t offset, int n, const svr_image_attr *const *in, const short *pads, double blur, const double res[3],
                 const double res0[3], int level, int first_plane, svr_image_attr *out, std::string &err) {
  const svr_image_attr &a0 = *in[0];
  std::vector<double> ker[3];
  if (blur > 0) {
    ker[0] = blur_kernel(blur, a0.dx); ker[1] = blur_kernel(blur, a0.dy);
    if (a0.nz != 1) ker[2] = blur_kernel(blur, a0.dz);     // GBWP.cc:90
  }
  const double temp = fabs(res0[0] - a0.dx) + fabs(res0[1] - a0.dy) + fabs(res0[2] - a0.dz);
  const int resample = level > 0 || temp > 0.000001;
  std::vector<svr_pyr_image> imgs(n);
  for (int i = 0; i < n; ++i) {
    out[i] = resample ? resampled_attr(*in[i], res[0], res[1], res[2]) : *in[i];
    const M4 o_i2w = image_to_world(out[i]), i_w2i = world_to_image(*in[i]);
    memcpy(imgs[i].i2w_out, o_i2w.m, sizeof(imgs[i].i2w_out));
    memcpy(imgs[i].w2i_in, i_w2i.m, sizeof(imgs[i].w2i_in));
    imgs[i].pad = pads[i];
  }
  const int in_dims[3] = {a0.nx, a0.ny, a0.nz}, out_dims[3] = {out[0].nx, out[0].ny, out[0].nz};
  std::vector<int> mn(n), mx(n);
  if (svr_pyr_level(ctx, slot, offset, n, in_dims, ker[0].data(), (int)ker[0].size(), ker[1].data(), (int)ker[1].size(), ker[2].data(),
                    (int)ker[2].size(), resample, out_dims, imgs.data(), first_plane, mn.data(), mx.data())) {
    err = std::string("svr_pyr_level: ") + svr_last_error(ctx);
    return 2;
  }
  for (int i = 0; i < n; ++i)
    if ((double)mx[i] - (double)mn[i] > 32767.0) { err = "Initialize: dynamic range of an image is too large"; return 1; }
  return 0;
}

struct GridKey {
  int nx, ny, nz;
  double dx, dy, dz;
  bool operator<(const GridKey &o) const { return std::tie(nx, ny, nz, dx, dy, dz) < std::tie(o.nx, o.ny, o.nz, o.dx, o.dy, o.dz); }
};
struct GridGroup { std::vector<int> members; size_t offset = 0; };

// irtkImageRegistration::Run for every target against one source, in lock step
int run_registrations(const Backend &be, std::vector<Target> &targets, const Vol<short> &source, int slice_to_volume, short target_padding,
                      long *n_eval, std::string &err) {
  if (targets.empty()) return 0;
  const short source_padding = guess_padding(source);
  const bool timing = getenv("SVR_REG_TIMING") != nullptr;        // dev: where a registration pass spends its wall time
  double t_src = 0, t_tgt = 0, t_pack = 0, t_opt = 0, t_eval = 0;
  long rounds = 0;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  // The engine makes the pyramids itself (csrc/svr_pyr.inc: the same arithmetic, no host pass over the voxels, no upload
  // per level); another backend (the oracle evaluator of the tests), or SVR_HOST_PYRAMID=1, takes the host code below.
  const bool dev = !be.be && be.ctx && !getenv("SVR_HOST_PYRAMID");
  std::map<GridKey, GridGroup> groups;                    // targets of one grid go through the kernels together
  if (dev) {
    if (svr_pyr_upload(be.ctx, 0, source.d.data(), source.n())) { err = std::string("svr_pyr_upload: ") + svr_last_error(be.ctx); return 2; }
    size_t total = 0;
    for (size_t r = 0; r < targets.size(); ++r) {
      const svr_image_attr &a = targets[r].full->a;
      groups[GridKey{a.nx, a.ny, a.nz, a.dx, a.dy, a.dz}].members.push_back((int)r);
      total += targets[r].full->n();
    }
    std::vector<int16_t> all(total);
    size_t at = 0;
    for (auto &g : groups) {
      g.second.offset = at;
      for (int r : g.second.members) { memcpy(&all[at], targets[r].full->d.data(), sizeof(int16_t) * targets[r].full->n()); at += targets[r].full->n(); }
    }
    if (svr_pyr_upload(be.ctx, 1, all.data(), total)) { err = std::string("svr_pyr_upload: ") + svr_last_error(be.ctx); return 2; }
  }
  for (int level = 2; level >= 0; --level) {
    double t0 = now();
    // ---- Initialize(level): every target and the source -------------------------------------------------------------
    Vol<short> src;
    const Schedule ps = guess_parameters(targets[0].full->a, source.a, slice_to_volume);
    if (dev) {
      const svr_image_attr *in = &source.a;
      if (int rc = device_level(be.ctx, 0, 0, 1, &in, &source_padding, ps.s_blur[level], ps.s_res[level], ps.s_res[0], level, 0, &src.a, err)) return rc;
    } else if (prepare_level(source, ps.s_blur[level], ps.s_res[level], ps.s_res[0], level, source_padding, src, err)) return 1;
    t_src += now() - t0; t0 = now();
    int tx = 0, ty = 0, planes = 0;
    std::vector<int> bad(targets.size(), 0);
    std::vector<std::string> errs(targets.size());
    size_t pixels = 0;
    for (const Target &t : targets) pixels += t.full->n();
    if (dev) {
      // the grids of the level first (the planes must be allocated before the kernels write them), group by group
      for (auto &g : groups)
        for (int r : g.second.members) {
          Target &t = targets[r];
          const Schedule p = guess_parameters(t.full->a, source.a, slice_to_volume);
          const double temp = fabs(p.t_res[0][0] - t.full->a.dx) + fabs(p.t_res[0][1] - t.full->a.dy) + fabs(p.t_res[0][2] - t.full->a.dz);
          t.lvl.a = (level > 0 || temp > 0.000001) ? resampled_attr(t.full->a, p.t_res[level][0], p.t_res[level][1], p.t_res[level][2]) : t.full->a;
          t.lvl.d.clear();
          t.phase = PH_START; t.step_i = 0; t.iter_j = 0; t.done = false;
          t.step = p.length[level]; t.delta = p.delta[level];
          tx = std::max(tx, t.lvl.a.nx); ty = std::max(ty, t.lvl.a.ny);
          t.first_plane = planes;
          planes += t.lvl.a.nz;
        }
      if (svr_ncc_alloc_targets(be.ctx, planes, tx, ty)) { err = std::string("svr_ncc_alloc_targets: ") + svr_last_error(be.ctx); return 2; }
      for (auto &g : groups) {
        const std::vector<int> &mem = g.second.members;
        const Schedule p = guess_parameters(targets[mem[0]].full->a, source.a, slice_to_volume);
        const size_t nvox = targets[mem[0]].full->n();
        for (size_t c0 = 0; c0 < mem.size(); c0 += 32768) {                        // the grid's y dimension holds 65535 images
          const int n = (int)std::min<size_t>(32768, mem.size() - c0);
          std::vector<const svr_image_attr *> in(n);
          std::vector<short> pads(n);
          std::vector<svr_image_attr> outa(n);
          for (int i = 0; i < n; ++i) {
            const Target &t = targets[mem[c0 + i]];
            in[i] = &t.full->a;
            pads[i] = t.own_padding ? t.padding : target_padding;
          }
          if (int rc = device_level(be.ctx, 1, g.second.offset + c0 * nvox, n, in.data(), pads.data(), p.t_blur[level], p.t_res[level], p.t_res[0], level,
                                    targets[mem[c0]].first_plane, outa.data(), err))
            return rc;
        }
      }
    } else
    parallel_rows((int)targets.size(), pixels / targets.size() * 60, [&](int r0, int r1) {
      for (int r = r0; r < r1; ++r) {
        Target &t = targets[r];
        const Schedule p = guess_parameters(t.full->a, source.a, slice_to_volume);
        bad[r] = prepare_level(*t.full, p.t_blur[level], p.t_res[level], p.t_res[0], level, t.own_padding ? t.padding : target_padding, t.lvl,
                               errs[r]);
        t.phase = PH_START; t.step_i = 0; t.iter_j = 0; t.done = false;
        t.step = p.length[level]; t.delta = p.delta[level];
      }
    });
    t_tgt += now() - t0; t0 = now();
    if (!dev) {
      for (size_t r = 0; r < targets.size(); ++r) {
        if (bad[r]) { err = errs[r]; return 1; }
        Target &t = targets[r];
        tx = std::max(tx, t.lvl.a.nx); ty = std::max(ty, t.lvl.a.ny);
        t.first_plane = planes;
        planes += t.lvl.a.nz;
      }
      std::vector<int16_t> packed((size_t)planes * tx * ty, (int16_t)-1);
      for (const Target &t : targets)
        for (int k = 0; k < t.lvl.a.nz; ++k)
          for (int y = 0; y < t.lvl.a.ny; ++y)
            memcpy(&packed[((size_t)(t.first_plane + k) * ty + y) * tx], &t.lvl.d[((size_t)k * t.lvl.a.ny + y) * t.lvl.a.nx],
                   sizeof(int16_t) * t.lvl.a.nx);
      if (be.set_targets(planes, tx, ty, packed.data())) { err = "svr_ncc_set_targets failed"; return 2; }
      const uint32_t ssz[3] = {(uint32_t)src.a.nx, (uint32_t)src.a.ny, (uint32_t)src.a.nz};
      if (be.set_source(ssz, src.d.data())) { err = "svr_ncc_set_source failed"; return 2; }
    }
    const M4 s_w2i = world_to_image(src.a);
    const Schedule p0 = ps;
    t_pack += now() - t0; t0 = now();
    // ---- the optimiser, one round = one batched evaluation --------------------------------------------------------------
    // A round costs the host a few double trigonometric calls per target; with tens of thousands of targets (the patches
    // of the patch-based path) that was 1.6 s of a 2.8 s pass, so the requests are laid out by a prefix sum and written,
    // and the answers consumed, by the host threads -- every target touches only its own state and its own slots.
    std::vector<int> live, req_off, plane_off, idx;
    std::vector<double> mats, value;
    std::vector<int64_t> sums;
    auto over_live = [&](size_t work, const std::function<void(int, int)> &fn) {
      if (live.size() >= 2048) parallel_rows((int)live.size(), work, fn);
      else fn(0, (int)live.size());
    };
    for (;;) {
      live.clear(); req_off.clear(); plane_off.clear();
      int n_req = 0, n_planes = 0;
      for (size_t ti = 0; ti < targets.size(); ++ti) {
        const Target &t = targets[ti];
        if (t.done) continue;
        const int nr = t.phase == PH_GRAD ? 12 : 1;         // START: old_similarity = Evaluate(), GDO.cc:31; GRAD: IR.cc:530-568; LINE: one probe
        live.push_back((int)ti); req_off.push_back(n_req); plane_off.push_back(n_planes);
        n_req += nr; n_planes += nr * t.lvl.a.nz;
      }
      if (live.empty()) break;
      idx.resize(n_planes); mats.resize(16 * (size_t)n_planes); sums.resize(6 * (size_t)n_planes); value.resize(n_req);
      over_live(3000, [&](int l0, int l1) {
        for (int l = l0; l < l1; ++l) {
          const Target &t = targets[live[l]];
          int at = plane_off[l];
          auto request = [&](const M4 &tm) {                 // the per-plane matrices of one evaluation (push_request)
            const M4 m = mul(mul(s_w2i, tm), image_to_world(t.lvl.a));
            double zx = m.m[3], zy = m.m[7], zz = m.m[11];
            for (int k = 0; k < t.lvl.a.nz; ++k, ++at) {
              idx[at] = t.first_plane + k;
              M4 q = m;
              q.m[3] = zx; q.m[7] = zy; q.m[11] = zz;
              memcpy(&mats[16 * (size_t)at], q.m, sizeof(q.m));
              zx += m.m[2]; zy += m.m[6]; zz += m.m[10];     // NextZ()
            }
          };
          if (t.phase == PH_GRAD) {
            for (int i = 0; i < 6; ++i) {
              double q[6];
              for (int k = 0; k < 6; ++k) q[k] = t.p[k];
              q[i] = t.p[i] + t.step; request(params_to_matrix(q));
              q[i] = t.p[i] - t.step; request(params_to_matrix(q));
            }
          } else {
            request(t.matrix);
          }
        }
      });
      const double te = now();
      if (be.evaluate(n_planes, idx.data(), mats.data(), sums.data())) { err = "svr_ncc_evaluate failed"; return 2; }
      t_eval += now() - te; ++rounds;
      if (n_eval) *n_eval += (long)n_req;
      over_live(600, [&](int l0, int l1) {
        for (int l = l0; l < l1; ++l) {
          Target &t = targets[live[l]];
          const int nz = t.lvl.a.nz, nr = t.phase == PH_GRAD ? 12 : 1;
          for (int q = 0; q < nr; ++q) {
            int64_t sacc[6] = {0, 0, 0, 0, 0, 0};
            const size_t at = (size_t)plane_off[l] + (size_t)q * nz;
            for (int c = 0; c < nz; ++c) for (int k = 0; k < 6; ++k) sacc[k] += sums[6 * (at + c) + k];
            value[req_off[l] + q] = ncc_from_sums(sacc);
          }
          size_t r = req_off[l];
          auto take_step = [&](double sign) {                // Put(i, Get(i) +- StepSize * dx[i]) for every i
            for (int i = 0; i < 6; ++i) t.p[i] = t.p[i] + sign * (t.step * t.dx[i]);
            t.matrix = params_to_matrix(t.p);
          };
          if (t.phase == PH_START) {
            t.old_sim = t.new_sim = t.sim = value[r++];
            for (int i = 0; i < 6; ++i) t.start[i] = t.p[i];   // irtkOptimizer::Run stores the parameters, O.cc:113-116
            t.phase = PH_GRAD;
          } else if (t.phase == PH_GRAD) {
            double norm = 0;
            for (int i = 0; i < 6; ++i) { const double s1 = value[r++], s2 = value[r++]; t.dx[i] = (float)(s1 - s2); }
            t.matrix = params_to_matrix(t.p);                  // Put(i, parameterValue) rebuilt the matrix from the parameters
            for (int i = 0; i < 6; ++i) norm += t.dx[i] * t.dx[i];
            norm = sqrt(norm);
            for (int i = 0; i < 6; ++i) t.dx[i] = norm > 0 ? (float)(t.dx[i] / norm) : 0.0f;
            t.new_sim = t.sim;                                 // first pass of the do-while, GDO.cc:47-54
            take_step(+1);
            t.phase = PH_LINE;
          } else {
            t.sim = value[r++];
            if (t.sim > t.new_sim + p0.epsilon) {              // keep stepping
              t.new_sim = t.sim;
              take_step(+1);
            } else {
              take_step(-1);                                   // last step was no improvement: back track, GDO.cc:58-61
              const double eps = t.new_sim > t.old_sim ? t.new_sim - t.old_sim : 0;
              double max_change = 0;
              for (int i = 0; i < 6; ++i) max_change = std::max(max_change, fabs(t.p[i] - t.start[i]));
              bool next_step = true;                           // IR.cc:482-506
              if (eps > p0.epsilon && max_change > t.delta) {
                if (++t.iter_j < p0.iterations[level]) next_step = false;
              }
              if (next_step) {
                t.iter_j = 0;
                t.step = t.step / 2;
                t.delta = t.delta / 2.0;
                if (++t.step_i >= p0.steps[level]) t.done = true;
              }
              t.phase = PH_START;
            }
          }
        }
      });
    }
    t_opt += now() - t0;
  }
  if (timing)
    fprintf(stderr, "[svr reg timing] source pyramid %.1f ms, target pyramids %.1f ms, pack + upload %.1f ms, optimiser %.1f ms (%ld rounds, %.1f ms in the batched evaluations)\n",
            1e3 * t_src, 1e3 * t_tgt, 1e3 * t_pack, 1e3 * t_opt, rounds, 1e3 * t_eval);
  return 0;
}

Vol<short> to_grey(const svr_image_attr &a, const double *d) {           // irtkGreyImage = irtkRealImage: static_cast per voxel
  Vol<short> v;
  v.a = a;
  v.d.resize(v.n());
  for (size_t i = 0; i < v.d.size(); ++i) v.d[i] = (short)d[i];
  return v;
}

void reset_origin(svr_image_attr &a, M4 &offset) {                       // ResetOrigin, RG.cc:823-834
  offset = ident();
  offset.m[3] = a.origin[0]; offset.m[7] = a.origin[1]; offset.m[11] = a.origin[2];
  a.origin[0] = a.origin[1] = a.origin[2] = 0;
}

void set_err(char err[256], const std::string &m) { if (err) { strncpy(err, m.c_str(), 255); err[255] = 0; } }

}  // namespace

extern "C" {

int svrh_stack_registrations(svr_ctx *ctx, const svr_ncc_backend *backend, int n_stacks, const svr_image_attr *attrs,
                             const double *const *stacks, double *transformations, int template_number,
                             const svr_image_attr *mask_attr, const double *mask_or_null, int flags, long *n_evaluations_or_null,
                             char err[256]) {
  if ((!ctx && !backend) || n_stacks < 1 || !attrs || !stacks || !transformations || template_number < 0 || template_number >= n_stacks) {
    set_err(err, "svrh_stack_registrations: bad arguments");
    return 1;
  }
  if (n_evaluations_or_null) *n_evaluations_or_null = 0;
  const Backend be{ctx, backend};
  std::vector<M4> T(n_stacks);
  for (int s = 0; s < n_stacks; ++s) {                                   // InvertStackTransformations, RG.cc:946
    M4 m;
    for (int q = 0; q < 16; ++q) m.m[q] = transformations[16 * s + q];
    T[s] = inverse_rigid_or_affine(m);
  }
  Vol<short> target = to_grey(attrs[template_number], stacks[template_number]);
  if (mask_or_null) {                                                     // RG.cc:959-984: outside the mask -> 0
    Vol<double> mask;
    mask.a = *mask_attr;
    mask.d.assign(mask_or_null, mask_or_null + mask.n());
    const M4 t_i2w = image_to_world(target.a), m_w2i = world_to_image(mask.a);
    for (int k = 0; k < target.a.nz; ++k)
      for (int j = 0; j < target.a.ny; ++j)
        for (int i = 0; i < target.a.nx; ++i) {
          const double wx = t_i2w.m[0] * i + t_i2w.m[1] * j + t_i2w.m[2] * k + t_i2w.m[3], wy = t_i2w.m[4] * i + t_i2w.m[5] * j + t_i2w.m[6] * k + t_i2w.m[7],
                       wz = t_i2w.m[8] * i + t_i2w.m[9] * j + t_i2w.m[10] * k + t_i2w.m[11];
          const double x = irtk_round(m_w2i.m[0] * wx + m_w2i.m[1] * wy + m_w2i.m[2] * wz + m_w2i.m[3]);
          const double y = irtk_round(m_w2i.m[4] * wx + m_w2i.m[5] * wy + m_w2i.m[6] * wz + m_w2i.m[7]);
          const double z = irtk_round(m_w2i.m[8] * wx + m_w2i.m[9] * wy + m_w2i.m[10] * wz + m_w2i.m[11]);
          if (x >= 0 && x < mask.a.nx && y >= 0 && y < mask.a.ny && z >= 0 && z < mask.a.nz) {
            if (mask.at((int)x, (int)y, (int)z) == 0) target.at(i, j, k) = 0;
          } else {
            target.at(i, j, k) = 0;
          }
        }
  }
  M4 mo = ident();
  // irtkStack3D3DRegistration<T>::ResetOrigin takes its arguments by value (irtkStack3D3DRegistration.cpp:151-162): the
  // patch-based command line registers with the target's origin in place and an identity offset
  if (!(flags & SVRH_STACKREG_KEEP_ORIGIN)) reset_origin(target.a, mo);   // RG.cc:987-988
  const M4 mo_inv = inverse_rigid_or_affine(mo);
  for (int s = 0; s < n_stacks; ++s) {                                    // ParallelStackRegistrations, RG.cc:877-911
    if (s == template_number) continue;
    const Vol<short> source = to_grey(attrs[s], stacks[s]);
    std::vector<Target> one(1);
    one[0].full = &target;
    one[0].matrix = mul(T[s], mo);                                        // include the offset: PutMatrix(m * mo)
    matrix_to_params(one[0].matrix, one[0].p);
    std::string e;
    const int rc = run_registrations(be, one, source, 0, (short)0, n_evaluations_or_null, e);   // ThickSlices, SetTargetPadding(0)
    if (rc) { set_err(err, e); return rc; }
    T[s] = mul(params_to_matrix(one[0].p), mo_inv);                       // undo the offset
  }
  for (int s = 0; s < n_stacks; ++s) {                                    // InvertStackTransformations, RG.cc:1000
    const M4 inv = inverse_rigid_or_affine(T[s]);
    for (int q = 0; q < 16; ++q) transformations[16 * s + q] = inv.m[q];
  }
  return 0;
}

int svrh_slice_to_volume_registration(svr_ctx *ctx, const svr_ncc_backend *backend, int n_slices, const float *slices, int sx, int sy,
                                      const svr_image_attr *attrs, double *transformations, const svr_image_attr *recon_attr,
                                      const float *reconstructed, int flags, long *n_evaluations_or_null, char err[256]) {
  if ((!ctx && !backend) || n_slices < 1 || !slices || !attrs || !transformations || !recon_attr || !reconstructed) {
    set_err(err, "svrh_slice_to_volume_registration: bad arguments");
    return 1;
  }
  if (n_evaluations_or_null) *n_evaluations_or_null = 0;
  const Backend be{ctx, backend};
  Vol<short> source;                                                      // irtkGreyImage source = _reconstructed, RG.cc:2031
  source.a = *recon_attr;
  source.d.resize(source.n());
  for (size_t i = 0; i < source.d.size(); ++i) source.d[i] = (short)reconstructed[i];
  std::vector<Vol<short>> grey(n_slices);
  std::vector<M4> mo_inv(n_slices);
  std::vector<Target> targets;
  std::vector<int> which;
  targets.reserve(n_slices);
  std::vector<char> live(n_slices, 0);
  std::vector<M4> mo_all(n_slices);
  parallel_rows(n_slices, (size_t)sx * sy * 60, [&](int s0, int s1) {       // ParallelSliceToVolumeRegistration, RG.cc:2002-2050
    for (int s = s0; s < s1; ++s) {
      Vol<double> sl;
      sl.a = attrs[s];
      sl.d.resize(sl.n());
      for (int y = 0; y < sl.a.ny; ++y)
        for (int x = 0; x < sl.a.nx; ++x) sl.at(x, y, 0) = slices[((size_t)s * sy + y) * sx + x];
      // the patch-based caller (ParallelPatchToVolumeRegistration, patchBased2D3DRegistration.cpp:113-122) resamples into a variable
      // that shadows the patch and goes out of scope: its targets are registered as they are
      const Vol<double> t = (flags & SVRH_S2V_NO_RESAMPLE) ? sl : resample_with_padding<double>(sl, recon_attr->dx, recon_attr->dx, recon_attr->dx, -1.0);
      grey[s].a = t.a;
      grey[s].d.resize(t.d.size());
      short smax = -32768;
      for (size_t i = 0; i < t.d.size(); ++i) { grey[s].d[i] = (short)t.d[i]; smax = std::max(smax, grey[s].d[i]); }
      if (!(smax > -1)) continue;                                           // nothing to register
      reset_origin(grey[s].a, mo_all[s]);
      live[s] = 1;
    }
  });
  for (int s = 0; s < n_slices; ++s) {
    if (!live[s]) continue;
    M4 m;
    for (int q = 0; q < 16; ++q) m.m[q] = transformations[16 * s + q];
    mo_inv[s] = inverse_rigid_or_affine(mo_all[s]);
    Target tg;
    tg.full = &grey[s];
    tg.matrix = mul(m, mo_all[s]);
    matrix_to_params(tg.matrix, tg.p);
    targets.push_back(tg);
    which.push_back(s);
  }
  std::string e;
  const int rc = run_registrations(be, targets, source, 1, (short)-1, n_evaluations_or_null, e);   // SliceToVolume, SetTargetPadding(-1)
  if (rc) { set_err(err, e); return rc; }
  for (size_t k = 0; k < targets.size(); ++k) {
    const M4 m = mul(params_to_matrix(targets[k].p), mo_inv[which[k]]);
    for (int q = 0; q < 16; ++q) transformations[16 * which[k] + q] = m.m[q];
  }
  return 0;
}

// SplitImage, RG.cc:4979-5035: every `packages`-th slice, starting at l = 0 .. packages-1
static void split_image(const Vol<double> &image, int packages, std::vector<Vol<double>> &out) {
  const int pkg_z = image.a.nz / packages;
  const M4 i2w = image_to_world(image.a);
  for (int l = 0; l < packages; ++l) {
    Vol<double> st;
    st.a = image.a;
    st.a.nz = (pkg_z * packages + l) < image.a.nz ? pkg_z + 1 : pkg_z;
    st.a.dz = image.a.dz * packages;
    st.d.resize(st.n());
    for (int k = 0; k < st.a.nz; ++k)
      for (int j = 0; j < st.a.ny; ++j)
        for (int i = 0; i < st.a.nx; ++i) st.at(i, j, k) = image.at(i, j, k * packages + l);
    // adjust the origin so that voxel (0,0,0) of the package sits on voxel (0,0,l) of the image
    const double x = i2w.m[2] * l + i2w.m[3], y = i2w.m[6] * l + i2w.m[7], z = i2w.m[10] * l + i2w.m[11];
    const M4 s = image_to_world(st.a);
    st.a.origin[0] += x - s.m[3]; st.a.origin[1] += y - s.m[7]; st.a.origin[2] += z - s.m[11];
    out.push_back(st);
  }
}
static Vol<double> z_region(const Vol<double> &im, int z1, int z2) {     // GetRegion(0, 0, z1, nx, ny, z2)
  Vol<double> o;
  o.a = im.a;
  o.a.nz = z2 - z1;
  const M4 i2w = image_to_world(im.a);
  const double c[3] = {(im.a.nx - 1) / 2.0, (im.a.ny - 1) / 2.0, z1 + (o.a.nz - 1) / 2.0};
  for (int k = 0; k < 3; ++k) o.a.origin[k] = i2w.m[4 * k] * c[0] + i2w.m[4 * k + 1] * c[1] + i2w.m[4 * k + 2] * c[2] + i2w.m[4 * k + 3];
  o.d.assign(im.d.begin() + (size_t)z1 * im.a.nx * im.a.ny, im.d.begin() + (size_t)z2 * im.a.nx * im.a.ny);
  return o;
}
static void split_even_odd(const Vol<double> &image, int packages, std::vector<Vol<double>> &out) {      // RG.cc:5037-5055
  std::vector<Vol<double>> packs;
  split_image(image, packages, packs);
  for (const Vol<double> &p : packs) split_image(p, 2, out);
}
static void split_even_odd_half(const Vol<double> &image, int packages, std::vector<Vol<double>> &out, int iter) {   // RG.cc:5057-5093
  std::vector<Vol<double>> packs;
  if (iter > 1) split_even_odd_half(image, packages, packs, iter - 1);
  else split_even_odd(image, packages, packs);
  for (const Vol<double> &p : packs) {
    if (p.a.nz >= 4) { out.push_back(z_region(p, 0, p.a.nz / 2)); out.push_back(z_region(p, p.a.nz / 2, p.a.nz)); }   // HalfImage
    else out.push_back(p);
  }
}

// irtkReconstruction::PackageToVolume, RG.cc:5096-5192: every package of every stack is registered to the reconstruction as
// one 3-D target (GuessParameterSliceToVolume; the target padding is the corner guess, no SetTargetPadding) and its
// transformation is given to all of its slices.  transformations: one per slice, stacks in order.
int svrh_package_to_volume(svr_ctx *ctx, const svr_ncc_backend *backend, int n_stacks, const svr_image_attr *attrs, const double *const *stacks,
                           const int *pack_num, int evenodd, int half, int half_iter, double *transformations,
                           const svr_image_attr *recon_attr, const float *reconstructed, long *n_evaluations_or_null, char err[256]) {
  if ((!ctx && !backend) || n_stacks < 1 || !attrs || !stacks || !pack_num || !transformations || !recon_attr || !reconstructed) {
    set_err(err, "svrh_package_to_volume: bad arguments");
    return 1;
  }
  if (n_evaluations_or_null) *n_evaluations_or_null = 0;
  const Backend be{ctx, backend};
  Vol<short> source;
  source.a = *recon_attr;
  source.d.resize(source.n());
  for (size_t i = 0; i < source.d.size(); ++i) source.d[i] = (short)reconstructed[i];
  struct Pack { Vol<short> grey; M4 mo_inv; int first; std::vector<int> slices; };
  std::vector<Pack> packs;
  int first_slice = 0;
  for (int s = 0; s < n_stacks; ++s) {
    if (pack_num[s] < 1) { set_err(err, "svrh_package_to_volume: package counts must be positive"); return 1; }
    Vol<double> st;
    st.a = attrs[s];
    st.d.assign(stacks[s], stacks[s] + st.n());
    std::vector<Vol<double>> pk;
    if (evenodd) { if (half) split_even_odd_half(st, pack_num[s], pk, half_iter); else split_even_odd(st, pack_num[s], pk); }
    else split_image(st, pack_num[s], pk);
    const M4 s_w2i = world_to_image(st.a);
    for (const Vol<double> &p : pk) {
      if (p.a.nz < 1) continue;
      Pack q;
      const M4 p_i2w = image_to_world(p.a);
      for (int k = 0; k < p.a.nz; ++k) {                                    // which slices of the stack the package holds
        const double wx = p_i2w.m[2] * k + p_i2w.m[3], wy = p_i2w.m[6] * k + p_i2w.m[7], wz = p_i2w.m[10] * k + p_i2w.m[11];
        const double z = s_w2i.m[8] * wx + s_w2i.m[9] * wy + s_w2i.m[10] * wz + s_w2i.m[11];
        q.slices.push_back((int)irtk_round(z) + first_slice);
      }
      q.first = q.slices[0];
      q.grey.a = p.a;
      q.grey.d.resize(p.d.size());
      for (size_t i = 0; i < p.d.size(); ++i) q.grey.d[i] = (short)p.d[i];
      packs.push_back(q);
    }
    first_slice += st.a.nz;
  }
  std::vector<Target> targets(packs.size());
  for (size_t k = 0; k < packs.size(); ++k) {
    Pack &q = packs[k];
    Target &tg = targets[k];
    tg.own_padding = 1;
    tg.padding = guess_padding(q.grey);                                    // before ResetOrigin: values only
    M4 mo, m;
    reset_origin(q.grey.a, mo);
    q.mo_inv = inverse_rigid_or_affine(mo);
    for (int c = 0; c < 16; ++c) m.m[c] = transformations[16 * (size_t)q.first + c];
    tg.full = &q.grey;
    tg.matrix = mul(m, mo);
    matrix_to_params(tg.matrix, tg.p);
  }
  std::string e;
  const int rc = run_registrations(be, targets, source, 1, (short)0, n_evaluations_or_null, e);
  if (rc) { set_err(err, e); return rc; }
  for (size_t k = 0; k < packs.size(); ++k) {
    const M4 m = mul(params_to_matrix(targets[k].p), packs[k].mo_inv);      // PutMatrix: the first slice keeps this matrix
    double p6[6];
    matrix_to_params(m, p6);
    const M4 rebuilt = params_to_matrix(p6);                                // the other slices: parameters copied, UpdateMatrix
    for (size_t j = 0; j < packs[k].slices.size(); ++j) {
      const int sl = packs[k].slices[j];
      const M4 &use = sl == packs[k].first ? m : rebuilt;
      for (int c = 0; c < 16; ++c) transformations[16 * (size_t)sl + c] = use.m[c];
    }
  }
  return 0;
}

// building blocks, exported for the tests
int svrh_irtk_resample_with_padding(const svr_image_attr *attr, const int16_t *data, double rx, double ry, double rz, int padding,
                                    svr_image_attr *out_attr, int16_t *out_or_null, long capacity) {
  Vol<short> v;
  v.a = *attr;
  v.d.assign(data, data + v.n());
  const Vol<short> o = resample_with_padding<short>(v, rx, ry, rz, (short)padding);
  *out_attr = o.a;
  if (out_or_null) {
    if ((long)o.d.size() > capacity) return 1;
    memcpy(out_or_null, o.d.data(), sizeof(int16_t) * o.d.size());
  }
  return 0;
}

int svrh_irtk_blur_with_padding(const svr_image_attr *attr, int16_t *data, double sigma, int padding) {
  Vol<short> v;
  v.a = *attr;
  v.d.assign(data, data + v.n());
  blur_with_padding(v, sigma, (short)padding);
  memcpy(data, v.d.data(), sizeof(int16_t) * v.d.size());
  return 0;
}

void svrh_irtk_rigid_parameters(const double matrix16[16], double params6[6], double *rebuilt16_or_null) {
  M4 m;
  for (int q = 0; q < 16; ++q) m.m[q] = matrix16[q];
  matrix_to_params(m, params6);
  if (rebuilt16_or_null) { const M4 r = params_to_matrix(params6); for (int q = 0; q < 16; ++q) rebuilt16_or_null[q] = r.m[q]; }
}

}  // extern "C"
