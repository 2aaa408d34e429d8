// svr_cli.cpp -- `SVRreconstructionGPU`: the reference's command line (source/reconstructionGPU2/
// reconstruction.cc, "main.cc") over the MI355X engine, in C++ like the reference's own main().
//
//   SVRreconstructionGPU -o recon.nii.gz -i s1.nii.gz s2.nii.gz ... -m mask.nii.gz [--thickness t1 t2 ...]
//                        [--resolution 0.75] [--iterations 4] [--useGPUReg] ...
//
// Option names and defaults: main.cc:164-211.  Set-up (main.cc:386-815): read the stacks, crop them to the
// mask, CreateTemplate, SetMask, MatchStackIntensitiesWithMasking, CreateSlicesAndTransformations,
// MaskSlices, SyncGPU.  Loop (main.cc:816-1237): [slice-to-volume registration] -> smoothing schedule ->
// Gaussian reconstruction -> robust statistics -> SR iterations -> mask; finally RestoreSliceIntensities +
// ScaleVolume and the volume is written.  The pre-processing functions restate irtkReconstruction's
// (irtkReconstructionGPU.cc = "RG.cc"; the line ranges are on each function) and agree with the Python
// mirror fetalreconstruction_amd/preprocess.py, which the tests compare them with.
//
// Transformations (-t) are `id`, IRTK rigid `dof` files or 4x4 text matrices, used as given.  Not built, refused
// loudly: the stack-to-stack registration that refines them (RG.cc:849-1001), the CPU/IRTK slice registration (slice
// registration runs with --useGPUReg only), packages, patch/superpixel modes, the CPU path.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <fstream>
#include <string>
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

void die(const std::string &m) { fprintf(stderr, "SVRreconstructionGPU: %s\n", m.c_str()); exit(1); }

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
    for (int z = 0; z < a.nz; ++z)
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
    im.d.swap(out);
  }
}

// irtkImageTransformation + nearest neighbour, target padding -1 on an all-zero target (RG.cc:782-793, 808-819)
Image transform_nn(const Image &src, const svr_image_attr &target, const M4 &t, double source_padding) {
  Image out;
  out.a = target;
  out.d.assign((size_t)target.nx * target.ny * target.nz, source_padding);
  const M4 m = mul(world_to_image(src.a), mul(t, image_to_world(target)));
  for (int z = 0; z < target.nz; ++z)
    for (int y = 0; y < target.ny; ++y)
      for (int x = 0; x < target.nx; ++x) {
        const double q[3] = {m.m[0] * x + m.m[1] * y + m.m[2] * z + m.m[3], m.m[4] * x + m.m[5] * y + m.m[6] * z + m.m[7],
                             m.m[8] * x + m.m[9] * y + m.m[10] * z + m.m[11]};
        const long i = (long)irtk_round(q[0]), j = (long)irtk_round(q[1]), k = (long)irtk_round(q[2]);
        if (i >= 0 && i < src.a.nx && j >= 0 && j < src.a.ny && k >= 0 && k < src.a.nz)
          out.at(x, y, z) = src.at((int)i, (int)j, (int)k);
      }
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

// irtkGenericImage::GetRegion(i1, j1, k1, i2, j2, k2)
Image get_region(const Image &im, int x1, int y1, int z1, int x2, int y2, int z2) {
  Image o;
  o.a = im.a;
  o.a.nx = x2 - x1; o.a.ny = y2 - y1; o.a.nz = z2 - z1;
  const M4 i2w = image_to_world(im.a);
  const double c[3] = {x1 + (o.a.nx - 1) / 2.0, y1 + (o.a.ny - 1) / 2.0, z1 + (o.a.nz - 1) / 2.0};
  for (int k = 0; k < 3; ++k) o.a.origin[k] = i2w.m[4 * k] * c[0] + i2w.m[4 * k + 1] * c[1] + i2w.m[4 * k + 2] * c[2] + i2w.m[4 * k + 3];
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
  std::vector<double> avg;
  for (size_t s = 0; s < stacks.size(); ++s) {
    const Image &st = stacks[s];
    const M4 m = mul(mw2i, mul(ts[s], image_to_world(st.a)));
    double sum = 0, num = 0;
    for (int z = 0; z < st.a.nz; ++z)
      for (int y = 0; y < st.a.ny; ++y)
        for (int x = 0; x < st.a.nx; ++x) {
          const long i = (long)irtk_round(m.m[0] * x + m.m[1] * y + m.m[2] * z + m.m[3]);
          const long j = (long)irtk_round(m.m[4] * x + m.m[5] * y + m.m[6] * z + m.m[7]);
          const long k = (long)irtk_round(m.m[8] * x + m.m[9] * y + m.m[10] * z + m.m[11]);
          if (i >= 0 && i < mask.a.nx && j >= 0 && j < mask.a.ny && k >= 0 && k < mask.a.nz && mask.at((int)i, (int)j, (int)k) == 1) {
            sum += st.at(x, y, z);
            num += 1;
          }
        }
    if (!(num > 0)) die("a stack has no overlap with the ROI");
    avg.push_back(sum / num);
  }
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

int main(int argc, char **argv) {
  std::string output, mask_name;
  std::vector<std::string> inputs, tspecs;
  std::vector<double> thickness;
  std::vector<int> force_excluded, devices;
  int iterations = 4, levels = 3, rec_first = 4, rec_last = 13;
  double resolution = 0.75, average = 700, delta = 150, lambda = 0.02, last_lambda = 0.01, smooth_mask = 4;
  bool no_matching = false, use_gpu_reg = false;
  // ---- options (main.cc:164-211) ---------------------------------------------------------------------
  auto is_opt = [](const char *s) { return s[0] == '-' && !(s[1] >= '0' && s[1] <= '9') && s[1] != '.'; };
  for (int i = 1; i < argc; ++i) {
    const std::string o = argv[i];
    auto multi = [&](std::vector<std::string> &dst) { while (i + 1 < argc && !is_opt(argv[i + 1])) dst.push_back(argv[++i]); };
    auto one = [&]() -> std::string { if (i + 1 >= argc) die("missing value for " + o); return argv[++i]; };
    if (o == "-o" || o == "--output") output = one();
    else if (o == "-m" || o == "--mask") mask_name = one();
    else if (o == "-i" || o == "--input") multi(inputs);
    else if (o == "-t" || o == "--transformation") multi(tspecs);
    else if (o == "--thickness") { std::vector<std::string> v; multi(v); for (auto &s : v) thickness.push_back(atof(s.c_str())); }
    else if (o == "--iterations") iterations = atoi(one().c_str());
    else if (o == "--sigma") (void)one();                               // bias field stdev: bias correction stays disabled
    else if (o == "--resolution") resolution = atof(one().c_str());
    else if (o == "--multires") levels = atoi(one().c_str());
    else if (o == "--average") average = atof(one().c_str());
    else if (o == "--delta") delta = atof(one().c_str());
    else if (o == "--lambda") lambda = atof(one().c_str());
    else if (o == "--lastIterLambda") last_lambda = atof(one().c_str());
    else if (o == "--smooth_mask") smooth_mask = atof(one().c_str());
    else if (o == "--no_intensity_matching") no_matching = true;
    else if (o == "--force_exclude") { std::vector<std::string> v; multi(v); for (auto &s : v) force_excluded.push_back(atoi(s.c_str())); }
    else if (o == "--rec_iterations_first") rec_first = atoi(one().c_str());
    else if (o == "--rec_iterations_last") rec_last = atoi(one().c_str());
    else if (o == "--useGPUReg") use_gpu_reg = true;
    else if (o == "--useCPUReg" || o == "--disableBiasCorrection" || o == "--debug_gpu") {}
    else if (o == "-d" || o == "--devices") { std::vector<std::string> v; multi(v); for (auto &s : v) devices.push_back(atoi(s.c_str())); }
    else if (o == "-h" || o == "--help") {
      printf("usage: SVRreconstructionGPU -o <volume> -i <stack_1> .. <stack_N> [-m <mask>] [-t id|<4x4.txt> ..] [--thickness th_1 ..]\n"
             "       [--iterations 4] [--resolution 0.75] [--multires 3] [--average 700] [--delta 150] [--lambda 0.02]\n"
             "       [--lastIterLambda 0.01] [--smooth_mask 4] [--no_intensity_matching] [--force_exclude i ..]\n"
             "       [--rec_iterations_first 4] [--rec_iterations_last 13] [--useGPUReg] [-d device]\n");
      return 0;
    } else {
      die("option " + o + " is not supported by this build (see csrc/svr_cli.cpp)");
    }
  }
  if (output.empty() || inputs.empty()) die("-o and -i are required (try --help)");
  const size_t n = inputs.size();
  if (tspecs.empty()) tspecs.assign(n, "id");
  if (tspecs.size() != n) die("one transformation per stack expected");

  // ---- set-up (main.cc:386-815) ----------------------------------------------------------------------
  std::vector<Image> stacks;
  std::vector<M4> ts;
  for (size_t k = 0; k < n; ++k) { stacks.push_back(read_image(inputs[k])); ts.push_back(load_transformation(tspecs[k])); }
  if (thickness.empty()) for (auto &s : stacks) thickness.push_back(2.0 * s.a.dz);            // main.cc:422-431
  if (thickness.size() != n) die("one thickness per stack expected");
  size_t tmpl = 0;
  for (size_t k = 0; k < n; ++k) if (tspecs[k] == "id") { tmpl = k; break; }
  Image mask_img;
  const bool have_mask = !mask_name.empty();
  if (have_mask) {
    mask_img = read_image(mask_name);
    const Image m = transform_nn(mask_img, stacks[tmpl].a, ts[tmpl], 0.0);                     // TransformMask RG.cc:805-821
    stacks[tmpl] = crop_image(stacks[tmpl], m);
  }
  const svr_image_attr tattr = create_template(stacks[tmpl].a, resolution);
  const Image vol_mask = set_mask(tattr, have_mask ? &mask_img : nullptr, smooth_mask);
  for (size_t k = 0; k < n; ++k) {                                                               // main.cc:645-662
    if (k == tmpl) continue;
    const Image m = transform_nn(vol_mask, stacks[k].a, ts[k], 0.0);
    stacks[k] = crop_image(stacks[k], m);
  }
  const std::vector<float> factors = match_stack_intensities(stacks, ts, vol_mask, average, no_matching);
  // CreateSlicesAndTransformations RG.cc:1835-1880 + MaskSlices RG.cc:1940-1988 + the packing of SyncGPU RG.cc:249-328
  int ns = 0, mx = 0, my = 0;
  for (auto &s : stacks) { ns += s.a.nz; mx = std::max(mx, s.a.nx); my = std::max(my, s.a.ny); }
  std::vector<float> grid((size_t)ns * mx * my, -1.0f), i2w(16 * (size_t)ns), w2i(16 * (size_t)ns), st(16 * (size_t)ns),
      sti(16 * (size_t)ns), dims(3 * (size_t)ns);
  std::vector<int> sizes_x(ns), sizes_y(ns), stack_index(ns);
  std::vector<svr_image_attr> sattr(ns);
  std::vector<double> T(16 * (size_t)ns);
  const M4 mw2i = world_to_image(vol_mask.a);
  double vmin = 1e300, vmax = -1e300;
  int sl = 0;
  for (size_t k = 0; k < n; ++k)
    for (int j = 0; j < stacks[k].a.nz; ++j, ++sl) {
      Image r = get_region(stacks[k], 0, 0, j, stacks[k].a.nx, stacks[k].a.ny, j + 1);
      r.a.dz = thickness[k];
      const M4 si2w = image_to_world(r.a), m = mul(mw2i, mul(ts[k], si2w));
      for (int y = 0; y < r.a.ny; ++y)
        for (int x = 0; x < r.a.nx; ++x) {
          double v = r.at(x, y, 0);
          if (v < 0.01) v = -1;
          const long i = (long)irtk_round(m.m[0] * x + m.m[1] * y + m.m[3]), jj = (long)irtk_round(m.m[4] * x + m.m[5] * y + m.m[7]),
                     kk = (long)irtk_round(m.m[8] * x + m.m[9] * y + m.m[11]);
          if (!(i >= 0 && i < vol_mask.a.nx && jj >= 0 && jj < vol_mask.a.ny && kk >= 0 && kk < vol_mask.a.nz) ||
              vol_mask.at((int)i, (int)jj, (int)kk) == 0)
            v = -1;
          grid[((size_t)sl * my + y) * mx + x] = (float)v;
          if (v > 0) { vmin = std::min(vmin, (double)(float)v); vmax = std::max(vmax, (double)(float)v); }
        }
      to_f16(si2w, &i2w[16 * (size_t)sl]); to_f16(world_to_image(r.a), &w2i[16 * (size_t)sl]);
      to_f16(ts[k], &st[16 * (size_t)sl]); to_f16(inverse_rigid_or_affine(ts[k]), &sti[16 * (size_t)sl]);
      for (int q = 0; q < 16; ++q) T[16 * (size_t)sl + q] = ts[k].m[q];
      dims[3 * (size_t)sl] = (float)r.a.dx; dims[3 * (size_t)sl + 1] = (float)r.a.dy; dims[3 * (size_t)sl + 2] = (float)r.a.dz;
      sizes_x[sl] = r.a.nx; sizes_y[sl] = r.a.ny; stack_index[sl] = (int)k; sattr[sl] = r.a;
    }
  if (!(vmax > 0)) die("no slice pixel lies inside the mask");
  fprintf(stderr, "%zu stacks, %d slices of up to %dx%d, volume %dx%dx%d at %g mm\n", n, ns, mx, my, tattr.nx, tattr.ny, tattr.nz,
          resolution);

  // ---- SyncGPU + generatePSFVolume + UpdateGPUTranformationMatrices (RG.cc:249-401, 1496-1610) ----------
  svr_ctx *ctx = nullptr;
  if (svr_create(devices.empty() ? 0 : devices[0], &ctx) || !ctx) die("no usable HIP device (svr_create failed)");
  const uint32_t vsize[3] = {(uint32_t)tattr.nx, (uint32_t)tattr.ny, (uint32_t)tattr.nz};
  const float vdim[3] = {(float)tattr.dx, (float)tattr.dy, (float)tattr.dz};
  std::vector<float> maskf(vol_mask.d.begin(), vol_mask.d.end());
  float ri2w[16], rw2i[16];
  to_f16(image_to_world(tattr), ri2w); to_f16(world_to_image(tattr), rw2i);
  ENG(svr_init_reconstruction_volume(ctx, vsize, vdim, nullptr, 12.0f));
  ENG(svr_set_mask(ctx, vsize, vdim, maskf.data(), 12.0f));
  const uint32_t ssize[3] = {(uint32_t)mx, (uint32_t)my, (uint32_t)ns};
  ENG(svr_init_storage_volumes(ctx, ssize, &dims[0]));
  ENG(svr_fill_slices(ctx, grid.data(), sizes_x.data(), sizes_y.data()));
  ENG(svr_set_slice_dims(ctx, dims.data(), 2.0f));
  {
    svr_image_attr pa;                                                   // PSF_SIZE 128 (RC.cuh:56), RG.cc:1534-1551
    memset(&pa, 0, sizeof(pa));
    pa.nx = pa.ny = pa.nz = 128; pa.dx = tattr.dx; pa.dy = tattr.dy; pa.dz = tattr.dz;
    pa.xaxis[0] = pa.yaxis[1] = pa.zaxis[2] = 1.0;
    float pi2w[16], pw2i[16];
    to_f16(image_to_world(pa), pi2w); to_f16(world_to_image(pa), pw2i);
    const uint32_t psize[3] = {128, 128, 128};
    ENG(svr_generate_psf_volume(ctx, nullptr, psize, &dims[0], vdim, pi2w, pw2i, 2.0f));
  }
  ENG(svr_set_slice_matrices(ctx, st.data(), sti.data(), i2w.data(), w2i.data(), i2w.data(), w2i.data(), ri2w, rw2i));

  svrh_recon *host = svrh_create(ctx, ns, 0, ns, nullptr);
  if (!host) die("svrh_create failed");
  svrh_set_intensity_range(host, vmin, vmax);                            // InitializeEMGPU RG.cc:2937-2951
  if (!force_excluded.empty()) svrh_set_force_excluded(host, force_excluded.data(), (int)force_excluded.size());
  if (use_gpu_reg) HOST(svrh_prepare_registration_slices(host, grid.data(), mx, my, sattr.data(), resolution));

  // ---- registration-reconstruction loop (main.cc:816-1237) ---------------------------------------------
  for (int it = 0; it < iterations; ++it) {
    if (it > 0 && use_gpu_reg) {
      HOST(svrh_slice_to_volume_registration_gpu(host, T.data()));
      for (int s = 0; s < ns; ++s) {                                      // UpdateGPUTranformationMatrices RG.cc:372-401
        M4 t;
        for (int q = 0; q < 16; ++q) t.m[q] = T[16 * (size_t)s + q];
        to_f16(t, &st[16 * (size_t)s]); to_f16(inverse_rigid_or_affine(t), &sti[16 * (size_t)s]);
      }
      ENG(svr_set_slice_matrices(ctx, st.data(), sti.data(), i2w.data(), w2i.data(), i2w.data(), w2i.data(), ri2w, rw2i));
    }
    if (it == iterations - 1) {                                           // main.cc:884-896
      svrh_set_smoothing_parameters(host, delta, last_lambda);
    } else {
      double l = lambda;
      for (int i = 0; i < levels; ++i) {
        if (it == iterations * (levels - i - 1) / levels) svrh_set_smoothing_parameters(host, delta, l);
        l *= 2;
      }
    }
    HOST(svrh_reconstruct_iteration(host, it == iterations - 1 ? rec_last : rec_first));   // main.cc:930-1140
    double sc[8];
    svrh_get_state(host, nullptr, nullptr, nullptr, nullptr, sc);
    fprintf(stderr, "iteration %d: sigma %.4g mix %.3f\n", it, sc[0], sc[1]);
  }
  ENG(svr_restore_slice_intensities(ctx, factors.data(), (int)factors.size(), stack_index.data()));   // main.cc:1189-1193
  HOST(svrh_scale_volume_gpu(host));
  std::vector<float> vol((size_t)tattr.nx * tattr.ny * tattr.nz);
  ENG(svr_sync_cpu(ctx, vol.data()));
  char err[256] = {0};
  if (svr_nifti_write(output.c_str(), &tattr, vol.data(), err)) die(output + ": " + err);
  svrh_destroy(host);
  svr_destroy(ctx);
  return 0;
}
