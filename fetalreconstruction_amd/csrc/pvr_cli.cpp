// pvr_cli.cpp -- `PVRreconstructionGPU`: the reference's patch-to-volume command line (source/reconstructionGPU2/
// patchBasedReconMain.cpp, "pvrmain") over the MI355X engine, in C++ like the reference's own main().
//
//   PVRreconstructionGPU -o recon.nii.gz -i s1.nii.gz s2.nii.gz ... -m mask.nii.gz [--patchSize 32 32]
//                        [--patchStride 16 16] [--resolution 0.75] [--iterations 7] [--sr_iterations 7] ...
//
// Option names and defaults: pvrmain:108-131.  Set-up (irtkPatchBasedReconstruction<T>::run,
// irtkPatchBasedReconstruction.cpp = "PBR.cpp" :193-310): binarise the mask, crop every stack to it, resample the
// mask to the isotropic voxel size (nearest neighbour), match the stack intensities (:656-790, target = mean of all
// positive voxels), min/max intensities (:792-814), CreateTemplate (:941-965), the reconstruction mask (:303-304);
// then PatchBasedVolume::generate2DPatches per stack (patchBasedObject.cuh:176-342) and `iterations + 1` passes of
// the loop in csrc/pvr_host.cpp (PBR.cpp:445-593).  The Python twin is fetalreconstruction_amd/pvr_cli.py.
//
// The stack-to-stack registration (irtkStack3D3DRegistration, :280-285) runs through csrc/irtk_reg.cpp with every similarity
// on the GPU; between the outer passes every patch is registered to the volume with the same schedule
// (patchBased2D3DRegistration<T>::runHybrid, what PBR.cpp:472-476 calls); --no_registration (not a reference option) skips both.
// -s / --superpixel cuts SLICO superpixel patches (csrc/svr_slic.h).  --useFullSlices makes every slice one patch (patchBasedObject.cuh:183-189).
// --existingReconTarget starts from a given volume (and its grid); --hierarchical runs iterations + 1 levels of shrinking patches (pvrmain:359-432).
// --dilateMask n dilates the mask n times, --packages p_1 .. p_N splits every stack into its interleaved packages (PBR.cpp:134-146).
// --resample resamples the cropped stacks to the output voxel size with IRTK's cubic B-spline interpolator.
#include <float.h>

#include <future>

#include "svr_prep.h"
#include "svr_shard.h"
#include "svr_slic.h"

namespace {

// irtkResampling::Initialize (irtkResampling.cc:74-130): int(n d_old / d) voxels of size d, same axes and origin
svr_image_attr resample_attr(const svr_image_attr &in, double d) {
  svr_image_attr a = in;
  int n[3] = {(int)(a.nx * a.dx / d), (int)(a.ny * a.dy / d), (int)(a.nz * a.dz / d)};
  double s[3] = {d, d, d};
  const double old[3] = {a.dx, a.dy, a.dz};
  for (int k = 0; k < 3; ++k) if (n[k] < 1) { n[k] = 1; s[k] = old[k]; }
  a.nx = n[0]; a.ny = n[1]; a.nz = n[2]; a.dx = s[0]; a.dy = s[1]; a.dz = s[2];
  return a;
}

// MatchStackIntensitiesWithMasking PBR.cpp:656-790
void match_stack_intensities_pvr(std::vector<Image> &stacks, const std::vector<M4> &ts, const Image &mask) {
  float average = 0;                                    // T m_average_value, accumulated voxel by voxel
  unsigned long long count = 0;
  for (const Image &s : stacks)
    for (double v : s.d)
      if (v > 0) { average += (float)v; count++; }
  if (count) average /= count;
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
          if (i >= 0 && i < mask.a.nx && j >= 0 && j < mask.a.ny && k >= 0 && k < mask.a.nz &&
              mask.at((int)i, (int)j, (int)k) == 1 && st.at(x, y, z) > 0) {
            sum += st.at(x, y, z);
            num += 1;
          }
        }
    if (!(num > 0)) die("a stack has no overlap with the ROI");
    avg[s] = sum / num;
  });
  for (size_t s = 0; s < stacks.size(); ++s) {
    const double f = average / avg[s];
    for (double &v : stacks[s].d) if (v > 0) v = (double)(float)(v * f);      // float voxels times a double factor
  }
}

// irtkDilation<T> with CONNECTIVITY_26 (irtkDilation.cc:50-78): an interior voxel becomes the maximum of its 26 neighbours and
// itself, the voxels on the faces of the image keep their value
void dilate_mask(Image &m, int iterations) {
  const int nx = m.a.nx, ny = m.a.ny, nz = m.a.nz;
  for (int it = 0; it < iterations; ++it) {
    const std::vector<double> in = m.d;
    for (int z = 1; z + 1 < nz; ++z)
      for (int y = 1; y + 1 < ny; ++y)
        for (int x = 1; x + 1 < nx; ++x) {
          double v = in[((size_t)z * ny + y) * nx + x];
          for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
              for (int dx = -1; dx <= 1; ++dx) v = std::max(v, in[((size_t)(z + dz) * ny + (y + dy)) * nx + (x + dx)]);
          m.d[((size_t)z * ny + y) * nx + x] = v;
        }
  }
}

// ConvertToInterpolationCoefficients, cubic spline, mirror boundaries (irtkBSplineInterpolateImageFunction.cc:68-144)
void bspline_coefficients(double *c, int n) {
  if (n == 1) return;
  const double z = sqrt(3.0) - 2.0;
  const double lambda = (1.0 - z) * (1.0 - 1.0 / z);
  for (int k = 0; k < n; ++k) c[k] *= lambda;
  const int horizon = (int)ceil(log(DBL_EPSILON) / log(fabs(z)));
  double zn = z, sum;
  if (horizon < n) {
    sum = c[0];
    for (int k = 1; k < horizon; ++k) { sum += zn * c[k]; zn *= z; }
    c[0] = sum;
  } else {
    const double iz = 1.0 / z;
    double z2n = pow(z, (double)(n - 1));
    sum = c[0] + z2n * c[n - 1];
    z2n *= z2n * iz;
    for (int k = 1; k <= n - 2; ++k) { sum += (zn + z2n) * c[k]; zn *= z; z2n *= iz; }
    c[0] = sum / (1.0 - zn * zn);
  }
  for (int k = 1; k < n; ++k) c[k] += z * c[k - 1];
  c[n - 1] = (z / (z * z - 1.0)) * (z * c[n - 2] + c[n - 1]);
  for (int k = n - 2; k >= 0; --k) c[k] = z * (c[k + 1] - c[k]);
}

// irtkResampling<T> with irtkBSplineInterpolateImageFunction (cubic, clamped to the input's range): what --resample does to every
// cropped stack (PBR.cpp:225-246; irtkResampling.cc:74-175, irtkBSplineInterpolateImageFunction.cc:146-433); float voxels
Image resample_bspline(const Image &img, double d) {
  const svr_image_attr &a = img.a;
  const int nx = a.nx, ny = a.ny, nz = a.nz;
  std::vector<double> coeff = img.d, line((size_t)std::max(nx, std::max(ny, nz)));
  auto C = [&](int x, int y, int z) -> double & { return coeff[((size_t)z * ny + y) * nx + x]; };
  for (int z = 0; z < nz; ++z)
    for (int y = 0; y < ny; ++y) bspline_coefficients(&C(0, y, z), nx);
  for (int z = 0; z < nz; ++z)
    for (int x = 0; x < nx; ++x) {
      for (int y = 0; y < ny; ++y) line[y] = C(x, y, z);
      bspline_coefficients(line.data(), ny);
      for (int y = 0; y < ny; ++y) C(x, y, z) = line[y];
    }
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < nx; ++x) {
      for (int z = 0; z < nz; ++z) line[z] = C(x, y, z);
      bspline_coefficients(line.data(), nz);
      for (int z = 0; z < nz; ++z) C(x, y, z) = line[z];
    }
  double vmin = img.d[0], vmax = img.d[0];
  for (double v : img.d) { vmin = std::min(vmin, v); vmax = std::max(vmax, v); }
  Image out;
  out.a = resample_attr(a, d);
  out.d.resize((size_t)out.a.nx * out.a.ny * out.a.nz);
  const M4 m = mul(world_to_image(a), image_to_world(out.a));
  const int dims[3] = {nx, ny, nz};
  for (int k = 0; k < out.a.nz; ++k)
    for (int j = 0; j < out.a.ny; ++j)
      for (int i = 0; i < out.a.nx; ++i) {
        int idx[3][4];
        double wgt[3][4];
        for (int r = 0; r < 3; ++r) {
          const double x = m.m[4 * r] * i + m.m[4 * r + 1] * j + m.m[4 * r + 2] * k + m.m[4 * r + 3];
          const int i0 = (int)floor(x) - 1;
          const double w = x - (double)(i0 + 1);
          wgt[r][3] = (1.0 / 6.0) * w * w * w;
          wgt[r][0] = (1.0 / 6.0) + (1.0 / 2.0) * w * (w - 1.0) - wgt[r][3];
          wgt[r][2] = w + wgt[r][0] - 2.0 * wgt[r][3];
          wgt[r][1] = 1.0 - wgt[r][0] - wgt[r][2] - wgt[r][3];
          const int n = dims[r], half = 2 * n - 2;
          for (int q = 0; q < 4; ++q) {
            int v = i0 + q;
            v = (n == 1) ? 0 : (v < 0 ? -v - half * ((-v) / half) : v - half * (v / half));
            if (n <= v) v = half - v;
            idx[r][q] = v;
          }
        }
        double value = 0.0;
        for (int c = 0; c < 4; ++c)
          for (int b = 0; b < 4; ++b)
            for (int e = 0; e < 4; ++e) value += wgt[0][e] * wgt[1][b] * wgt[2][c] * C(idx[0][e], idx[1][b], idx[2][c]);
        value = std::min(std::max(value, vmin), vmax);
        out.d[((size_t)k * out.a.ny + j) * out.a.nx + i] = (double)(float)value;
      }
  return out;
}

// patchBasedPackageSplitter<T>::makePackageVolumes (patchBasedPackageSplitter.cpp:76-146): package l holds slices l, l + packages, ...
// of the stack at `packages` times the slice spacing, with its first voxel where slice l's first voxel was
std::vector<Image> split_packages(const Image &stack, int packages) {
  std::vector<Image> out;
  const svr_image_attr &a = stack.a;
  const int pkg_z = a.nz / packages;
  const M4 i2w = image_to_world(a);
  for (int l = 0; l < packages; ++l) {
    Image p;
    p.a = a;
    p.a.nz = (pkg_z * packages + l < a.nz) ? pkg_z + 1 : pkg_z;
    p.a.dz = a.dz * packages;
    const M4 first = image_to_world(p.a);
    for (int k = 0; k < 3; ++k) p.a.origin[k] += (i2w.m[4 * k + 2] * l + i2w.m[4 * k + 3]) - first.m[4 * k + 3];
    p.d.resize((size_t)p.a.nx * p.a.ny * p.a.nz);
    const size_t plane = (size_t)a.nx * a.ny;
    for (int k = 0; k < p.a.nz; ++k) memcpy(&p.d[(size_t)k * plane], &stack.d[(size_t)(k * packages + l) * plane], plane * sizeof(double));
    out.push_back(std::move(p));
  }
  return out;
}

struct Patches {
  std::vector<float> data, i2w, w2i;                   // [n][py][px], [n][16], [n][16]
  std::vector<float> ri2w, mo, invmo;                  // the origin-reset matrices (patchBasedObject.cuh:285-304), [n][16] each
  std::vector<svr_image_attr> attr;                    // the patches as images (targets of the patch-to-volume registration)
  int n = 0;
};

// A patch pixel sits exactly on a slice pixel (and, for a mask drawn on the stack's grid, on a mask voxel): its coordinate is an
// integer in exact arithmetic, and the reference truncates what the double arithmetic makes of it (patchBasedObject.cuh:262-277),
// so on an oblique grid 19.999999999999996 reads pixel 19 or 20 depending on the last bit.  Here a coordinate within 1e-6 of an
// integer IS that integer; everything else is untouched.
inline double snap(double v) { const double r = nearbyint(v); return fabs(v - r) < 1e-6 ? r : v; }

// PatchBasedVolume<T>::generate2DPatches, patchBasedObject.cuh:176-342
void append(Patches &to, const Patches &from) {
  to.data.insert(to.data.end(), from.data.begin(), from.data.end());
  to.i2w.insert(to.i2w.end(), from.i2w.begin(), from.i2w.end()); to.w2i.insert(to.w2i.end(), from.w2i.begin(), from.w2i.end());
  to.ri2w.insert(to.ri2w.end(), from.ri2w.begin(), from.ri2w.end()); to.mo.insert(to.mo.end(), from.mo.begin(), from.mo.end());
  to.invmo.insert(to.invmo.end(), from.invmo.begin(), from.invmo.end());
  to.attr.insert(to.attr.end(), from.attr.begin(), from.attr.end());
  to.n += from.n;
}

void generate_2d_patches(const Image &stack, double thickness, const Image &mask, int px, int py, int sx, int sy, Patches &result) {
  const svr_image_attr &a = stack.a;
  const M4 m_w2i = world_to_image(mask.a);
  std::vector<Patches> per_slice(a.nz);                  // the slices are cut on the host threads and appended in slice order
  parallel_for(a.nz, [&](int z) {
    Patches &out = per_slice[z];
    std::vector<float> patch((size_t)px * py);
    {
    svr_image_attr sl = a;                               // GetRegion(0, 0, z, x, y, z + 1) + PutPixelSize :202-203
    sl.nz = 1;
    sl.dz = thickness * 2;
    region_origin(a, 0, 0, z, sl);
    const M4 sl_i2w = image_to_world(sl), sl_w2i = world_to_image(sl);
    svr_image_attr p0 = sl;
    p0.nx = px; p0.ny = py;
    p0.origin[0] = p0.origin[1] = p0.origin[2] = 0;
    const M4 p0_i2w = image_to_world(p0);
    for (int y = 0; y < a.ny + py; y += sy)
      for (int x = 0; x < a.nx + px; x += sx) {
        svr_image_attr pa = p0;                          // shift the origin so that pixel (0,0) sits on slice pixel (x,y) :232-246
        for (int k = 0; k < 3; ++k) pa.origin[k] = (sl_i2w.m[4 * k] * x + sl_i2w.m[4 * k + 1] * y + sl_i2w.m[4 * k + 3]) - p0_i2w.m[4 * k + 3];
        const M4 p_i2w = image_to_world(pa);
        int set_count = 0;
        for (int j = 0; j < py; ++j)
          for (int i = 0; i < px; ++i) {
            double wx = i, wy = j, wz = 0;                 // patch.ImageToWorld, then slice / mask WorldToImage (:262-277)
            apply_point(p_i2w, wx, wy, wz);
            double xx = wx, yy = wy, zz = wz, x1 = wx, y1 = wy, z1 = wz;
            apply_point(sl_w2i, xx, yy, zz);
            apply_point(m_w2i, x1, y1, z1);
            xx = snap(xx); yy = snap(yy); x1 = snap(x1); y1 = snap(y1); z1 = snap(z1);
            float v = 0;                                 // a patch starts as an all-zero image
            if (xx >= 0 && yy >= 0 && xx < a.nx && yy < a.ny && x1 >= 0 && y1 >= 0 && z1 >= 0 && x1 < mask.a.nx && y1 < mask.a.ny &&
                z1 < mask.a.nz && mask.at((int)x1, (int)y1, (int)z1) > 0) {
              v = (float)stack.at((int)xx, (int)yy, z);
              if (v != 0 && v != -1) set_count++;
            }
            patch[(size_t)j * px + i] = v;
          }
        if (set_count > 1.0f / 3.0f * py * px) {         // :318
          out.data.insert(out.data.end(), patch.begin(), patch.end());
          float f[16];
          to_f16(p_i2w, f); out.i2w.insert(out.i2w.end(), f, f + 16);
          to_f16(world_to_image(pa), f); out.w2i.insert(out.w2i.end(), f, f + 16);
          to_f16(p0_i2w, f); out.ri2w.insert(out.ri2w.end(), f, f + 16);           // RI2W: the patch with its origin at 0
          M4 mo = ident();
          for (int k = 0; k < 3; ++k) mo.m[4 * k + 3] = pa.origin[k];
          to_f16(mo, f); out.mo.insert(out.mo.end(), f, f + 16);
          to_f16(inverse_rigid_or_affine(mo), f); out.invmo.insert(out.invmo.end(), f, f + 16);
          out.attr.push_back(pa);
          out.n++;
        }
      }
    }
  });
  for (const Patches &p : per_slice) append(result, p);
}

#define PVRH(call) do { int rc_ = (call); if (rc_) die(std::string(#call) + ": " + pvrh_last_error(host)); } while (0)

}  // namespace

int main(int argc, char **argv) {
  prog_name = "PVRreconstructionGPU";
  std::string output, mask_name;
  std::vector<std::string> inputs, tspecs;
  std::vector<double> thickness;
  std::vector<int> devices, psize, pstride, packages;
  int iterations = 7, sr_iterations = 7, dilate = 0;
  double resolution = 0.75;
  bool no_matching = false, dry_run = false, no_registration = false, superpixel = false, full_slices = false, hierarchical = false, resample = false, coeff_table = false;
  int spx_size = 16, spx_extend = 50;                    // pvrmain:104-106
  std::string existing_name;                             // --existingReconTarget (pvrmain:118)
  std::string dump_name;                                 // test hooks: --dumpProblem <file> [--dryRun]
  // ---- options (pvrmain:108-131) ---------------------------------------------------------------------
  auto is_opt = [](const char *s) { return s[0] == '-' && !(s[1] >= '0' && s[1] <= '9') && s[1] != '.'; };
  for (int i = 1; i < argc; ++i) {
    const std::string o = argv[i];
    auto multi = [&](std::vector<std::string> &dst) { while (i + 1 < argc && !is_opt(argv[i + 1])) dst.push_back(argv[++i]); };
    auto ints = [&](std::vector<int> &dst) { std::vector<std::string> v; multi(v); for (auto &s : v) dst.push_back(atoi(s.c_str())); };
    auto one = [&]() -> std::string { if (i + 1 >= argc) die("missing value for " + o); return argv[++i]; };
    if (o == "-o" || o == "--output") output = one();
    else if (o == "-m" || o == "--mask") mask_name = one();
    else if (o == "-i" || o == "--input") multi(inputs);
    else if (o == "-t" || o == "--transformation") multi(tspecs);
    else if (o == "--thickness") { std::vector<std::string> v; multi(v); for (auto &s : v) thickness.push_back(atof(s.c_str())); }
    else if (o == "--useFullSlices") full_slices = true;
    else if (o == "--hierarchical") hierarchical = true;
    else if (o == "--resample") resample = true;
    else if (o == "--packages") ints(packages);
    else if (o == "--dilateMask") dilate = atoi(one().c_str());
    else if (o == "--existingReconTarget") existing_name = one();
    else if (o == "--patchSize") ints(psize);
    else if (o == "--patchStride") ints(pstride);
    else if (o == "--resolution") resolution = atof(one().c_str());
    else if (o == "--iterations") iterations = atoi(one().c_str());
    else if (o == "--sr_iterations") sr_iterations = atoi(one().c_str());
    else if (o == "--noMatchIntensities") no_matching = true;
    else if (o == "--coeffTable") coeff_table = true;                      // not a reference option: keep the PSF taps in HBM
    else if (o == "-d" || o == "--devices") ints(devices);
    else if (o == "--dumpProblem") dump_name = one();
    else if (o == "--dryRun") dry_run = true;
    else if (o == "--no_registration") no_registration = true;
    else if (o == "-s" || o == "--superpixel") superpixel = true;
    else if (o == "--spxSize") spx_size = atoi(one().c_str());
    else if (o == "--spxExtend") spx_extend = atoi(one().c_str());
    else if (o == "-h" || o == "--help") {
      printf("usage: PVRreconstructionGPU -o <volume> -i <stack_1> .. <stack_N> -m <mask> [-t id|<dof>|<4x4.txt> ..]\n"
             "       [--thickness th_1 ..] [--patchSize 32 32] [--patchStride 16 16] [--resolution 0.75] [--iterations 7]\n"
             "       [--sr_iterations 7] [--noMatchIntensities] [--no_registration] [-s [--spxSize 16] [--spxExtend 50]] [--useFullSlices] [-d device]\n");
      return 0;
    } else {
      die("option " + o + " is not supported by this build (see csrc/pvr_cli.cpp)");
    }
  }
  if (output.empty() || inputs.empty() || mask_name.empty()) die("-o, -i and -m are required (try --help)");
  if (psize.empty()) psize = {32, 32};
  if (pstride.empty()) pstride = {16, 16};
  if (psize.size() != 2 || pstride.size() != 2 || psize[0] < 1 || psize[1] < 1 || pstride[0] < 1 || pstride[1] < 1)
    die("--patchSize and --patchStride take two positive integers");
  size_t n = inputs.size();
  if (tspecs.empty()) tspecs.assign(n, "id");
  if (tspecs.size() != n) die("one transformation per stack expected");

  StageClock clk;
  // the HIP runtime comes up on a second thread while the stacks are read: a first context (the stack-to-stack registration
  // makes one, the reconstruction another) then costs milliseconds instead of 75
  std::future<void> runtime_up = std::async(std::launch::async, [&] {
    svr_ctx *warm = nullptr;
    if (!dry_run && svr_create(devices.empty() ? 0 : devices[0], &warm) == 0 && warm) svr_destroy(warm);
  });
  before_exit = [&] { if (runtime_up.valid()) runtime_up.wait(); };
  // ---- set-up (pvrmain:184-257, PBR.cpp:193-310) --------------------------------------------------------
  std::vector<Image> stacks;
  std::vector<M4> ts;
  stacks.resize(n);
  parallel_for((int)n, [&](int k) { stacks[k] = read_image(inputs[k]); });                        // gunzip is serial per file
  for (size_t k = 0; k < n; ++k) ts.push_back(load_transformation(tspecs[k]));
  clk.mark("read stacks");
  std::vector<double> half_thickness;                    // m_thickness: dz, or the given thickness / 2 (pvrmain:209-217)
  if (thickness.empty()) for (auto &s : stacks) half_thickness.push_back(s.a.dz);
  else { if (thickness.size() != n) die("one thickness per stack expected"); for (double t : thickness) half_thickness.push_back(t / 2.0); }
  size_t tmpl = 0;
  for (size_t k = 0; k < n; ++k) if (tspecs[k] == "id") { tmpl = k; break; }
  if (packages.size() == n) {                            // setImageStacks, PBR.cpp:134-146: every package becomes a stack of its own;
    std::vector<Image> ps;                               // m_template_num keeps indexing the new list
    std::vector<M4> pt;
    std::vector<double> ph;
    for (size_t k = 0; k < n; ++k) {
      if (packages[k] < 1) die("--packages takes positive integers");
      for (Image &p : split_packages(stacks[k], packages[k])) { ps.push_back(std::move(p)); pt.push_back(ts[k]); ph.push_back(half_thickness[k]); }
    }
    stacks.swap(ps); ts.swap(pt); half_thickness.swap(ph);
    n = stacks.size();
  }
  Image mask = read_image(mask_name);
  for (double &v : mask.d) v = ((long long)v == 0) ? 0.0 : 1.0;                               // PBR.cpp:201-209
  if (dilate > 0) dilate_mask(mask, dilate);                                                     // :212-223
  for (size_t k = 0; k < n; ++k) {                                                               // :229-236
    const Image m = transform_nn(mask, stacks[k].a, ts[k], 0.0);
    stacks[k] = crop_image(stacks[k], m);
    if (resample) stacks[k] = resample_bspline(stacks[k], resolution);                          // :237-246
  }
  clk.mark("mask, crop (resample)");
  const Image iso_mask = transform_nn(mask, resample_attr(mask.a, resolution), ident(), 0.0);    // :258-266
  clk.mark("isotropic mask");
  if (runtime_up.valid()) runtime_up.wait();
  if (!no_registration && n > 1) {                       // irtkStack3D3DRegistration<T>::run, :280-285
    svr_ctx *rctx = nullptr;
    if (svr_create(devices.empty() ? 0 : devices[0], &rctx) || !rctx) die("no usable HIP device (svr_create failed)");
    std::vector<svr_image_attr> at(n);
    std::vector<const double *> ptr(n);
    std::vector<double> tm(16 * n);
    for (size_t k = 0; k < n; ++k) { at[k] = stacks[k].a; ptr[k] = stacks[k].d.data(); for (int q = 0; q < 16; ++q) tm[16 * k + q] = ts[k].m[q]; }
    long evals = 0;
    char e[256] = {0};
    if (svrh_stack_registrations(rctx, nullptr, (int)n, at.data(), ptr.data(), tm.data(), (int)tmpl, &iso_mask.a, iso_mask.d.data(),
                                 SVRH_STACKREG_KEEP_ORIGIN, &evals, e))
      die(std::string("stack registration: ") + e);
    for (size_t k = 0; k < n; ++k) for (int q = 0; q < 16; ++q) ts[k].m[q] = tm[16 * k + q];
    fprintf(stderr, "stack-to-stack registration: %ld similarity evaluations\n", evals);
    svr_destroy(rctx);
  }
  clk.mark("registration of the stacks");
  if (!no_matching) match_stack_intensities_pvr(stacks, ts, iso_mask);                           // :288-294
  clk.mark("match stack intensities");
  float vmin = 3.402823466e38f, vmax = 1.175494351e-38f;                                         // computeMinMaxIntensities :792-814
  for (const Image &s : stacks)
    for (double v : s.d)
      if (v > 0) { vmax = std::max(vmax, (float)v); vmin = std::min(vmin, (float)v); }
  svr_image_attr tattr = resample_attr(stacks[tmpl].a, resolution);                              // CreateTemplate :941-965
  std::vector<float> existing;                           // setExistingReconstructionTarget :185-191: the volume and its grid
  if (!existing_name.empty()) {
    const Image ex = read_image(existing_name);
    tattr = ex.a;
    existing.assign(ex.d.begin(), ex.d.end());
  }
  const Image recon_mask = transform_nn(iso_mask, tattr, ts[tmpl], 0.0);                         // :303-304
  clk.mark("template, reconstruction mask");
  if (superpixel && full_slices) die("--superpixel with --useFullSlices is not supported by this build");
  if (hierarchical && full_slices) hierarchical = false;                                         // pvrmain:282-285 "SVR ON"
  bool first_level = true;

  // one irtkPatchBasedReconstruction<T>::run from the patch extraction on (the set-up above gives the same result every time)
  auto run_level = [&](const std::vector<int> &psize, const std::vector<int> &pstride, int iterations, const std::vector<float> &existing,
                       std::vector<float> &vol_out) -> bool {
  // ---- patches (PBR.cpp:385-399) -----------------------------------------------------------------------
  int px = psize[0], py = psize[1];
  if (full_slices) {                                     // the patch is the slice; stacks of different sizes share a grid padded with -1
    px = py = 0;
    for (size_t k = 0; k < n; ++k) { px = std::max(px, stacks[k].a.nx); py = std::max(py, stacks[k].a.ny); }
  }
  Patches P;
  std::vector<char> spx_masks;
  std::vector<int> counts;
  std::vector<float> st, sti, dims;
  std::vector<double> Td;                                // the registrators' m_transformations, one per patch
  for (size_t k = 0; k < n; ++k) {
    const int before = P.n;
    if (superpixel) {                                    // pvrmain:291-296: patch size = spxSize, stride = spxExtend
      SpxPatches sp;
      slic_superpixel_patches(stacks[k], half_thickness[k], iso_mask, spx_size, spx_size, spx_extend, sp);
      if (P.n > 0 && (sp.px != px || sp.py != py)) die("superpixel patches of different sizes (slices narrower than 64 pixels differ between the stacks)");
      px = sp.px; py = sp.py;
      P.data.insert(P.data.end(), sp.data.begin(), sp.data.end());
      P.i2w.insert(P.i2w.end(), sp.i2w.begin(), sp.i2w.end()); P.w2i.insert(P.w2i.end(), sp.w2i.begin(), sp.w2i.end());
      P.ri2w.insert(P.ri2w.end(), sp.ri2w.begin(), sp.ri2w.end()); P.mo.insert(P.mo.end(), sp.mo.begin(), sp.mo.end());
      P.invmo.insert(P.invmo.end(), sp.invmo.begin(), sp.invmo.end());
      P.attr.insert(P.attr.end(), sp.attr.begin(), sp.attr.end());
      spx_masks.insert(spx_masks.end(), sp.masks.begin(), sp.masks.end());
      P.n += sp.n;
    } else if (full_slices) {                            // patchBasedObject.cuh:183-189: size = the slice, stride = size + 1
      const int nx = stacks[k].a.nx, ny = stacks[k].a.ny;
      Patches one;
      generate_2d_patches(stacks[k], half_thickness[k], iso_mask, nx, ny, nx + 1, ny + 1, one);
      for (int q = 0; q < one.n; ++q) {
        std::vector<float> padded((size_t)px * py, -1.0f);
        for (int j = 0; j < ny; ++j) memcpy(&padded[(size_t)j * px], &one.data[((size_t)q * ny + j) * nx], nx * sizeof(float));
        P.data.insert(P.data.end(), padded.begin(), padded.end());
      }
      P.i2w.insert(P.i2w.end(), one.i2w.begin(), one.i2w.end()); P.w2i.insert(P.w2i.end(), one.w2i.begin(), one.w2i.end());
      P.ri2w.insert(P.ri2w.end(), one.ri2w.begin(), one.ri2w.end()); P.mo.insert(P.mo.end(), one.mo.begin(), one.mo.end());
      P.invmo.insert(P.invmo.end(), one.invmo.begin(), one.invmo.end());
      P.attr.insert(P.attr.end(), one.attr.begin(), one.attr.end());
      P.n += one.n;
    } else {
      generate_2d_patches(stacks[k], half_thickness[k], iso_mask, px, py, pstride[0], pstride[1], P);
    }
    counts.push_back(P.n - before);
    float t[16], ti[16];
    to_f16(ts[k], t); to_f16(inverse_rigid_or_affine(ts[k]), ti);
    for (int q = before; q < P.n; ++q) {
      st.insert(st.end(), t, t + 16); sti.insert(sti.end(), ti, ti + 16);
      Td.insert(Td.end(), ts[k].m, ts[k].m + 16);
      dims.push_back((float)stacks[k].a.dx); dims.push_back((float)stacks[k].a.dy); dims.push_back((float)stacks[k].a.dz);   // getDim()
    }
  }
  clk.mark("patch generation");
  const int ns = P.n;
  if (ns == 0) die("no patch overlaps the mask");
  fprintf(stderr, "%zu stacks, %d patches of %dx%d, volume %dx%dx%d at %g mm\n", n, ns, px, py, tattr.nx, tattr.ny, tattr.nz, resolution);

  if (!dump_name.empty() && first_level) {               // what the engine is about to receive, for the CPU tests
    FILE *f = fopen(dump_name.c_str(), "wb");
    if (!f) die("cannot write " + dump_name);
    const int hdr[8] = {ns, px, py, (int)n, tattr.nx, tattr.ny, tattr.nz, 1};   // [7] = 1: the geometry block at the end
    const float mm[2] = {vmin, vmax};
    std::vector<float> mf(recon_mask.d.begin(), recon_mask.d.end());
    fwrite(hdr, sizeof(int), 8, f); fwrite(counts.data(), sizeof(int), n, f); fwrite(mm, sizeof(float), 2, f);
    fwrite(P.data.data(), sizeof(float), P.data.size(), f); fwrite(P.i2w.data(), sizeof(float), P.i2w.size(), f);
    fwrite(mf.data(), sizeof(float), mf.size(), f);
    if (superpixel) fwrite(spx_masks.data(), 1, spx_masks.size(), f);
    {
      // everything else svr_set_slice_matrices / svr_set_slice_dims / svr_init_reconstruction_volume receive: per patch
      // w2i, T, Tinv, dims; the volume's image-to-world / world-to-image matrices and voxel size
      float gi2w[16], gw2i[16];
      to_f16(image_to_world(tattr), gi2w); to_f16(world_to_image(tattr), gw2i);
      const float gdim[3] = {(float)tattr.dx, (float)tattr.dy, (float)tattr.dz};
      fwrite(P.w2i.data(), sizeof(float), P.w2i.size(), f); fwrite(st.data(), sizeof(float), st.size(), f);
      fwrite(sti.data(), sizeof(float), sti.size(), f); fwrite(dims.data(), sizeof(float), dims.size(), f);
      fwrite(gi2w, sizeof(float), 16, f); fwrite(gw2i, sizeof(float), 16, f); fwrite(gdim, sizeof(float), 3, f);
    }
    fclose(f);
  }
  if (dry_run) return false;
  first_level = false;

  // ---- ranks: one engine context per device of -d.  Rank r takes the r-th of nr work-balanced segments of EVERY stack's patches
  // (svr_shard.h spatial_order: a rank's patches are neighbours in space), work = the pixels that carry data, weighted by orientation.
  // From here on the per-patch arrays are in the SHARDED numbering (rank after rank); order[k] = patch k's index in the global numbering
  // (stack after stack), which the host object keeps using where the reference's arithmetic depends on it (pvrh_set_unit_order).  The
  // reference's patch-based path is single-GPU (patchBasedReconMain.cpp:78,177-179, irtkPatchBasedReconstruction.cpp:402); SURVEY 8e:
  // "identical with patches as the unit"
  const int nr = (int)std::max<size_t>(1, devices.size());
  if (ns < nr) die("fewer patches than devices");
  std::vector<int> rlo(nr, 0), rhi(nr, ns), order;
  if (nr > 1) {
    std::vector<double> work(ns, 0.0);
    const size_t pp = (size_t)px * py;
    parallel_for(ns, [&](int q) {
      long c = 0;
      for (size_t i = 0; i < pp; ++i) c += P.data[(size_t)q * pp + i] > 0.0f;
      // a patch whose normal is the volume's x axis costs more per pixel (its runs span a band of centre planes, csrc/svr_cell.inc;
      // measured per rank: tools/shard_probe.py, sharding.py slice_cost_weights): x (1 + 0.2 n_x^2)
      const M4 rw = world_to_image(tattr);
      double nw[3], nt[3], nv[3], len = 0;
      for (int k = 0; k < 3; ++k) nw[k] = P.i2w[16 * (size_t)q + 4 * k + 2];
      for (int k = 0; k < 3; ++k) nt[k] = st[16 * (size_t)q + 4 * k] * nw[0] + st[16 * (size_t)q + 4 * k + 1] * nw[1] + st[16 * (size_t)q + 4 * k + 2] * nw[2];
      for (int k = 0; k < 3; ++k) { nv[k] = rw.m[4 * k] * nt[0] + rw.m[4 * k + 1] * nt[1] + rw.m[4 * k + 2] * nt[2]; len += nv[k] * nv[k]; }
      const double ax2 = len > 0 ? nv[0] * nv[0] / len : 0.0;
      work[q] = (double)c * (1.0 + 0.2 * ax2);
    });
    std::vector<int> stack_of;
    for (size_t k = 0; k < counts.size(); ++k) stack_of.insert(stack_of.end(), counts[k], (int)k);
    svr::spatial_order(work, stack_of, nr, order, rlo, rhi);
    for (int r = 0; r < nr; ++r) if (rhi[r] <= rlo[r]) die("a device would get no patch: fewer devices, please");
    svr::permute_rows(P.data, pp, order);
    svr::permute_rows(P.i2w, 16, order); svr::permute_rows(P.w2i, 16, order); svr::permute_rows(P.attr, 1, order);
    if (superpixel) svr::permute_rows(spx_masks, 4096, order);
    svr::permute_rows(st, 16, order); svr::permute_rows(sti, 16, order); svr::permute_rows(dims, 3, order); svr::permute_rows(Td, 16, order);
  }
  std::vector<svr_ctx *> ctxs(nr, nullptr);
  std::vector<pvrh_recon *> hosts(nr, nullptr);
  for (int r = 0; r < nr; ++r)
    if (svr_create(devices.empty() ? 0 : devices[r], &ctxs[r]) || !ctxs[r])
      die("no usable HIP device " + std::to_string(devices.empty() ? 0 : devices[r]) + " (svr_create failed)");
  svr_ctx *ctx = ctxs[0];
  svr_group *group = nr > 1 ? svr_group_create(nr, devices.data()) : nullptr;
  if (nr > 1 && !group) die("cannot set up the rank group (librccl not found?)");
  auto par = [&](const std::function<void(int)> &fn) {   // every rank in its own thread (the collectives block until all have arrived)
    if (nr == 1) { fn(0); return; }
    std::vector<std::thread> th;
    for (int r = 0; r < nr; ++r) th.emplace_back(fn, r);
    for (auto &t : th) t.join();
  };
#define ENGR(r, call) do { int rc_ = (call); if (rc_) die(std::string(#call) + ": " + svr_last_error(ctxs[r])); } while (0)
#define PVRHR(r, call) do { int rc_ = (call); if (rc_) die(std::string(#call) + ": " + pvrh_last_error(hosts[r])); } while (0)

  // ---- upload (the engine in PVR mode; m_quality_factor = 1, PBR.cpp:415), per rank --------------------------------
  const uint32_t vsize[3] = {(uint32_t)tattr.nx, (uint32_t)tattr.ny, (uint32_t)tattr.nz};
  const float vdim[3] = {(float)tattr.dx, (float)tattr.dy, (float)tattr.dz};
  std::vector<float> maskf(recon_mask.d.begin(), recon_mask.d.end());
  float ri2w[16], rw2i[16];
  to_f16(image_to_world(tattr), ri2w); to_f16(world_to_image(tattr), rw2i);
  if (!existing.empty() && existing.size() != (size_t)tattr.nx * tattr.ny * tattr.nz) die("existing reconstruction target of the wrong size");
  float pi2w[16], pw2i[16];
  {
    svr_image_attr pa;
    memset(&pa, 0, sizeof(pa));
    pa.nx = pa.ny = pa.nz = 128; pa.dx = tattr.dx; pa.dy = tattr.dy; pa.dz = tattr.dz;
    pa.xaxis[0] = pa.yaxis[1] = pa.zaxis[2] = 1.0;
    to_f16(image_to_world(pa), pi2w); to_f16(world_to_image(pa), pw2i);
  }
  auto set_matrices = [&](int r) {
    const size_t o = 16 * (size_t)rlo[r];
    ENGR(r, svr_set_slice_matrices(ctxs[r], st.data() + o, sti.data() + o, P.i2w.data() + o, P.w2i.data() + o, P.i2w.data() + o, P.w2i.data() + o, ri2w, rw2i));
  };
  par([&](int r) {
    const int nl = rhi[r] - rlo[r];
    const size_t o = (size_t)rlo[r];
    ENGR(r, svr_set_option(ctxs[r], "pvr", 1));
    if (coeff_table) ENGR(r, svr_set_option(ctxs[r], "coeff_table", 1));
    ENGR(r, svr_set_option(ctxs[r], "tune_tiles", 32768));                 // a run is a few dozen PSF launches: cheap tuning trials
    ENGR(r, svr_init_reconstruction_volume(ctxs[r], vsize, vdim, existing.empty() ? nullptr : existing.data(), 12.0f));   // copyFromHost :310-314
    ENGR(r, svr_set_mask(ctxs[r], vsize, vdim, maskf.data(), 12.0f));
    const uint32_t ssize[3] = {(uint32_t)px, (uint32_t)py, (uint32_t)nl};
    std::vector<int> sizes_x(nl, px), sizes_y(nl, py);
    ENGR(r, svr_init_storage_volumes(ctxs[r], ssize, &dims[3 * o]));
    ENGR(r, svr_fill_slices(ctxs[r], P.data.data() + o * px * py, sizes_x.data(), sizes_y.data()));
    if (superpixel) ENGR(r, svr_set_spx_masks(ctxs[r], spx_masks.data() + o * 4096));
    ENGR(r, svr_set_slice_dims(ctxs[r], dims.data() + 3 * o, 1.0f));
    const uint32_t psz[3] = {128, 128, 128};
    ENGR(r, svr_generate_psf_volume(ctxs[r], nullptr, psz, &dims[3 * o], vdim, pi2w, pw2i, 1.0f));
    set_matrices(r);
    hosts[r] = pvrh_create_sharded(ctxs[r], counts.data(), (int)counts.size(), vmin, vmax, rlo[r], rhi[r],
                                   group ? svr_group_join(group, r, ctxs[r]) : nullptr);
    if (!hosts[r]) die("pvrh_create_sharded failed" + std::string(group ? " (the rank group could not be joined)" : ""));
    if (nr > 1 && pvrh_set_unit_order(hosts[r], order.data())) die("pvrh_set_unit_order failed");
  });
  if (nr > 1)
    fprintf(stderr, "%d ranks on devices%s, patches per rank%s, collectives: %s\n", nr,
            [&] { std::string t; for (int d : devices) t += " " + std::to_string(d); return t; }().c_str(),
            [&] { std::string t; for (int r = 0; r < nr; ++r) t += " " + std::to_string(rhi[r] - rlo[r]); return t; }().c_str(),
            svr_group_uses_rccl(group) ? "RCCL" : "host memory (a device is named more than once: test mode)");

  clk.mark("engine set-up and upload");
  // ---- the loop (PBR.cpp:445-593) ----------------------------------------------------------------------------
  pvrh_recon *host = hosts[0];
  for (int it = 0; it < iterations + 1; ++it) {
    const bool have_volume = it > 0 || !existing.empty();                                      // PBR.cpp:456
    if (have_volume && !no_registration && superpixel) {
      // runHybrid registers the square CPU patches of generatePatchesCPU and updateTransformationMatrices then reads one
      // transformation per GPU patch from that shorter list: undefined in the reference for superpixel patches, not done
      fprintf(stderr, "superpixel mode: the patch-to-volume registration is skipped\n");
    } else if (have_volume && !no_registration) {        // PBR.cpp:452-489: runHybrid, the IRTK schedule on every patch
      std::vector<float> vol((size_t)tattr.nx * tattr.ny * tattr.nz);
      ENG(svr_sync_cpu(ctx, vol.data()));                // m_GPURecon.copyToHost (every rank holds the same volume)
      std::vector<long> evals(nr, 0);
      par([&](int r) {                                   // every rank registers its own patches against the shared volume
        char e[256] = {0};
        const size_t o = (size_t)rlo[r];
        if (svrh_slice_to_volume_registration(ctxs[r], nullptr, rhi[r] - rlo[r], P.data.data() + o * px * py, px, py, P.attr.data() + o,
                                              Td.data() + 16 * o, &tattr, vol.data(), SVRH_S2V_NO_RESAMPLE, &evals[r], e))
          die(std::string("patch-to-volume registration: ") + e);
      });
      for (int q = 0; q < ns; ++q) {                     // updateTransformationMatrices (patchBasedObject.cuh:151-169)
        M4 t;
        for (int k = 0; k < 16; ++k) t.m[k] = Td[16 * (size_t)q + k];
        to_f16(t, &st[16 * (size_t)q]); to_f16(inverse_rigid_or_affine(t), &sti[16 * (size_t)q]);
      }
      for (int r = 0; r < nr; ++r) set_matrices(r);
      long total = 0;
      for (long v : evals) total += v;
      fprintf(stderr, "patch-to-volume registration: %ld similarity evaluations\n", total);
    }
    if (have_volume && !no_registration) clk.mark("registration of the patches");
    par([&](int r) { PVRHR(r, pvrh_reconstruct_iteration(hosts[r], sr_iterations)); });
    clk.mark("reconstruction iteration");
    double sc[8];
    pvrh_get_state(host, nullptr, nullptr, nullptr, sc);
    fprintf(stderr, "iteration %d: sigma %.4g mix %.3f\n", it, sc[0], sc[1]);
  }
  vol_out.resize((size_t)tattr.nx * tattr.ny * tattr.nz);
  ENG(svr_sync_cpu(ctx, vol_out.data()));
  if (clk.on) {
    int o[6] = {0, 0, 0, 0, 0, 0};
    const char *names[6] = {"tile_w", "tile_h", "wave_cap", "fwd_tile_w", "fwd_tile_h", "fwd_unit_cap"};
    for (int k = 0; k < 6; ++k) (void)svr_get_option(ctx, names[k], &o[k]);
    fprintf(stderr, "[timing] tuned: scatter tiles %dx%d box %d, gather tiles %dx%d box %d\n", o[0], o[1], o[2], o[3], o[4], o[5]);
  }
  clk.mark("volume download");
  for (int r = 0; r < nr; ++r) pvrh_destroy(hosts[r]);
  svr_group_destroy(group);
  for (int r = 0; r < nr; ++r) svr_destroy(ctxs[r]);
  return true;
  };

  std::vector<float> vol;
  if (!hierarchical) {
    if (!run_level(psize, pstride, iterations, existing, vol)) return 0;
  } else {
    // pvrmain:359-432: iterations + 1 levels of one registration-reconstruction iteration each; a level starts from the volume of
    // the level before (its "reconimage1_<size>_<stride>.nii.gz") and cuts patches 4 pixels smaller (stride 2 smaller, not for superpixels)
    std::vector<int> ps = psize, pt = pstride;
    for (int level = 0; level <= iterations; ++level) {
      if (superpixel ? spx_size < 1 : (ps[0] < 1 || ps[1] < 1 || pt[0] < 1 || pt[1] < 1))
        die("hierarchical mode: the patch size reached zero at level " + std::to_string(level) + " (--patchSize - 4 * --iterations must stay positive)");
      fprintf(stderr, "hierarchical level %d: patch size %d stride %d\n", level, superpixel ? spx_size : ps[0], superpixel ? spx_extend : pt[0]);
      std::vector<float> next;
      if (!run_level(ps, pt, 1, existing, next)) return 0;
      existing = next;
      vol.swap(next);
      if (superpixel) spx_size -= 4;
      else { ps[0] -= 4; ps[1] -= 4; pt[0] -= 2; pt[1] -= 2; }
    }
  }
  clk.mark("engine teardown");
  char err[256] = {0};
  if (svr_nifti_write(output.c_str(), &tattr, vol.data(), err)) die(output + ": " + err);
  clk.mark("write the volume");
  return 0;
}
