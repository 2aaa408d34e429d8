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
// Transformations (-t) are `id`, IRTK rigid `dof` files or 4x4 text matrices; like the reference they are the start of the
// stack-to-stack registration (StackRegistrations, before and after the other stacks are cropped, main.cc:661,711).
// Slice-to-volume registration is the reference's default IRTK schedule (csrc/irtk_reg.cpp, every similarity evaluation
// on the GPU) or, with --useGPUReg, the reference's GPU registration.  --no_registration (not a reference option) skips
// both.  --packages runs PackageToVolume with the schedule of main.cc:832-864; --tfolder reads transformation<i>.dof per
// slice and --debug writes them next to the output.  Not built, refused loudly: patch/superpixel
// modes, the CPU reconstruction path.
#include <functional>
#include <thread>

#include <future>

#include "svr_prep.h"
#include "svr_shard.h"


int main(int argc, char **argv) {
  std::string output, mask_name, tfolder, sfolder;
  bool debug = false, dry_run = false, save_slice_transformations = false;
  std::string dump_name;
  std::vector<std::string> inputs, tspecs;
  std::vector<double> thickness;
  std::vector<int> force_excluded, devices, packages;
  int iterations = 4, levels = 3, rec_first = 4, rec_last = 13, num_stacks_tuner = 0;
  double resolution = 0.75, average = 700, delta = 150, lambda = 0.02, last_lambda = 0.01, smooth_mask = 4;
  bool no_matching = false, use_gpu_reg = false, no_registration = false;
  int coeff_table = -1;                                                   // -1: the engine's default (on since round 6), 1 / 0: --coeffTable / --noCoeffTable
  // ---- options (main.cc:164-211) ---------------------------------------------------------------------
  auto is_opt = [](const char *s) { return s[0] == '-' && !(s[1] >= '0' && s[1] <= '9') && s[1] != '.'; };
  for (int i = 1; i < argc; ++i) {
    const std::string o = argv[i];
    auto multi = [&](std::vector<std::string> &dst) { while (i + 1 < argc && !is_opt(argv[i + 1])) dst.push_back(argv[++i]); };
    auto one = [&]() -> std::string { if (i + 1 >= argc) die("missing value for " + o); return argv[++i]; };
    // po::value<bool> options of the reference take a value (--debug 1); a bare flag is accepted as `true`
    auto opt_bool = [&](bool bare) -> bool {
      if (i + 1 < argc) {
        const std::string v = argv[i + 1];
        if (v == "1" || v == "true" || v == "yes" || v == "on") { ++i; return true; }
        if (v == "0" || v == "false" || v == "no" || v == "off") { ++i; return false; }
      }
      return bare;
    };
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
    else if (o == "--no_intensity_matching") no_matching = !opt_bool(false);   // the value lands in `intensity_matching` (main.cc:186): 0 switches it off
    else if (o == "--num_stacks_tuner") num_stacks_tuner = atoi(one().c_str());
    else if (o == "--log_prefix" || o == "--low_intensity_cutoff" || o == "--patchSize" || o == "--patchStride") (void)one();   // no log files; bias / patch modes are off
    else if (o == "--no_log" || o == "--global_bias_correction") (void)opt_bool(true);
    else if (o == "--force_exclude") { std::vector<std::string> v; multi(v); for (auto &s : v) force_excluded.push_back(atoi(s.c_str())); }
    else if (o == "--rec_iterations_first") rec_first = atoi(one().c_str());
    else if (o == "--rec_iterations_last") rec_last = atoi(one().c_str());
    else if (o == "--useGPUReg") use_gpu_reg = true;
    else if (o == "-p" || o == "--packages") { std::vector<std::string> v; multi(v); for (auto &x : v) packages.push_back(atoi(x.c_str())); }
    else if (o == "--no_registration") no_registration = true;
    else if (o == "--tfolder") tfolder = one();
    else if (o == "--sfolder") sfolder = one();
    else if (o == "--coeffTable") coeff_table = 1;                        // not a reference option: keep the PSF taps in HBM (CoeffInit on the GPU path); the default since round 6
    else if (o == "--noCoeffTable") coeff_table = 0;                      // ... every tap evaluated in every pass, like the reference's GPU kernels: same volume, bit for bit
    else if (o == "--debug") debug = opt_bool(true);
    else if (o == "--saveSliceTransformations") save_slice_transformations = true;   // main.cc:211, 1213-1217
    else if (o == "--dumpProblem") dump_name = one();                     // test hooks: what the engine is about to receive [--dryRun: stop there]
    else if (o == "--dryRun") dry_run = true;
    else if (o == "--useCPUReg" || o == "--disableBiasCorrection" || o == "--debug_gpu") {}
    else if (o == "-d" || o == "--devices") { std::vector<std::string> v; multi(v); for (auto &s : v) devices.push_back(atoi(s.c_str())); }
    else if (o == "-h" || o == "--help") {
      printf("usage: SVRreconstructionGPU -o <volume> -i <stack_1> .. <stack_N> [-m <mask>] [-t id|<4x4.txt> ..] [--thickness th_1 ..]\n"
             "       [--iterations 4] [--resolution 0.75] [--multires 3] [--average 700] [--delta 150] [--lambda 0.02]\n"
             "       [--lastIterLambda 0.01] [--smooth_mask 4] [--no_intensity_matching] [--force_exclude i ..]\n"
             "       [--rec_iterations_first 4] [--rec_iterations_last 13] [--packages p_1 ..] [--useGPUReg] [--no_registration] [--tfolder dir] [--sfolder dir]\n"
             "       [--saveSliceTransformations] [--coeffTable | --noCoeffTable] [-d device_1 .. device_N]\n");
      return 0;
    } else {
      die("option " + o + " is not supported by this build (see csrc/svr_cli.cpp)");
    }
  }
  if (output.empty() || inputs.empty()) die("-o and -i are required (try --help)");
  if (num_stacks_tuner > 0 && (size_t)num_stacks_tuner < inputs.size()) {      // main.cc:406-419: only the first stacks are used
    inputs.resize(num_stacks_tuner);
    if (tspecs.size() > (size_t)num_stacks_tuner) tspecs.resize(num_stacks_tuner);
    if (thickness.size() > (size_t)num_stacks_tuner) thickness.resize(num_stacks_tuner);
    if (packages.size() > (size_t)num_stacks_tuner) packages.resize(num_stacks_tuner);
  }
  const size_t n = inputs.size();
  if (tspecs.empty()) tspecs.assign(n, "id");
  if (tspecs.size() != n) die("one transformation per stack expected");
  if (!packages.empty() && packages.size() != n) die("one package count per stack expected");

  size_t tmpl = n;
  for (size_t k = 0; k < n; ++k) if (tspecs[k] == "id") { tmpl = k; break; }
  if (tmpl == n) die("Please identify the template by assigning id transformation.");          // main.cc:452-457

  StageClock clk;
  // the HIP runtime and the context come up (75 ms) while the stacks are read and cropped; first use: the stack registrations
  svr_ctx *ctx = nullptr;
  std::future<int> ctx_ready = std::async(std::launch::async, [&] { return dry_run ? 1 : svr_create(devices.empty() ? 0 : devices[0], &ctx); });
  before_exit = [&] { if (ctx_ready.valid()) ctx_ready.wait(); };     // an error while reading must not exit under the runtime's feet
  auto need_ctx = [&] {
    if (ctx_ready.valid() && (ctx_ready.get() || !ctx)) die("no usable HIP device (svr_create failed)");
  };

  // ---- set-up (main.cc:386-815) ----------------------------------------------------------------------
  std::vector<Image> stacks;
  std::vector<M4> ts;
  stacks.resize(n);
  parallel_for((int)n, [&](int k) { stacks[k] = read_image(inputs[k]); });                        // gunzip is serial per file
  for (size_t k = 0; k < n; ++k) ts.push_back(load_transformation(tspecs[k]));
  clk.mark("read stacks");
  if (thickness.empty()) for (auto &s : stacks) thickness.push_back(2.0 * s.a.dz);            // main.cc:422-431
  if (thickness.size() != n) die("one thickness per stack expected");
  Image mask_img;
  const bool have_mask = !mask_name.empty() || sfolder.empty();          // main.cc:461: no mask is made up when --sfolder is given
  if (!mask_name.empty()) {
    mask_img = read_image(mask_name);
  } else if (have_mask) {
    // no mask given: CreateMask(stacks[templateNumber]) binarises the template stack (> 0), in case it was padded; the
    // normal mask path follows (main.cc:458-480, RG.cc:736-748)
    mask_img = stacks[tmpl];
    for (auto &v : mask_img.d) v = v > 0.0 ? 1.0 : 0.0;
  }
  if (have_mask) {
    const Image m = transform_nn(mask_img, stacks[tmpl].a, ts[tmpl], 0.0);                     // TransformMask RG.cc:805-821
    stacks[tmpl] = crop_image(stacks[tmpl], m);
  }
  const svr_image_attr tattr = create_template(stacks[tmpl].a, resolution);
  const Image vol_mask = set_mask(tattr, have_mask ? &mask_img : nullptr, smooth_mask);
  clk.mark("mask, crop, template");
  auto stack_registrations = [&]() {                                                             // StackRegistrations, RG.cc:849-1001
    if (no_registration || n < 2 || !sfolder.empty()) return;                                    // main.cc:658, 708
    if (dry_run) die("--dryRun makes no engine context and cannot register the stacks: add --no_registration (or --sfolder)");
    need_ctx();
    std::vector<svr_image_attr> at(n);
    std::vector<const double *> ptr(n);
    std::vector<double> tm(16 * n);
    for (size_t k = 0; k < n; ++k) { at[k] = stacks[k].a; ptr[k] = stacks[k].d.data(); for (int q = 0; q < 16; ++q) tm[16 * k + q] = ts[k].m[q]; }
    long evals = 0;
    char e[256] = {0};
    if (svrh_stack_registrations(ctx, nullptr, (int)n, at.data(), ptr.data(), tm.data(), (int)tmpl, have_mask ? &vol_mask.a : nullptr,
                                 have_mask ? vol_mask.d.data() : nullptr, 0, &evals, e))
      die(std::string("stack registration: ") + e);
    for (size_t k = 0; k < n; ++k) for (int q = 0; q < 16; ++q) ts[k].m[q] = tm[16 * k + q];
    fprintf(stderr, "stack-to-stack registration: %ld similarity evaluations\n", evals);
  };
  stack_registrations();                                                                         // main.cc:657-662
  for (size_t k = 0; k < n; ++k) {                                                               // main.cc:676-700
    if (k == tmpl) continue;
    const Image m = transform_nn(vol_mask, stacks[k].a, ts[k], 0.0);
    stacks[k] = crop_image(stacks[k], m);
  }
  stack_registrations();                                                                         // main.cc:707-713
  clk.mark("stack registrations, crops");
  const std::vector<float> factors = match_stack_intensities(stacks, ts, vol_mask, average, no_matching);
  clk.mark("match stack intensities");
  // CreateSlicesAndTransformations RG.cc:1835-1880 + MaskSlices RG.cc:1940-1988 + the packing of SyncGPU RG.cc:249-328
  struct SliceSrc { Image r; M4 t; int stack; };
  std::vector<SliceSrc> srcs;
  for (size_t k = 0; k < n; ++k)
    for (int j = 0; j < stacks[k].a.nz; ++j) {
      SliceSrc q{get_region(stacks[k], 0, 0, j, stacks[k].a.nx, stacks[k].a.ny, j + 1), ts[k], (int)k};
      q.r.a.dz = thickness[k];
      srcs.push_back(q);
    }
  if (!sfolder.empty()) {
    // replaceSlices, RG.cc:4767-4822: every file of the folder is one slice that is already in place -- identity
    // transformation, stack 0, 4 mm thickness, "equally many as loaded from stacks".  The reference takes the files in
    // directory_iterator order (unspecified); here they are taken in the order of their names.
    std::vector<std::string> files = list_directory(sfolder);
    if (files.size() != srcs.size())
      die("--sfolder: " + std::to_string(files.size()) + " files, but the stacks hold " + std::to_string(srcs.size()) + " slices");
    for (size_t i = 0; i < files.size(); ++i) {
      SliceSrc q{read_image(sfolder + "/" + files[i]), ident(), 0};
      if (q.r.a.nz != 1) die("--sfolder: " + files[i] + " is not a single slice");
      q.r.a.dz = 4.0;
      srcs[i] = q;
    }
  }
  int ns = (int)srcs.size(), mx = 0, my = 0;
  for (auto &q : srcs) { mx = std::max(mx, q.r.a.nx); my = std::max(my, q.r.a.ny); }
  std::vector<float> grid((size_t)ns * mx * my, -1.0f), i2w(16 * (size_t)ns), w2i(16 * (size_t)ns), st(16 * (size_t)ns),
      sti(16 * (size_t)ns), dims(3 * (size_t)ns);
  std::vector<int> sizes_x(ns), sizes_y(ns), stack_index(ns);
  std::vector<svr_image_attr> sattr(ns);
  std::vector<double> T(16 * (size_t)ns);
  const M4 mw2i = world_to_image(vol_mask.a);
  double vmin = 1e300, vmax = -1e300;
  for (int sl = 0; sl < ns; ++sl) {
      const Image &r = srcs[sl].r;
      const M4 &tk = srcs[sl].t;
      const M4 si2w = image_to_world(r.a);
      for (int y = 0; y < r.a.ny; ++y)
        for (int x = 0; x < r.a.nx; ++x) {
          double v = r.at(x, y, 0);
          if (v < 0.01) v = -1;
          double qx = x, qy = y, qz = 0;                   // ImageToWorld, Transform, WorldToImage: three applications (RG.cc:1961-1972)
          apply_point(si2w, qx, qy, qz); apply_point(tk, qx, qy, qz); apply_point(mw2i, qx, qy, qz);
          const long i = (long)irtk_round(qx), jj = (long)irtk_round(qy), kk = (long)irtk_round(qz);
          if (!(i >= 0 && i < vol_mask.a.nx && jj >= 0 && jj < vol_mask.a.ny && kk >= 0 && kk < vol_mask.a.nz) ||
              vol_mask.at((int)i, (int)jj, (int)kk) == 0)
            v = -1;
          grid[((size_t)sl * my + y) * mx + x] = (float)v;
          if (v > 0) { vmin = std::min(vmin, (double)(float)v); vmax = std::max(vmax, (double)(float)v); }
        }
      to_f16(si2w, &i2w[16 * (size_t)sl]); to_f16(world_to_image(r.a), &w2i[16 * (size_t)sl]);
      to_f16(tk, &st[16 * (size_t)sl]); to_f16(inverse_rigid_or_affine(tk), &sti[16 * (size_t)sl]);
      for (int q = 0; q < 16; ++q) T[16 * (size_t)sl + q] = tk.m[q];
      dims[3 * (size_t)sl] = (float)r.a.dx; dims[3 * (size_t)sl + 1] = (float)r.a.dy; dims[3 * (size_t)sl + 2] = (float)r.a.dz;
      sizes_x[sl] = r.a.nx; sizes_y[sl] = r.a.ny; stack_index[sl] = srcs[sl].stack; sattr[sl] = r.a;
  }
  if (!dump_name.empty()) {
    // what the engine is about to receive, for the CPU tests (tests/test_prep_oracle.py compares it with the oracle's restatement
    // of CreateTemplate / SetMask / TransformMask / CropImage / MatchStackIntensities / MaskSlices): header, the template's
    // attributes, the volume mask, the cropped stacks' attributes, the slice grid, the slices' transformations, the factors
    FILE *f = fopen(dump_name.c_str(), "wb");
    if (!f) die("cannot write " + dump_name);
    const int hdr[8] = {ns, mx, my, (int)n, tattr.nx, tattr.ny, tattr.nz, 2};
    fwrite(hdr, sizeof(int), 8, f);
    fwrite(&tattr, sizeof(svr_image_attr), 1, f);
    fwrite(vol_mask.d.data(), sizeof(double), vol_mask.d.size(), f);
    for (size_t k = 0; k < n; ++k) fwrite(&stacks[k].a, sizeof(svr_image_attr), 1, f);
    fwrite(grid.data(), sizeof(float), grid.size(), f);
    fwrite(T.data(), sizeof(double), T.size(), f);
    fwrite(factors.data(), sizeof(float), factors.size(), f);
    fwrite(sizes_x.data(), sizeof(int), sizes_x.size(), f);
    fwrite(sizes_y.data(), sizeof(int), sizes_y.size(), f);
    fclose(f);
  }
  if (dry_run) { fflush(nullptr); _exit(0); }             // (no context was made; skip the teardown of a runtime that never came up)
  if (!(vmax > 0)) die("no slice pixel lies inside the mask");
  fprintf(stderr, "%zu stacks, %d slices of up to %dx%d, volume %dx%dx%d at %g mm, stack factors", n, ns, mx, my, tattr.nx, tattr.ny, tattr.nz,
          resolution);
  for (float f : factors) fprintf(stderr, " %.9g", f);
  fprintf(stderr, "\n");

  // ---- ranks: one engine context per device of -d (main.cc:191).  Rank r takes the r-th of nr work-balanced segments of EVERY stack
  // (svr_shard.h spatial_order: a rank's slices are neighbours in space), work = estimated PSF work = active pixels x (9.4 + live
  // planes of the 16) x (1 + 0.2 n_x^2), see sharding.py slice_cost_weights.  (reconstruction_cuda2.cu:1413-1457 shards by slice count
  // in slice order and drops the remainder.)  From here on every per-slice array is in the SHARDED numbering -- rank after rank --
  // and `order[k]` is slice k's index in the reference's order: files named by slice number (--tfolder, --debug), --force_exclude
  // and the package registration, which works on whole stacks, go through it. -----------------------------------------------
  const int nr = (int)std::max<size_t>(1, devices.size());
  std::vector<int> rlo(nr, 0), rhi(nr, ns), order(ns), inv_order(ns);
  for (int s = 0; s < ns; ++s) order[s] = inv_order[s] = s;
  if (nr > 1) {
    if (ns < nr) die("fewer slices than devices");
    std::vector<double> work(ns, 0.0);
    const M4 rw = world_to_image(tattr);
    for (int s = 0; s < ns; ++s) {
      long c = 0;
      for (size_t i = 0; i < (size_t)mx * my; ++i) c += grid[(size_t)s * mx * my + i] != -1.0f;
      // slice normal in volume axes: reconW2I * T * sliceI2W applied to the slice's z direction
      double nw[3], nt[3], nv[3], len = 0;
      for (int k = 0; k < 3; ++k) nw[k] = i2w[16 * (size_t)s + 4 * k + 2];
      for (int k = 0; k < 3; ++k) nt[k] = T[16 * (size_t)s + 4 * k] * nw[0] + T[16 * (size_t)s + 4 * k + 1] * nw[1] + T[16 * (size_t)s + 4 * k + 2] * nw[2];
      for (int k = 0; k < 3; ++k) { nv[k] = rw.m[4 * k] * nt[0] + rw.m[4 * k + 1] * nt[1] + rw.m[4 * k + 2] * nt[2]; len += nv[k] * nv[k]; }
      len = std::max(sqrt(len), 1e-12);
      const double ax = fabs(nv[0]) / len, ay = fabs(nv[1]) / len, az = fabs(nv[2]) / len;
      const double ne = std::max(ay, az), no = std::min(ay, az), sigma = dims[3 * (size_t)s + 2] / 2.3548 / tattr.dx;
      const double live = std::min(16.0, 2.0 * (5.1 * sigma + 8.0 * (ax + no)) / std::max(ne, 1e-3) + 1.0);
      work[s] = (double)c * (9.4 + live) * (1.0 + 0.2 * ax * ax);
    }
    svr::spatial_order(work, stack_index, nr, order, rlo, rhi);
    for (int r = 0; r < nr; ++r) if (rhi[r] <= rlo[r]) die("a device would get no slice: fewer devices, please");
    for (int k = 0; k < ns; ++k) inv_order[order[k]] = k;
    svr::permute_rows(grid, (size_t)mx * my, order);
    svr::permute_rows(i2w, 16, order); svr::permute_rows(w2i, 16, order); svr::permute_rows(st, 16, order); svr::permute_rows(sti, 16, order);
    svr::permute_rows(dims, 3, order); svr::permute_rows(sizes_x, 1, order); svr::permute_rows(sizes_y, 1, order);
    svr::permute_rows(stack_index, 1, order); svr::permute_rows(sattr, 1, order); svr::permute_rows(T, 16, order);
    for (int &s : force_excluded) if (s >= 0 && s < ns) s = inv_order[s];
  }
  need_ctx();
  std::vector<svr_ctx *> ctxs(nr, nullptr);
  std::vector<svrh_recon *> hosts(nr, nullptr);
  ctxs[0] = ctx;
  for (int r = 1; r < nr; ++r)
    if (svr_create(devices[r], &ctxs[r]) || !ctxs[r]) die("no usable HIP device " + std::to_string(devices[r]) + " (svr_create failed)");
  svr_group *group = nr > 1 ? svr_group_create(nr, devices.data()) : nullptr;
  if (nr > 1 && !group) die("cannot set up the rank group (librccl not found?)");
  // every rank in its own thread (the collectives inside block until all ranks have arrived)
  auto par = [&](const std::function<void(int)> &fn) {
    if (nr == 1) { fn(0); return; }
    std::vector<std::thread> th;
    for (int r = 0; r < nr; ++r) th.emplace_back(fn, r);
    for (auto &t : th) t.join();
  };
#define ENGR(r, call) do { int rc_ = (call); if (rc_) die(std::string(#call) + ": " + svr_last_error(ctxs[r])); } while (0)
#define HOSTR(r, call) do { int rc_ = (call); if (rc_) die(std::string(#call) + ": " + svrh_last_error(hosts[r])); } while (0)
  float ri2w[16], rw2i[16];
  to_f16(image_to_world(tattr), ri2w); to_f16(world_to_image(tattr), rw2i);
  auto set_matrices = [&](int r) {
    const size_t o = 16 * (size_t)rlo[r];
    ENGR(r, svr_set_slice_matrices(ctxs[r], st.data() + o, sti.data() + o, i2w.data() + o, w2i.data() + o, i2w.data() + o, w2i.data() + o, ri2w, rw2i));
  };

  // ---- SyncGPU + generatePSFVolume + UpdateGPUTranformationMatrices (RG.cc:249-401, 1496-1610), per rank ----------
  const uint32_t vsize[3] = {(uint32_t)tattr.nx, (uint32_t)tattr.ny, (uint32_t)tattr.nz};
  const float vdim[3] = {(float)tattr.dx, (float)tattr.dy, (float)tattr.dz};
  std::vector<float> maskf(vol_mask.d.begin(), vol_mask.d.end());
  float pi2w[16], pw2i[16];
  {
    svr_image_attr pa;                                                   // PSF_SIZE 128 (RC.cuh:56), RG.cc:1534-1551
    memset(&pa, 0, sizeof(pa));
    pa.nx = pa.ny = pa.nz = 128; pa.dx = tattr.dx; pa.dy = tattr.dy; pa.dz = tattr.dz;
    pa.xaxis[0] = pa.yaxis[1] = pa.zaxis[2] = 1.0;
    to_f16(image_to_world(pa), pi2w); to_f16(world_to_image(pa), pw2i);
  }
  par([&](int r) {
    const int nl = rhi[r] - rlo[r];
    const size_t o = (size_t)rlo[r];
    if (coeff_table >= 0) ENGR(r, svr_set_option(ctxs[r], "coeff_table", coeff_table));
    ENGR(r, svr_set_option(ctxs[r], "tune_tiles", 32768));                // a run is a few dozen PSF launches: cheap tuning trials
    ENGR(r, svr_init_reconstruction_volume(ctxs[r], vsize, vdim, nullptr, 12.0f));
    ENGR(r, svr_set_mask(ctxs[r], vsize, vdim, maskf.data(), 12.0f));
    const uint32_t ssize[3] = {(uint32_t)mx, (uint32_t)my, (uint32_t)nl};
    ENGR(r, svr_init_storage_volumes(ctxs[r], ssize, &dims[3 * o]));
    ENGR(r, svr_fill_slices(ctxs[r], grid.data() + o * mx * my, sizes_x.data() + o, sizes_y.data() + o));
    ENGR(r, svr_set_slice_dims(ctxs[r], dims.data() + 3 * o, 2.0f));
    const uint32_t psize[3] = {128, 128, 128};
    ENGR(r, svr_generate_psf_volume(ctxs[r], nullptr, psize, &dims[3 * o], vdim, pi2w, pw2i, 2.0f));
    set_matrices(r);
    hosts[r] = svrh_create(ctxs[r], ns, rlo[r], rhi[r], group ? svr_group_join(group, r, ctxs[r]) : nullptr);
    if (!hosts[r]) die("svrh_create failed");
    svrh_set_intensity_range(hosts[r], vmin, vmax);                      // InitializeEMGPU RG.cc:2937-2951
    svrh_set_intensity_matching(hosts[r], no_matching ? 0 : 1);         // main.cc:1018, 1062
    if (nr > 1 && svrh_set_unit_order(hosts[r], order.data())) die("svrh_set_unit_order failed");
    if (!force_excluded.empty()) svrh_set_force_excluded(hosts[r], force_excluded.data(), (int)force_excluded.size());
    if (use_gpu_reg) HOSTR(r, svrh_prepare_registration_slices(hosts[r], grid.data() + o * mx * my, mx, my, sattr.data() + o, resolution));
  });
  if (nr > 1)
    fprintf(stderr, "%d ranks on devices%s, collectives: %s\n", nr, [&] { std::string t; for (int d : devices) t += " " + std::to_string(d); return t; }().c_str(),
            svr_group_uses_rccl(group) ? "RCCL" : "host memory (a device is named more than once: test mode)");
  svrh_recon *host = hosts[0];
  auto update_matrices_from_T = [&]() {                                  // UpdateGPUTranformationMatrices RG.cc:372-401
    for (int s = 0; s < ns; ++s) {
      M4 t;
      for (int q = 0; q < 16; ++q) t.m[q] = T[16 * (size_t)s + q];
      to_f16(t, &st[16 * (size_t)s]); to_f16(inverse_rigid_or_affine(t), &sti[16 * (size_t)s]);
    }
    for (int r = 0; r < nr; ++r) set_matrices(r);
  };

  if (!tfolder.empty()) {                                                // ReadTransformation, RG.cc:4733-4765
    for (int s = 0; s < ns; ++s) {
      double p6[6];
      char e[256] = {0};
      const std::string path = tfolder + "/transformation" + std::to_string(order[s]) + ".dof";       // (files are named in the reference's slice order)
      if (svr_dof_read(path.c_str(), p6, &T[16 * (size_t)s], e)) die(path + ": " + e);
    }
    update_matrices_from_T();
  }

  clk.mark("slices, engine set-up, upload");
  // ---- registration-reconstruction loop (main.cc:816-1237) ---------------------------------------------
  for (int it = 0; it < iterations; ++it) {
    bool slice_reg = it > 0 && !no_registration;
    if (slice_reg && !packages.empty() && it <= iterations * (levels - 1) / levels && it < iterations - 1) {
      // packages first (main.cc:832-864): plain, even/odd, even/odd halves; from iteration 4 on also the slices
      std::vector<svr_image_attr> at(n);
      std::vector<const double *> ptr(n);
      for (size_t k = 0; k < n; ++k) { at[k] = stacks[k].a; ptr[k] = stacks[k].d.data(); }
      std::vector<float> vol((size_t)tattr.nx * tattr.ny * tattr.nz);
      ENG(svr_sync_cpu(ctx, vol.data()));
      long evals = 0;
      char e[256] = {0};
      svr::unpermute_rows(T, 16, order);                                   // (packages are sub-stacks: one transformation per slice, stack after stack)
      if (svrh_package_to_volume(ctx, nullptr, (int)n, at.data(), ptr.data(), packages.data(), it >= 2, it >= 3, it >= 4 ? it - 2 : 1, T.data(),
                                 &tattr, vol.data(), &evals, e))
        die(std::string("package-to-volume registration: ") + e);
      svr::permute_rows(T, 16, order);
      fprintf(stderr, "package-to-volume registration: %ld similarity evaluations\n", evals);
      slice_reg = it >= 4;
      if (!slice_reg) update_matrices_from_T();
    }
    if (slice_reg) {                                                      // main.cc:829-880
      if (use_gpu_reg) {                                                  // every rank registers its own slices
        par([&](int r) { HOSTR(r, svrh_slice_to_volume_registration_gpu(hosts[r], T.data() + 16 * (size_t)rlo[r])); });
      } else {                                                            // SliceToVolumeRegistration, RG.cc:2291-2303
        std::vector<float> vol((size_t)tattr.nx * tattr.ny * tattr.nz);
        ENG(svr_sync_cpu(ctx, vol.data()));                               // _reconstructed after SyncCPU, main.cc:1189
        long evals = 0;
        char e[256] = {0};
        if (svrh_slice_to_volume_registration(ctx, nullptr, ns, grid.data(), mx, my, sattr.data(), T.data(), &tattr, vol.data(), 0, &evals, e))
          die(std::string("slice-to-volume registration: ") + e);
        fprintf(stderr, "slice-to-volume registration: %ld similarity evaluations\n", evals);
      }
      update_matrices_from_T();
    }
    for (int r = 0; r < nr; ++r) {
      if (it == iterations - 1) {                                         // main.cc:884-896
        svrh_set_smoothing_parameters(hosts[r], delta, last_lambda);
      } else {
        double l = lambda;
        for (int i = 0; i < levels; ++i) {
          if (it == iterations * (levels - i - 1) / levels) svrh_set_smoothing_parameters(hosts[r], delta, l);
          l *= 2;
        }
      }
    }
    if (slice_reg) clk.mark("registration");
    par([&](int r) { HOSTR(r, svrh_reconstruct_iteration(hosts[r], it == iterations - 1 ? rec_last : rec_first)); });   // main.cc:930-1140
    double sc[8];
    svrh_get_state(host, nullptr, nullptr, nullptr, nullptr, sc);
    fprintf(stderr, "iteration %d: sigma %.4g mix %.3f\n", it, sc[0], sc[1]);
    clk.mark("reconstruction iteration");
    if (save_slice_transformations) {
      // SaveSlices + SaveTransformations after every iteration (main.cc:1213-1217; RG.cc:4884-4892, 4903-4919), into the working
      // directory like the reference: slice<i>.nii.gz (the masked slice), croppedSliceTransformation<i>.dof (the slice's transformation)
      // and croppedSliceToVolumeTransformation<i>.dof (reconstructed W2I x transformation x slice I2W squeezed into a rigid
      // transformation by PutMatrix, as the reference does) -- <i> = the slice's index in the reference's order
      const M4 rw2i = world_to_image(tattr);
      for (int s = 0; s < ns; ++s) {
        const int i = order[s];
        const svr_image_attr &a = sattr[s];
        std::vector<float> img((size_t)a.nx * a.ny);
        for (int y = 0; y < a.ny; ++y) memcpy(&img[(size_t)y * a.nx], &grid[((size_t)s * my + y) * mx], a.nx * sizeof(float));
        char e[256] = {0};
        if (svr_nifti_write(("slice" + std::to_string(i) + ".nii.gz").c_str(), &a, img.data(), e)) die(std::string("slice file: ") + e);
        M4 t;
        for (int q = 0; q < 16; ++q) t.m[q] = T[16 * (size_t)s + q];
        double p6[6];
        svrh_irtk_rigid_parameters(t.m, p6, nullptr);
        if (svr_dof_write(("croppedSliceTransformation" + std::to_string(i) + ".dof").c_str(), p6, e)) die(std::string("dof file: ") + e);
        const M4 c = mul(rw2i, mul(t, image_to_world(a)));
        svrh_irtk_rigid_parameters(c.m, p6, nullptr);
        if (svr_dof_write(("croppedSliceToVolumeTransformation" + std::to_string(i) + ".dof").c_str(), p6, e)) die(std::string("dof file: ") + e);
      }
    }
  }
  par([&](int r) {                                                       // main.cc:1189-1193
    ENGR(r, svr_restore_slice_intensities(ctxs[r], factors.data(), (int)factors.size(), stack_index.data() + rlo[r]));
    HOSTR(r, svrh_scale_volume_gpu(hosts[r]));
  });
  std::vector<float> vol((size_t)tattr.nx * tattr.ny * tattr.nz);
  ENG(svr_sync_cpu(ctx, vol.data()));
  char err[256] = {0};
  clk.mark("restore, scale, download");
  if (svr_nifti_write(output.c_str(), &tattr, vol.data(), err)) die(output + ": " + err);
  clk.mark("write the volume");
  if (debug) {                                                           // SaveTransformations, RG.cc:4903-4915
    const size_t cut = output.find_last_of('/');
    const std::string folder = cut == std::string::npos ? "." : output.substr(0, cut);
    for (int s = 0; s < ns; ++s) {
      double p6[6];
      svrh_irtk_rigid_parameters(&T[16 * (size_t)s], p6, nullptr);
      char e[256] = {0};
      const std::string path = folder + "/transformation" + std::to_string(order[s]) + ".dof";
      if (svr_dof_write(path.c_str(), p6, e)) die(path + ": " + e);
    }
  }
  for (int r = 0; r < nr; ++r) svrh_destroy(hosts[r]);
  svr_group_destroy(group);
  for (int r = 0; r < nr; ++r) svr_destroy(ctxs[r]);
  clk.mark("teardown");
  return 0;
}
