// pvr_host.cpp -- the host side of the patch-to-volume reconstruction loop in C++:
// svr::irtkPatchBasedReconstruction mirrors the reconstruction part of irtkPatchBasedReconstruction<T>::run
// (source/reconstructionGPU2/irtkPatchBasedReconstruction.cpp = "PBR.cpp" :445-593) and the host halves of
// patchBasedRobustStatistics_gpu<T> ("PRS.cu" = patchBasedRobustStatistics_gpu.cu): initializeEMValues :78-95,
// EStep :224-556, MStep :570-640, Scale :672-745, InitializeRobustStatistics :793-845, with T = float.
// Device work goes through the engine's C-ABI (include/svr_hip.h) with the engine option "pvr" set.
//
// Sharded over ranks (svr_shard.h): the engine of a rank holds the patches [lo, hi) of the global numbering, `scale`,
// `patch_weight` and `patch_potential` stay GLOBAL vectors on every rank and the patch-level EM runs replicated on them.
//
// Kept quirks: the patch potentials of stack i are written at the patch index inside the stack, without the
// stack offset (PRS.cu:256-276); __step of G_ is 0.00001f (:97-101) while m_step is 0.0001; delta 1, lambda 0.1
// (patchBasedSuperresolution_gpu.cu:291-295).
#include <math.h>

#include <algorithm>

#include <string>
#include <vector>

#include "../../include/svr_host.h"
#include "svr_shard.h"

namespace svr {

class irtkPatchBasedReconstruction {
 public:
  svr_ctx *e;
  std::vector<int> counts;             // patches per stack
  int n;
  std::string err;
  float m_min_intensity, m_max_intensity;
  bool m_adaptive;
  float m_delta, m_lambda, m_alpha, m_step;
  float m_sigma_gpu, m_mix_gpu, m_m_gpu, m_sigma_s_gpu, m_mix_s_gpu, m_mean_s_gpu, m_mean_s2_gpu, m_sigma_s2_gpu;
  std::vector<float> scale, patch_weight, patch_potential;
  Shard sh;
  int lo, hi;
  // order[k] = the reference's index (stack after stack) of patch k of this object's numbering; empty = the same numbering
  // (pvrh_set_unit_order; csrc/svr_host.cpp has the why).  The patch-level EM -- sums in patch order, and the reference's within-stack
  // indexing of the potentials (PRS.cu:256-276), which only means something in its own numbering -- runs in the reference's order.
  std::vector<int> order;
  std::vector<float> to_ref(const std::vector<float> &v) const {
    if (order.empty()) return v;
    std::vector<float> r(v.size());
    for (size_t k = 0; k < v.size(); ++k) r[order[k]] = v[k];
    return r;
  }
  std::vector<float> from_ref(const std::vector<float> &r) const {
    if (order.empty()) return r;
    std::vector<float> v(r.size());
    for (size_t k = 0; k < r.size(); ++k) v[k] = r[order[k]];
    return v;
  }
  bool scale_stale = false;            // the other ranks' scales arrive with the next exchange (the E-step's)

  int fail(int rc, const char *what) {
    err = std::string(what) + ": " + std::to_string(rc) + " " + svr_last_error(e);
    return rc;
  }
#define PENG(call) do { int rc_ = (call); if (rc_) return fail(rc_, #call); } while (0)

  // ---- the patch-level EM on the device (round 5; csrc/svr_em.inc, patch form) -------------------------------------------------
  // As in csrc/svr_host.cpp: the host half of EStep below -- potentials down, the two-class EM over the patches, patch weights up -- was
  // the one wait of an SR iteration and, sharded, its one host exchange.  With SVR_DEVICE_SLICE_EM (default on; sharded: when the
  // launcher supplies allgather_device) every rank's potentials and scales meet on the device and the EM runs there as one workgroup;
  // `em_on_host` says whose copy of {scale, patch_weight, patch_potential, the eight scalars} is current: pull_state() brings the
  // device's over in one wait when somebody reads it, push_state() sends the host's when the host changed it.
  bool dev_patch_em = getenv("SVR_DEVICE_SLICE_EM") ? atoi(getenv("SVR_DEVICE_SLICE_EM")) != 0 : true;
  bool sem_ready = false, em_on_host = true;
  bool use_device_patch_em() const { return dev_patch_em && device_em && (!sh.on || sh.coll.allgather_device); }
  int push_state() {
    if (!sem_ready) {
      const int W = sh.on ? sh.coll.world : 1, R = sh.on ? sh.coll.rank : 0;
      std::vector<double> b((size_t)W + 1, 0.0);                       // every rank's range of this numbering: one small exchange, once
      b[R] = lo;
      if (R == W - 1) b[W] = hi;
      if (sh.on && W > 1) { if (int rc = sh.coll.allreduce_host(sh.coll.user, b.data(), W + 1, 0)) return fail(rc, "allreduce_host (patch ranges)"); }
      std::vector<int> rlo((size_t)W + 1);
      for (int r = 0; r <= W; ++r) rlo[r] = (int)b[r];
      PENG(svr_slice_em_setup(e, n, W, R, rlo.data(), order.empty() ? nullptr : order.data(), (double)m_step));
      std::vector<int> src(n, -1);                                     // PRS.cu:256-276: no stack offset on the left
      int ofs = 0;
      for (int c : counts) {
        for (int j = 0; j < c; ++j) src[j] = ofs + j;
        ofs += c;
      }
      PENG(svr_slice_em_set_patch_form(e, src.data()));
      sem_ready = true;
      em_on_host = true;
    }
    if (em_on_host) {
      const std::vector<unsigned char> excl(n, 0);
      const double s5[5] = {m_mean_s_gpu, m_mean_s2_gpu, m_sigma_s_gpu, m_sigma_s2_gpu, m_mix_s_gpu};
      const float em3[3] = {m_sigma_gpu, m_mix_gpu, m_m_gpu};
      PENG(svr_slice_em_set_state(e, patch_weight.data(), excl.data(), s5, em3));
    }
    return 0;
  }
  int pull_state() {
    if (em_on_host) return 0;
    double s5[5];
    float em3[3];
    PENG(svr_slice_em_fetch(e, scale.data(), patch_weight.data(), patch_potential.data(), nullptr, s5, em3));
    m_mean_s_gpu = (float)s5[0]; m_mean_s2_gpu = (float)s5[1]; m_sigma_s_gpu = (float)s5[2]; m_sigma_s2_gpu = (float)s5[3]; m_mix_s_gpu = (float)s5[4];
    m_sigma_gpu = em3[0]; m_mix_gpu = em3[1]; m_m_gpu = em3[2];
    em_on_host = true;
    return 0;
  }

  irtkPatchBasedReconstruction(svr_ctx *engine, const int *patches_per_stack, int n_stacks, float min_i, float max_i, int lo_ = 0,
                               int hi_ = -1, const svr_collectives *c = nullptr)
      : e(engine), counts(patches_per_stack, patches_per_stack + n_stacks), n(0), m_min_intensity(min_i),
        m_max_intensity(max_i), m_adaptive(false) {
    for (int c_ : counts) n += c_;
    lo = lo_; hi = hi_ < 0 ? n : hi_;
    sh.init(engine, n, lo, hi, c);
    m_delta = 1.0f;
    m_lambda = 0.1f;
    m_alpha = (0.05f / m_lambda) * m_delta * m_delta;
    m_step = 0.0001f;
    m_sigma_gpu = m_mix_gpu = m_m_gpu = m_sigma_s_gpu = m_mix_s_gpu = m_mean_s_gpu = m_mean_s2_gpu = m_sigma_s2_gpu = 0;
    scale.assign(n, 1.0f);
    patch_weight.assign(n, 1.0f);
    patch_potential.assign(n, 0.0f);
  }


  static float G_(float x, float s) { return 0.00001f * expf(-x * x / (2.0f * s)) / sqrtf(6.28f * s); }   // PRS.cu:97-101

  int initializeEMValues() {                                                 // PRS.cu:78-95
    if (int rc = settle()) return rc;
    scale.assign(n, 1.0f);
    patch_weight.assign(n, 1.0f);
    PENG(svr_update_scale_vector(e, scale.data() + lo, patch_weight.data() + lo));
    PENG(svr_initialize_em_values(e));
    return 0;
  }

  int exchange(const double *mine, int n_mine, std::vector<double> &all, std::vector<float> *pot) {
    if (int rc = settle()) return rc;                  // this rank's own part of the scale vector
    std::vector<float> *vec[3] = {scale_stale ? &scale : nullptr, nullptr, pot};
    const int rc = sh.exchange(mine, n_mine, all, vec);
    if (rc) { err = rc == SVR_E_STATE ? "exchange: the ranks are not in the same step of the reconstruction" : "exchange: the collective failed"; return rc; }
    scale_stale = false;
    return 0;
  }
  // One rank: the scale vector and the M-step's scalars stay on the device until the E-step fetches them with its
  // potentials in one wait (svr_mstep_estep); `settle` brings them over for anything else that reads them.
  bool scale_pending = false;
  int mstep_pending = 0;
  int settle() {
    if (int rc = pull_state()) return rc;              // (the device's patch-level state, if it is the current one)
    if (mstep_pending) {
      const int iter = mstep_pending;
      mstep_pending = 0;
      if (int rc = MStepNow(iter)) return rc;
    }
    if (scale_pending) {
      PENG(svr_get_scale_vector(e, scale.data() + lo));
      scale_pending = false;
    }
    return 0;
  }
  int flush() {
    if (int rc = settle()) return rc;
    if (!sh.on || !scale_stale) return 0;
    std::vector<double> none;
    return exchange(nullptr, 0, none, nullptr);
  }

  int InitializeRobustStatistics() {                                         // PRS.cu:793-845
    if (int rc = settle()) return rc;
    double s2[2];
    PENG(svr_robust_statistics_sums(e, s2));
    if (sh.on) {
      std::vector<double> all;
      if (int rc = exchange(s2, 2, all, nullptr)) return rc;
      s2[0] = s2[1] = 0;
      for (int r = 0; r < sh.coll.world; ++r) { s2[0] += all[2 * r]; s2[1] += all[2 * r + 1]; }   // rank order: the same bits everywhere
    }
    if (s2[1] == 0) { err = "ERROR: sb = 0!! no sigma computed!"; return 10001; }   // the reference exits here
    m_sigma_gpu = (float)s2[0] / (float)s2[1];
    m_sigma_s_gpu = 0.025f;
    m_mix_gpu = 0.9f;
    m_mix_s_gpu = 0.9f;
    m_m_gpu = (float)(1.0f / (2.1f * m_max_intensity - 1.9f * m_min_intensity));
    return 0;
  }

  int EStep() {                                                              // PRS.cu:224-556
    if (use_device_patch_em()) {
      // [M-step] + E-step + the patch-level EM without a wait and without a host exchange (csrc/svr_em.inc; svr_host.cpp EStepGPU)
      if (int rc = push_state()) return rc;
      const int iter = mstep_pending;
      mstep_pending = 0;
      void *send = nullptr, *recv = nullptr;
      if (iter > 0 && sh.on) {
        PENG(svr_mstep_partial(e, sh.coll.world, &send, &recv));
        if (int rc = sh.before_device_collective()) return fail(rc, "svr_stream_sync");
        if (int rc = sh.coll.allgather_device(sh.coll.user, send, recv, 16)) return fail(rc, "allgather_device (M-step sums)");
      }
      size_t nf = 0;
      PENG(svr_mstep_estep_device(e, iter, m_step, &send, &recv, &nf));
      if (sh.on) {
        if (int rc = sh.before_device_collective()) return fail(rc, "svr_stream_sync");
        if (int rc = sh.coll.allgather_device(sh.coll.user, send, recv, nf)) return fail(rc, "allgather_device (patch potentials)");
      }
      PENG(svr_slice_em_run(e));
      em_on_host = false;
      scale_pending = scale_stale = false;               // (the scales travelled with the gather)
      return 0;
    }
    std::vector<float> pot(n, 0.0f);
    if (mstep_pending && sh.on) {
      // sharded (round 4; svr_host.cpp EStepGPU): the ranks' M-step sums meet on the device, one wait and one host exchange per SR iteration
      const int iter = mstep_pending;
      mstep_pending = 0;
      void *send = nullptr, *recv = nullptr;
      PENG(svr_mstep_partial(e, sh.coll.world, &send, &recv));
      if (int rc = sh.before_device_collective()) return fail(rc, "svr_stream_sync");
      if (int rc = sh.coll.allgather_device(sh.coll.user, send, recv, 16)) return fail(rc, "allgather_device (M-step sums)");
      float em3[3] = {m_sigma_gpu, m_mix_gpu, m_m_gpu};
      PENG(svr_mstep_estep_ranks(e, sh.coll.world, iter, m_step, em3, pot.data() + lo, scale_pending ? scale.data() + lo : nullptr, nullptr));
      m_sigma_gpu = em3[0]; m_mix_gpu = em3[1]; m_m_gpu = em3[2];
      scale_pending = false;
    } else if (mstep_pending) {                      // one rank: M-step + E-step + the scale vector, one wait for the device
      const int iter = mstep_pending;
      mstep_pending = 0;
      float em3[3] = {m_sigma_gpu, m_mix_gpu, m_m_gpu};
      PENG(svr_mstep_estep(e, iter, m_step, em3, pot.data() + lo, scale_pending ? scale.data() + lo : nullptr, nullptr));
      m_sigma_gpu = em3[0]; m_mix_gpu = em3[1]; m_m_gpu = em3[2];
      scale_pending = false;
    } else {
      if (int rc = settle()) return rc;
      PENG(svr_estep(e, m_m_gpu, m_sigma_gpu, m_mix_gpu, pot.data() + lo));
    }
    if (sh.on) { std::vector<double> none; if (int rc = exchange(nullptr, 0, none, &pot)) return rc; }   // (and the scale vector)
    std::vector<float> pp(n, 0.0f);
    int ofs = 0;
    pot = to_ref(pot);                                                       // from here on in the reference's patch order
    const std::vector<float> scale = to_ref(this->scale);
    std::vector<float> pw = to_ref(patch_weight);
    for (int c : counts) {                                                   // :256-276: no stack offset on the left
      for (int j = 0; j < c; ++j) pp[j] = pot[ofs + j];
      ofs += c;
    }
    for (int i = 0; i < n; ++i)
      if (scale[i] < 0.2 || scale[i] > 5) pp[i] = -1;                       // :307-311
    double sum = 0, den = 0, sum2 = 0, den2 = 0, maxs = 0, mins = 1;
    for (int i = 0; i < n; ++i)
      if (pp[i] >= 0) {
        sum += pp[i] * pw[i];
        den += pw[i];
        sum2 += pp[i] * (1.0 - pw[i]);
        den2 += (1.0 - pw[i]);
        if (pp[i] > maxs) maxs = pp[i];
        if (pp[i] < mins) mins = pp[i];
      }
    m_mean_s_gpu = den > 0 ? (float)(sum / den) : (float)mins;
    m_mean_s2_gpu = den2 > 0 ? (float)(sum2 / den2) : (float)((maxs + m_mean_s_gpu) / 2.0);
    sum = den = sum2 = den2 = 0;
    for (int i = 0; i < n; ++i)
      if (pp[i] >= 0) {
        sum += (pp[i] - m_mean_s_gpu) * (pp[i] - m_mean_s_gpu) * pw[i];
        den += pw[i];
        sum2 += (pp[i] - m_mean_s2_gpu) * (pp[i] - m_mean_s2_gpu) * (1 - pw[i]);
        den2 += (1 - pw[i]);
      }
    if (sum > 0 && den > 0) {
      m_sigma_s_gpu = (float)(sum / den);
      if (m_sigma_s_gpu < m_step * m_step / 6.28) m_sigma_s_gpu = (float)(m_step * m_step / 6.28);
    } else {
      m_sigma_s_gpu = 0.025f;
    }
    if (sum2 > 0 && den2 > 0) {
      m_sigma_s2_gpu = (float)(sum2 / den2);
      if (m_sigma_s2_gpu < m_step * m_step / 6.28) m_sigma_s2_gpu = (float)(m_step * m_step / 6.28);
    } else {
      m_sigma_s2_gpu = (m_mean_s2_gpu - m_mean_s_gpu) * (m_mean_s2_gpu - m_mean_s_gpu) / 4;
      if (m_sigma_s2_gpu < m_step * m_step / 6.28) m_sigma_s2_gpu = (float)(m_step * m_step / 6.28);
    }
    double gs1, gs2;
    for (int i = 0; i < n; ++i) {                                            // :415-452
      if (pp[i] == -1) { pw[i] = 0; continue; }
      if (den <= 0 || m_mean_s2_gpu <= m_mean_s_gpu) { pw[i] = 1; continue; }
      gs1 = pp[i] < m_mean_s2_gpu ? G_(pp[i] - m_mean_s_gpu, m_sigma_s_gpu) : 0;
      gs2 = pp[i] > m_mean_s_gpu ? G_(pp[i] - m_mean_s2_gpu, m_sigma_s2_gpu) : 0;
      const double likelihood = gs1 * m_mix_s_gpu + gs2 * (1 - m_mix_s_gpu);
      if (likelihood > 0) {
        pw[i] = (float)(gs1 * m_mix_s_gpu / likelihood);
      } else {
        if (pp[i] <= m_mean_s_gpu) pw[i] = 1;
        if (pp[i] >= m_mean_s2_gpu) pw[i] = 0;
        if (pp[i] < m_mean_s2_gpu && pp[i] > m_mean_s_gpu) pw[i] = 1;
      }
    }
    sum = 0;
    int num = 0;
    for (int i = 0; i < n; ++i)
      if (pp[i] >= 0) { sum += pw[i]; num++; }
    m_mix_s_gpu = num > 0 ? (float)(sum / num) : 0.9f;                       // :455-468
    patch_potential = from_ref(pp);
    patch_weight = from_ref(pw);
    PENG(svr_update_scale_vector(e, this->scale.data() + lo, patch_weight.data() + lo));     // copyToWeightsAndScales :486-491
    return 0;
  }

  int MStep(int iter) {                                                      // PRS.cu:570-640
    if (iter > 0 && (!sh.on || (device_em && sh.coll.allgather_device))) {
      if (mstep_pending) { if (int rc = settle()) return rc; }   // (only an M-step still waiting; the scale vector stays pending for the fused fetch)
      mstep_pending = iter;                          // runs with the E-step that follows (PBR.cpp:540-545), or in settle
      return 0;
    }
    return MStepNow(iter);
  }
  bool device_em = getenv("SVR_DEVICE_EM") ? atoi(getenv("SVR_DEVICE_EM")) != 0 : true;   // sharded: the M-step's sums meet on the device
  int MStepNow(int iter) {
    if (int rc = pull_state()) return rc;
    double s5[5];
    PENG(svr_mstep_sums_fetch(e, s5, scale_pending ? scale.data() + lo : nullptr, nullptr));
    scale_pending = false;
    if (sh.on) {
      std::vector<double> all;
      if (int rc = exchange(s5, 5, all, nullptr)) return rc;                 // three sums, a minimum, a maximum: one collective
      s5[0] = s5[1] = s5[2] = 0;
      for (int r = 0; r < sh.coll.world; ++r) {
        for (int k = 0; k < 3; ++k) s5[k] += all[5 * r + k];
        s5[3] = r ? std::min(s5[3], all[5 * r + 3]) : all[3];
        s5[4] = r ? std::max(s5[4], all[5 * r + 4]) : all[4];
      }
    }
    const float sigma = (float)s5[0], mix = (float)s5[1], num = (float)s5[2], mn = (float)s5[3], mx = (float)s5[4];
    if (mix > 0) m_sigma_gpu = sigma / mix;
    if (m_sigma_gpu < m_step * m_step / 6.28f) m_sigma_gpu = m_step * m_step / 6.28f;
    if (iter > 1) m_mix_gpu = mix / num;
    m_m_gpu = 1.0f / (mx - mn);
    return 0;
  }

  int Scale() {                                                              // PRS.cu:672-745
    PENG(svr_calculate_scale_vector(e, nullptr));                            // stays on the device: fetched with the E-step's potentials
    PENG(svr_adopt_scale_vector(e));                                         // (sharded: with the M-step's sums), or by settle.  copyToScales: no lag
    scale_pending = true;
    scale_stale = sh.on;                                                     // read next in the E-step, whose exchange completes it
    return 0;
  }

  // patchBased2D3DRegistration<T>::run for all patches (PBR.cpp:452-489): registers them against the current reconstruction and
  // hands the new transformations back to the engine.  T / Tinv [n][16] in/out, the other matrices as uploaded.
  int registerPatches(const float *ri2w, const float *mo, const float *invmo, float *T, float *Tinv, const float *i2w, const float *w2i,
                      const float *recon_i2w, const float *recon_w2i, long long counters3[3]) {
    PENG(svr_pvr_register_patches(e, ri2w, mo, invmo, T, Tinv, counters3));
    PENG(svr_set_slice_matrices(e, T, Tinv, i2w, w2i, i2w, w2i, recon_i2w, recon_w2i));
    return 0;
  }

  // one outer iteration without the patch registration (PBR.cpp:490-548)
  int reconstruct_iteration(int rec_iterations) {
    int rc;
    if ((rc = initializeEMValues())) return rc;
    int nvox = 0;
    if (!sh.on) {
      PENG(svr_gaussian_reconstruction(e, &nvox));     // reset + patchBasedPSFReconstruction_gpu + equalize
    } else {
      PENG(svr_gaussian_reconstruction_local(e));
      PENG(sh.allreduce_pair(SVR_BUF_RECONSTRUCTED, 2 * svr_volume_voxels(e)));
      PENG(svr_gaussian_reconstruction_finish(e, &nvox));
    }
    PENG(svr_simulate_slices(e, nullptr));          // (the patch-based loop never reads the inside flags: no wait)
    if ((rc = InitializeRobustStatistics())) return rc;
    if ((rc = EStep())) return rc;
    for (int i = 0; i < rec_iterations; ++i) {
      if ((rc = sr_iteration(i))) return rc;
    }
    return 0;
  }

  // one SR iteration (PBR.cpp:505-546): Scale, resetAddonCmap + run + regularize, simulate, M-step, E-step
  int sr_iteration(int i) {
    int rc;
    {
      if ((rc = Scale())) return rc;
      // (the patch weights of a device-side EM are already on the device: NULL = keep them; whatever anybody sent the engine in between
      // is replaced by the EM's, device to device)
      const float *pw = em_on_host ? patch_weight.data() + lo : nullptr;
      if (!pw) PENG(svr_slice_em_apply_weights(e));
      if (!sh.on) {
        PENG(svr_superresolution(e, i + 1, pw, m_adaptive, m_alpha, m_min_intensity, m_max_intensity, m_delta,
                                 m_lambda, 0, 12.0f, 0.01f));
      } else {
        PENG(sh.superresolution(pw, m_adaptive, m_alpha, m_min_intensity, m_max_intensity, m_delta, m_lambda));
      }
      PENG(svr_simulate_slices(e, nullptr));
      if ((rc = MStep(i + 1))) return rc;
      if ((rc = EStep())) return rc;
    }
    return 0;
  }
#undef PENG
};

}  // namespace svr

struct pvrh_recon {
  svr::irtkPatchBasedReconstruction impl;
  pvrh_recon(svr_ctx *e, const int *c, int ns, float mn, float mx, int lo, int hi, const svr_collectives *coll) : impl(e, c, ns, mn, mx, lo, hi, coll) {}
};

extern "C" {

pvrh_recon *pvrh_create(svr_ctx *engine, const int *patches_per_stack, int n_stacks, float min_intensity, float max_intensity) {
  if (!engine || !patches_per_stack || n_stacks <= 0) return nullptr;
  return new pvrh_recon(engine, patches_per_stack, n_stacks, min_intensity, max_intensity, 0, -1, nullptr);
}
pvrh_recon *pvrh_create_sharded(svr_ctx *engine, const int *patches_per_stack, int n_stacks, float min_intensity, float max_intensity,
                                int patch_lo, int patch_hi, const svr_collectives *coll) {
  if (!engine || !patches_per_stack || n_stacks <= 0) return nullptr;
  long total = 0;
  for (int k = 0; k < n_stacks; ++k) total += patches_per_stack[k];
  if (patch_lo < 0 || patch_hi < patch_lo || patch_hi > total) return nullptr;
  if (coll && coll->struct_size < SVR_COLLECTIVES_MIN_SIZE) return nullptr;
  if (coll && coll->world > 1 && (!coll->allreduce_volume_pair || !coll->allreduce_host)) return nullptr;
  return new pvrh_recon(engine, patches_per_stack, n_stacks, min_intensity, max_intensity, patch_lo, patch_hi, coll);
}
int pvrh_set_unit_order(pvrh_recon *r, const int *order_or_null) {
  if (!r) return SVR_E_ARG;
  svr::irtkPatchBasedReconstruction &m = r->impl;
  if (!order_or_null) { if (int rc = m.settle()) return rc; m.order.clear(); m.sem_ready = false; return SVR_OK; }
  std::vector<char> seen(m.n, 0);
  for (int k = 0; k < m.n; ++k) {
    const int i = order_or_null[k];
    if (i < 0 || i >= m.n || seen[i]) { m.err = "pvrh_set_unit_order: not a permutation of the patches"; return SVR_E_ARG; }
    seen[i] = 1;
  }
  if (int rc = m.settle()) return rc;
  m.order.assign(order_or_null, order_or_null + m.n);
  m.sem_ready = false;                                 // the device-side EM learns the new numbering at its next use
  return SVR_OK;
}
void pvrh_force_collectives(pvrh_recon *r, int on) { if (r) { (void)r->impl.settle(); r->impl.sh.force(on != 0); } }
void pvrh_set_slab_update(pvrh_recon *r, int on) { if (r) r->impl.sh.slabs = on != 0; }
int pvrh_sr_iteration(pvrh_recon *r, int i) { return r->impl.sr_iteration(i); }
void pvrh_destroy(pvrh_recon *r) { delete r; }
const char *pvrh_last_error(const pvrh_recon *r) { return r ? r->impl.err.c_str() : "null"; }
int pvrh_initialize_em_values(pvrh_recon *r) { return r->impl.initializeEMValues(); }
int pvrh_initialize_robust_statistics(pvrh_recon *r) { return r->impl.InitializeRobustStatistics(); }
int pvrh_estep(pvrh_recon *r) { return r->impl.EStep(); }
int pvrh_mstep(pvrh_recon *r, int iter) { return r->impl.MStep(iter); }
int pvrh_scale(pvrh_recon *r) { return r->impl.Scale(); }
int pvrh_reconstruct_iteration(pvrh_recon *r, int rec_iterations) { return r->impl.reconstruct_iteration(rec_iterations); }
int pvrh_register_patches(pvrh_recon *r, const float *ri2w, const float *mo, const float *invmo, float *T, float *Tinv, const float *i2w,
                          const float *w2i, const float *recon_i2w, const float *recon_w2i, long long counters3[3]) {
  return r->impl.registerPatches(ri2w, mo, invmo, T, Tinv, i2w, w2i, recon_i2w, recon_w2i, counters3);
}
int pvrh_get_state(pvrh_recon *r, float *scale, float *patch_weight, float *patch_potential, double scalars8[8]) {
  svr::irtkPatchBasedReconstruction &p = r->impl;
  if (int rc = p.flush()) return rc;       // sharded: collective (the other ranks' scales may still be on their way)
  if (scale) std::copy(p.scale.begin(), p.scale.end(), scale);
  if (patch_weight) std::copy(p.patch_weight.begin(), p.patch_weight.end(), patch_weight);
  if (patch_potential) std::copy(p.patch_potential.begin(), p.patch_potential.end(), patch_potential);
  if (scalars8) {
    const double s[8] = {p.m_sigma_gpu, p.m_mix_gpu, p.m_m_gpu, p.m_mean_s_gpu, p.m_mean_s2_gpu, p.m_sigma_s_gpu,
                         p.m_sigma_s2_gpu, p.m_mix_s_gpu};
    for (int k = 0; k < 8; ++k) scalars8[k] = s[k];
  }
  return 0;
}

}  // extern "C"
