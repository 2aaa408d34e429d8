// pvr_host.cpp -- the host side of the patch-to-volume reconstruction loop in C++:
// svr::irtkPatchBasedReconstruction mirrors the reconstruction part of irtkPatchBasedReconstruction<T>::run
// (source/reconstructionGPU2/irtkPatchBasedReconstruction.cpp = "PBR.cpp" :445-593) and the host halves of
// patchBasedRobustStatistics_gpu<T> ("PRS.cu" = patchBasedRobustStatistics_gpu.cu): initializeEMValues :78-95,
// EStep :224-556, MStep :570-640, Scale :672-745, InitializeRobustStatistics :793-845, with T = float.
// Device work goes through the engine's C-ABI (include/svr_hip.h) with the engine option "pvr" set.
//
// Kept quirks: the patch potentials of stack i are written at the patch index inside the stack, without the
// stack offset (PRS.cu:256-276); __step of G_ is 0.00001f (:97-101) while m_step is 0.0001; delta 1, lambda 0.1
// (patchBasedSuperresolution_gpu.cu:291-295).
#include <math.h>

#include <string>
#include <vector>

#include "../../include/svr_host.h"

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

  irtkPatchBasedReconstruction(svr_ctx *engine, const int *patches_per_stack, int n_stacks, float min_i, float max_i)
      : e(engine), counts(patches_per_stack, patches_per_stack + n_stacks), n(0), m_min_intensity(min_i),
        m_max_intensity(max_i), m_adaptive(false) {
    for (int c : counts) n += c;
    m_delta = 1.0f;
    m_lambda = 0.1f;
    m_alpha = (0.05f / m_lambda) * m_delta * m_delta;
    m_step = 0.0001f;
    m_sigma_gpu = m_mix_gpu = m_m_gpu = m_sigma_s_gpu = m_mix_s_gpu = m_mean_s_gpu = m_mean_s2_gpu = m_sigma_s2_gpu = 0;
    scale.assign(n, 1.0f);
    patch_weight.assign(n, 1.0f);
    patch_potential.assign(n, 0.0f);
  }

  int fail(int rc, const char *what) {
    err = std::string(what) + ": " + std::to_string(rc) + " " + svr_last_error(e);
    return rc;
  }
#define PENG(call) do { int rc_ = (call); if (rc_) return fail(rc_, #call); } while (0)

  static float G_(float x, float s) { return 0.00001f * expf(-x * x / (2.0f * s)) / sqrtf(6.28f * s); }   // PRS.cu:97-101

  int initializeEMValues() {                                                 // PRS.cu:78-95
    scale.assign(n, 1.0f);
    patch_weight.assign(n, 1.0f);
    PENG(svr_update_scale_vector(e, scale.data(), patch_weight.data()));
    PENG(svr_initialize_em_values(e));
    return 0;
  }

  int InitializeRobustStatistics() {                                         // PRS.cu:793-845
    double s2[2];
    PENG(svr_robust_statistics_sums(e, s2));
    if (s2[1] == 0) { err = "ERROR: sb = 0!! no sigma computed!"; return 10001; }   // the reference exits here
    m_sigma_gpu = (float)s2[0] / (float)s2[1];
    m_sigma_s_gpu = 0.025f;
    m_mix_gpu = 0.9f;
    m_mix_s_gpu = 0.9f;
    m_m_gpu = (float)(1.0f / (2.1f * m_max_intensity - 1.9f * m_min_intensity));
    return 0;
  }

  int EStep() {                                                              // PRS.cu:224-556
    std::vector<float> pot(n);
    PENG(svr_estep(e, m_m_gpu, m_sigma_gpu, m_mix_gpu, pot.data()));
    std::vector<float> pp(n, 0.0f);
    int ofs = 0;
    for (int c : counts) {                                                   // :256-276: no stack offset on the left
      for (int j = 0; j < c; ++j) pp[j] = pot[ofs + j];
      ofs += c;
    }
    std::vector<float> &pw = patch_weight;
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
    patch_potential = pp;
    PENG(svr_update_scale_vector(e, scale.data(), patch_weight.data()));     // copyToWeightsAndScales :486-491
    return 0;
  }

  int MStep(int iter) {                                                      // PRS.cu:570-640
    double s5[5];
    PENG(svr_mstep_sums(e, s5));
    const float sigma = (float)s5[0], mix = (float)s5[1], num = (float)s5[2], mn = (float)s5[3], mx = (float)s5[4];
    if (mix > 0) m_sigma_gpu = sigma / mix;
    if (m_sigma_gpu < m_step * m_step / 6.28f) m_sigma_gpu = m_step * m_step / 6.28f;
    if (iter > 1) m_mix_gpu = mix / num;
    m_m_gpu = 1.0f / (mx - mn);
    return 0;
  }

  int Scale() {                                                              // PRS.cu:672-745
    PENG(svr_calculate_scale_vector(e, scale.data()));
    PENG(svr_update_scale_vector(e, scale.data(), patch_weight.data()));     // copyToScales: no lag
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
    PENG(svr_gaussian_reconstruction(e, &nvox));     // reset + patchBasedPSFReconstruction_gpu + equalize
    std::vector<unsigned char> inside(n);
    PENG(svr_simulate_slices(e, inside.data()));
    if ((rc = InitializeRobustStatistics())) return rc;
    if ((rc = EStep())) return rc;
    for (int i = 0; i < rec_iterations; ++i) {
      if ((rc = Scale())) return rc;
      PENG(svr_superresolution(e, i + 1, patch_weight.data(), m_adaptive, m_alpha, m_min_intensity, m_max_intensity, m_delta,
                               m_lambda, 0, 12.0f, 0.01f));   // resetAddonCmap + run + regularize
      PENG(svr_simulate_slices(e, inside.data()));
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
  pvrh_recon(svr_ctx *e, const int *c, int ns, float mn, float mx) : impl(e, c, ns, mn, mx) {}
};

extern "C" {

pvrh_recon *pvrh_create(svr_ctx *engine, const int *patches_per_stack, int n_stacks, float min_intensity, float max_intensity) {
  if (!engine || !patches_per_stack || n_stacks <= 0) return nullptr;
  return new pvrh_recon(engine, patches_per_stack, n_stacks, min_intensity, max_intensity);
}
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
