// svr_host.cpp -- svr::irtkReconstruction: the reference's host algorithm object (GPU operator
// surface of irtkReconstruction, irtkReconstructionGPU.cc = "RG.cc") in C++ above the C-ABI engine.
// Plain host C++: no HIP calls here, everything device-side goes through include/svr_hip.h.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/svr_host.h"
#include "svr_shard.h"

namespace svr {

class irtkReconstruction {
 public:
  // engine + sharding
  svr_ctx *reconstructionGPU;          // RG.h: Reconstruction* reconstructionGPU
  int ns, lo, hi;
  Shard sh;                            // this rank's slice range, the collectives, the one exchange per step (svr_shard.h)
  std::string err;

  // members named as in RG.h / RG.cc:159-221
  double _step;
  int _quality_factor;
  float _sigma_bias;
  float _sigma_gpu, _mix_gpu, _m_gpu;
  float _mean_s_gpu, _mean_s2_gpu, _sigma_s_gpu, _sigma_s2_gpu, _mix_s_gpu;
  double _delta, _lambda, _alpha;
  float _low_intensity_cutoff;
  bool _global_bias_correction, _adaptive, _disableBiasC;
  bool _intensity_matching;            // reconstruction.cc:114,183: false skips Bias / Scale / NormaliseBias in every SR iteration
  double _max_intensity, _min_intensity;
  std::vector<int> _force_excluded, _small_slices;
  std::vector<float> _scale_gpu, _slice_weight_gpu, _slice_potential_gpu;
  std::vector<unsigned char> _slice_inside_gpu;

  irtkReconstruction(svr_ctx *engine, int n_global, int lo_, int hi_, const svr_collectives *c)
      : reconstructionGPU(engine), ns(n_global), lo(lo_), hi(hi_) {
    sh.init(engine, n_global, lo_, hi_, c);
    _step = 0.0001;
    _quality_factor = 2;
    _sigma_bias = 12;
    _sigma_s_gpu = 0.025f;
    _sigma_s2_gpu = 0.025f;
    _mix_s_gpu = 0.9f;
    _mix_gpu = 0.9f;
    _delta = 1;
    _lambda = 0.1f;
    _alpha = (0.05f / _lambda) * _delta * _delta;
    _low_intensity_cutoff = 0.01f;
    _global_bias_correction = false;
    _adaptive = false;
    _disableBiasC = true;   // reconstruction.cc:121,202
    _intensity_matching = true;
    _max_intensity = 1; _min_intensity = 0;
    _sigma_gpu = 0; _m_gpu = 0; _mean_s_gpu = 0; _mean_s2_gpu = 0;
    _scale_gpu.assign(ns, 1.0f);
    _slice_weight_gpu.assign(ns, 1.0f);
    _slice_potential_gpu.assign(ns, 0.0f);
    _slice_inside_gpu.assign(ns, 1);
  }

  int fail(int rc, const char *what) {
    err = std::string(what) + ": " + (rc >= 10000 || rc < 0 ? "" : "hip error ") + std::to_string(rc) + " " +
          svr_last_error(reconstructionGPU);
    return rc;
  }
#define ENG(call) do { int rc_ = (call); if (rc_) return fail(rc_, #call); } while (0)

  // -- sharding helpers ------------------------------------------------------------------
  const float *local(const std::vector<float> &v) const { return v.data() + lo; }
  // The numbering of a sharded run need not be the reference's (svrh_set_unit_order): a launcher that deals the r-th part of EVERY
  // stack to rank r (spatially compact shards, sharding.shard_units / csrc/svr_shard.h spatial_order) uploads the slices rank after
  // rank.  order[k] = the reference's index of slice k of this object's numbering; empty = the same numbering.  Everything per slice is
  // indifferent to the numbering; what the reference does ACROSS slices in slice order -- the sums of the slice-level EM -- is done in
  // the reference's order (to_ref / from_ref), so a permuted run adds the same numbers in the same order as an unpermuted one.
  std::vector<int> order;
  template <class T> std::vector<T> to_ref(const std::vector<T> &v) const {
    if (order.empty()) return v;
    std::vector<T> r(v.size());
    for (size_t k = 0; k < v.size(); ++k) r[order[k]] = v[k];
    return r;
  }
  template <class T> std::vector<T> from_ref(const std::vector<T> &r) const {
    if (order.empty()) return r;
    std::vector<T> v(r.size());
    for (size_t k = 0; k < r.size(); ++k) v[k] = r[order[k]];
    return v;
  }
  int ref_index(int k) const { return order.empty() ? k : order[k]; }

  // ---- the slice-level EM on the device (round 5; csrc/svr_em.inc) --------------------------------------------------------------
  // The host half of EStepGPU below -- potentials down, a two-class EM over the slices, slice weights up -- was the one wait of an SR
  // iteration and, sharded, its one host exchange.  With SVR_DEVICE_SLICE_EM (default on; sharded: when the launcher supplies
  // allgather_device) the E-step's potentials, the scale vector and slice_inside of every rank meet on the device (one all-gather of
  // 3 x maxn floats) and the EM runs there as one workgroup: an SR iteration only queues launches.  `_em_on_host` says whose copy of
  // the slice-level state (_scale_gpu, _slice_weight_gpu, _slice_potential_gpu, _slice_inside_gpu, the eight scalars) is current;
  // pull_state() brings the device's over in one wait when somebody reads it, push_state() sends the host's when it changed it.
  bool dev_slice_em = getenv("SVR_DEVICE_SLICE_EM") ? atoi(getenv("SVR_DEVICE_SLICE_EM")) != 0 : true;
  bool _sem_ready = false, _em_on_host = true;
  bool use_device_slice_em() const { return dev_slice_em && device_em && (!sh.on || sh.coll.allgather_device); }
  int push_state() {
    if (!_sem_ready) {
      const int W = sh.on ? sh.coll.world : 1, R = sh.on ? sh.coll.rank : 0;
      std::vector<double> b((size_t)W + 1, 0.0);                       // every rank's range of this numbering: one small exchange, once
      b[R] = lo;
      if (R == W - 1) b[W] = hi;
      if (sh.on && W > 1) ENG(sh.coll.allreduce_host(sh.coll.user, b.data(), W + 1, 0));
      std::vector<int> rlo((size_t)W + 1);
      for (int r = 0; r <= W; ++r) rlo[r] = (int)b[r];
      ENG(svr_slice_em_setup(reconstructionGPU, ns, W, R, rlo.data(), order.empty() ? nullptr : order.data(), _step));
      _sem_ready = true;
      _em_on_host = true;
    }
    if (_em_on_host) {
      std::vector<unsigned char> excl(ns, 0);
      for (int i : _force_excluded) if (i >= 0 && i < ns) excl[i] = 1;
      for (int i : _small_slices) if (i >= 0 && i < ns) excl[i] = 1;
      const double s5[5] = {_mean_s_gpu, _mean_s2_gpu, _sigma_s_gpu, _sigma_s2_gpu, _mix_s_gpu};
      const float em3[3] = {_sigma_gpu, _mix_gpu, _m_gpu};
      ENG(svr_slice_em_set_state(reconstructionGPU, _slice_weight_gpu.data(), excl.data(), s5, em3));
    }
    return 0;
  }
  int pull_state() {
    if (_em_on_host) return 0;
    double s5[5];
    float em3[3];
    ENG(svr_slice_em_fetch(reconstructionGPU, _scale_gpu.data(), _slice_weight_gpu.data(), _slice_potential_gpu.data(), _slice_inside_gpu.data(), s5, em3));
    _mean_s_gpu = (float)s5[0]; _mean_s2_gpu = (float)s5[1]; _sigma_s_gpu = (float)s5[2]; _sigma_s2_gpu = (float)s5[3]; _mix_s_gpu = (float)s5[4];
    _sigma_gpu = em3[0]; _mix_gpu = em3[1]; _m_gpu = em3[2];
    _em_on_host = true;
    return 0;
  }
  // The ns-sized vectors a rank has only its own part of (`_scale_stale`, `_inside_stale`) ride along with the next exchange
  // that every rank makes anyway (Shard::exchange: one collective): the M-step's sums, the E-step's potentials, the
  // robust-statistics sums.
  //   mine[n_mine] -> all[world][n_mine];  pot (or NULL): the slice potentials, this rank's range filled -> complete
  bool _scale_stale = false, _inside_stale = false;
  int exchange(const double *mine, int n_mine, std::vector<double> &all, std::vector<float> *pot) {
    if (int rc = settle()) return rc;                  // this rank's own parts of the vectors that travel
    std::vector<float> inside;
    if (_inside_stale) inside.assign(_slice_inside_gpu.begin(), _slice_inside_gpu.end());
    std::vector<float> *vec[3] = {_scale_stale ? &_scale_gpu : nullptr, _inside_stale ? &inside : nullptr, pot};
    const int rc = sh.exchange(mine, n_mine, all, vec);
    if (rc) { err = rc == SVR_E_STATE ? "exchange: the ranks are not in the same step of the reconstruction" : "exchange: the collective failed"; return rc; }
    if (_inside_stale) for (int i = 0; i < ns; ++i) _slice_inside_gpu[i] = inside[i] > 0.5f;
    _scale_stale = _inside_stale = false;
    return 0;
  }
  // One rank, nothing to exchange: the scale vector, slice_inside and the M-step's scalars stay on the device until the
  // E-step fetches them with its potentials in one wait (svr_mstep_estep) -- one wait per SR iteration instead of four.
  // `settle` brings over whatever is still there when something else wants to read it.
  bool _scale_pending = false, _inside_pending = false;
  int _mstep_pending = 0;                            // iteration number of an M-step not yet run, or 0
  int settle() {
    if (int rc = pull_state()) return rc;              // (the device's slice-level state, if it is the current one)
    if (_mstep_pending) {
      const int iter = _mstep_pending;
      _mstep_pending = 0;
      if (sh.on) { if (int rc = mstep_exchange(iter)) return rc; }       // (collective: the ranks run the same operator sequence)
      else ENG(svr_mstep(reconstructionGPU, iter, (float)_step, &_sigma_gpu, &_mix_gpu, &_m_gpu));
    }
    if (_scale_pending) {
      std::vector<float> loc(hi - lo);
      ENG(svr_get_scale_vector(reconstructionGPU, loc.data()));
      std::copy(loc.begin(), loc.end(), _scale_gpu.begin() + lo);
      _scale_pending = false;
    }
    if (_inside_pending) {
      std::vector<unsigned char> inside(hi - lo);
      ENG(svr_get_slice_inside(reconstructionGPU, inside.data()));
      for (int i = 0; i < hi - lo; ++i) _slice_inside_gpu[lo + i] = inside[i] != 0;
      _inside_pending = false;
    }
    return 0;
  }
  // completes the vectors of which a rank only holds its own part (collective: every rank calls it); svrh_get_state does
  int flush() {
    if (int rc = settle()) return rc;
    if (!sh.on || (!_scale_stale && !_inside_stale)) return 0;
    std::vector<double> none;
    return exchange(nullptr, 0, none, nullptr);
  }

  // RG.h:605-612
  void SetSmoothingParameters(double delta, double lambda) {
    _delta = delta;
    _lambda = lambda * delta * delta;
    _alpha = 0.05 / lambda;
    if (_alpha > 1) _alpha = 1;
  }

  // RG.cc:2905-2919
  int InitializeEMValuesGPU() {
    if (int rc = settle()) return rc;
    _slice_weight_gpu.assign(ns, 1);
    _scale_gpu.assign(ns, 1);
    ENG(svr_update_scale_vector(reconstructionGPU, local(_scale_gpu), local(_slice_weight_gpu)));
    ENG(svr_initialize_em_values(reconstructionGPU));
    return 0;
  }

  // RG.cc:2695-2762.  voxel_num has one entry per device and its median indexes out of range for
  // one device (RG.cc:2714-2726), so no slice is ever "small" on the GPU path.
  int GaussianReconstructionGPU() {
    if (!sh.on) {
      int n = 0;
      ENG(svr_gaussian_reconstruction(reconstructionGPU, &n));
    } else {
      ENG(svr_gaussian_reconstruction_local(reconstructionGPU));
      ENG(sh.allreduce_pair(SVR_BUF_RECONSTRUCTED, 2 * svr_volume_voxels(reconstructionGPU)));
      int n = 0;
      ENG(svr_gaussian_reconstruction_finish(reconstructionGPU, &n));
    }
    _small_slices.clear();
    return 0;
  }

  // RG.cc:1163-1175
  int SimulateSlicesGPU() {
    if (!sh.on) {
      ENG(svr_simulate_slices(reconstructionGPU, nullptr));
      _inside_pending = true;
      return 0;
    }
    ENG(svr_simulate_slices(reconstructionGPU, nullptr));   // sharded: this rank's flags come over with the M-step's sums,
    _inside_pending = true;                                 // the other ranks' with the exchange that follows
    _inside_stale = true;
    return 0;
  }

  // RG.cc:2988-3019
  int InitializeRobustStatisticsGPU() {
    if (int rc = settle()) return rc;
    if (!sh.on) {
      ENG(svr_initialize_robust_statistics(reconstructionGPU, &_sigma_gpu));
    } else {
      double s2[2], t[2] = {0, 0};
      ENG(svr_robust_statistics_sums(reconstructionGPU, s2));
      std::vector<double> all;
      ENG(exchange(s2, 2, all, nullptr));         // (brings the other ranks' slice_inside along)
      for (int r = 0; r < sh.coll.world; ++r) { t[0] += all[2 * r]; t[1] += all[2 * r + 1]; }
      _sigma_gpu = (float)t[0] / (float)t[1];
    }
    for (int i = 0; i < ns; ++i)
      if (!_slice_inside_gpu[i]) _slice_weight_gpu[i] = 0;
    for (size_t i = 0; i < _force_excluded.size(); i++) _slice_weight_gpu[_force_excluded[i]] = 0;
    _sigma_s_gpu = 0.025f;
    _mix_gpu = 0.9f;
    _mix_s_gpu = 0.9f;
    _m_gpu = (float)(1.0f / (2.1f * _max_intensity - 1.9f * _min_intensity));
    ENG(svr_update_scale_vector(reconstructionGPU, local(_scale_gpu), local(_slice_weight_gpu)));
    return 0;
  }

  double G(double x, double s) { return _step * exp(-x * x / (2 * s)) / (sqrt(6.28 * s)); }   // RG.h:529-532

  // RG.cc:3184-3440: voxel posteriors on the GPU, slice-level EM on the host
  int EStepGPU() {
    if (use_device_slice_em()) {
      // [M-step] + E-step + the slice-level EM without a wait and without a host exchange (csrc/svr_em.inc)
      if (int rc = push_state()) return rc;
      const int iter = _mstep_pending;
      _mstep_pending = 0;
      void *send = nullptr, *recv = nullptr;
      if (iter > 0 && sh.on) {                         // the ranks' M-step sums meet on the device (round 4)
        ENG(svr_mstep_partial(reconstructionGPU, sh.coll.world, &send, &recv));
        if (int rc = sh.before_device_collective()) return fail(rc, "svr_stream_sync");
        if (int rc = sh.coll.allgather_device(sh.coll.user, send, recv, 16)) return fail(rc, "allgather_device (M-step sums)");
      }
      size_t n = 0;
      ENG(svr_mstep_estep_device(reconstructionGPU, iter, (float)_step, &send, &recv, &n));
      if (sh.on) {                                     // every rank's potentials, scales and slice_inside: one all-gather of 3 x maxn floats
        if (int rc = sh.before_device_collective()) return fail(rc, "svr_stream_sync");
        if (int rc = sh.coll.allgather_device(sh.coll.user, send, recv, n)) return fail(rc, "allgather_device (slice potentials)");
      }
      ENG(svr_slice_em_run(reconstructionGPU));
      _em_on_host = false;
      _scale_pending = _inside_pending = _scale_stale = _inside_stale = false;   // (they travelled with the gather)
      return 0;
    }
    std::vector<float> loc(hi - lo);
    if (_mstep_pending && sh.on) {
      // sharded (round 4): the M-step's five sums of every rank meet ON THE DEVICE (the launcher's all-gather on the engine's stream),
      // are added up there in rank order, and the E-step runs on the result: one wait and one host exchange (the potentials') per SR
      // iteration instead of two of each
      const int iter = _mstep_pending;
      _mstep_pending = 0;
      void *send = nullptr, *recv = nullptr;
      ENG(svr_mstep_partial(reconstructionGPU, sh.coll.world, &send, &recv));
      if (int rc = sh.before_device_collective()) return fail(rc, "svr_stream_sync");
      if (int rc = sh.coll.allgather_device(sh.coll.user, send, recv, 16)) return fail(rc, "allgather_device (M-step sums)");
      float em3[3] = {_sigma_gpu, _mix_gpu, _m_gpu};
      std::vector<float> sc(_scale_pending ? hi - lo : 0);
      std::vector<unsigned char> inside(_inside_pending ? hi - lo : 0);
      ENG(svr_mstep_estep_ranks(reconstructionGPU, sh.coll.world, iter, (float)_step, em3, loc.data(), _scale_pending ? sc.data() : nullptr,
                                _inside_pending ? inside.data() : nullptr));
      _sigma_gpu = em3[0]; _mix_gpu = em3[1]; _m_gpu = em3[2];
      if (_scale_pending) std::copy(sc.begin(), sc.end(), _scale_gpu.begin() + lo);
      if (_inside_pending) for (int i = 0; i < hi - lo; ++i) _slice_inside_gpu[lo + i] = inside[i] != 0;
      _scale_pending = _inside_pending = false;
    } else if (_mstep_pending) {                       // one rank: M-step + E-step + whatever is still on the device, one wait
      const int iter = _mstep_pending;
      _mstep_pending = 0;
      float em3[3] = {_sigma_gpu, _mix_gpu, _m_gpu};
      std::vector<float> sc(_scale_pending ? ns : 0);
      std::vector<unsigned char> inside(_inside_pending ? ns : 0);
      ENG(svr_mstep_estep(reconstructionGPU, iter, (float)_step, em3, loc.data(), _scale_pending ? sc.data() : nullptr,
                          _inside_pending ? inside.data() : nullptr));
      _sigma_gpu = em3[0]; _mix_gpu = em3[1]; _m_gpu = em3[2];
      if (_scale_pending) _scale_gpu = sc;
      if (_inside_pending) for (int i = 0; i < ns; ++i) _slice_inside_gpu[i] = inside[i] != 0;
      _scale_pending = _inside_pending = false;
    } else {
      if (int rc = settle()) return rc;
      ENG(svr_estep(reconstructionGPU, _m_gpu, _sigma_gpu, _mix_gpu, loc.data()));
    }
    _slice_potential_gpu.assign(ns, 0.0f);
    std::copy(loc.begin(), loc.end(), _slice_potential_gpu.begin() + lo);
    if (sh.on) { std::vector<double> none; ENG(exchange(nullptr, 0, none, &_slice_potential_gpu)); }   // (and the scale vector)
    // from here on in the reference's slice order (identity unless svrh_set_unit_order): RG.cc:3282-3420 op for op
    std::vector<float> slice_potential_gpu = to_ref(_slice_potential_gpu);
    const std::vector<float> _scale_gpu = to_ref(this->_scale_gpu);
    std::vector<float> _slice_weight_gpu = to_ref(this->_slice_weight_gpu);
    int inputIndex;
    for (size_t i = 0; i < _force_excluded.size(); i++) slice_potential_gpu[ref_index(_force_excluded[i])] = -1;
    for (size_t i = 0; i < _small_slices.size(); i++) slice_potential_gpu[ref_index(_small_slices[i])] = -1;
    for (inputIndex = 0; inputIndex < ns; inputIndex++)
      if ((_scale_gpu[inputIndex] < 0.2) || (_scale_gpu[inputIndex] > 5)) slice_potential_gpu[inputIndex] = -1;

    double sum = 0, den = 0, sum2 = 0, den2 = 0, maxs = 0, mins = 1;
    for (inputIndex = 0; inputIndex < ns; inputIndex++)
      if (slice_potential_gpu[inputIndex] >= 0) {
        sum += slice_potential_gpu[inputIndex] * _slice_weight_gpu[inputIndex];
        den += _slice_weight_gpu[inputIndex];
        sum2 += slice_potential_gpu[inputIndex] * (1.0 - _slice_weight_gpu[inputIndex]);
        den2 += (1.0 - _slice_weight_gpu[inputIndex]);
        if (slice_potential_gpu[inputIndex] > maxs) maxs = slice_potential_gpu[inputIndex];
        if (slice_potential_gpu[inputIndex] < mins) mins = slice_potential_gpu[inputIndex];
      }
    if (den > 0) _mean_s_gpu = (float)(sum / den);
    else _mean_s_gpu = (float)mins;
    if (den2 > 0) _mean_s2_gpu = (float)(sum2 / den2);
    else _mean_s2_gpu = (float)((maxs + _mean_s_gpu) / 2.0);

    sum = 0; den = 0; sum2 = 0; den2 = 0;
    for (inputIndex = 0; inputIndex < ns; inputIndex++)
      if (slice_potential_gpu[inputIndex] >= 0) {
        sum += (slice_potential_gpu[inputIndex] - _mean_s_gpu) * (slice_potential_gpu[inputIndex] - _mean_s_gpu) *
               _slice_weight_gpu[inputIndex];
        den += _slice_weight_gpu[inputIndex];
        sum2 += (slice_potential_gpu[inputIndex] - _mean_s2_gpu) * (slice_potential_gpu[inputIndex] - _mean_s2_gpu) *
                (1 - _slice_weight_gpu[inputIndex]);
        den2 += (1 - _slice_weight_gpu[inputIndex]);
      }
    if ((sum > 0) && (den > 0)) {
      _sigma_s_gpu = (float)(sum / den);
      if (_sigma_s_gpu < _step * _step / 6.28) _sigma_s_gpu = (float)(_step * _step / 6.28);
    } else {
      _sigma_s_gpu = 0.025f;
    }
    if ((sum2 > 0) && (den2 > 0)) {
      _sigma_s2_gpu = (float)(sum2 / den2);
      if (_sigma_s2_gpu < _step * _step / 6.28) _sigma_s2_gpu = (float)(_step * _step / 6.28);
    } else {
      _sigma_s2_gpu = (_mean_s2_gpu - _mean_s_gpu) * (_mean_s2_gpu - _mean_s_gpu) / 4;
      if (_sigma_s2_gpu < _step * _step / 6.28) _sigma_s2_gpu = (float)(_step * _step / 6.28);
    }

    double gs1, gs2;
    for (inputIndex = 0; inputIndex < ns; inputIndex++) {
      if (slice_potential_gpu[inputIndex] == -1) { _slice_weight_gpu[inputIndex] = 0; continue; }
      if ((den <= 0) || (_mean_s2_gpu <= _mean_s_gpu)) { _slice_weight_gpu[inputIndex] = 1; continue; }
      if (slice_potential_gpu[inputIndex] < _mean_s2_gpu) gs1 = G(slice_potential_gpu[inputIndex] - _mean_s_gpu, _sigma_s_gpu);
      else gs1 = 0;
      if (slice_potential_gpu[inputIndex] > _mean_s_gpu) gs2 = G(slice_potential_gpu[inputIndex] - _mean_s2_gpu, _sigma_s2_gpu);
      else gs2 = 0;
      double likelihood = gs1 * _mix_s_gpu + gs2 * (1 - _mix_s_gpu);
      if (likelihood > 0) _slice_weight_gpu[inputIndex] = (float)(gs1 * _mix_s_gpu / likelihood);
      else {
        if (slice_potential_gpu[inputIndex] <= _mean_s_gpu) _slice_weight_gpu[inputIndex] = 1;
        if (slice_potential_gpu[inputIndex] >= _mean_s2_gpu) _slice_weight_gpu[inputIndex] = 0;
        if ((slice_potential_gpu[inputIndex] < _mean_s2_gpu) && (slice_potential_gpu[inputIndex] > _mean_s_gpu))
          _slice_weight_gpu[inputIndex] = 1;
      }
    }
    sum = 0;
    int num = 0;
    for (inputIndex = 0; inputIndex < ns; inputIndex++)
      if (slice_potential_gpu[inputIndex] >= 0) { sum += _slice_weight_gpu[inputIndex]; num++; }
    if (num > 0) _mix_s_gpu = (float)(sum / num);
    else _mix_s_gpu = 0.9f;
    this->_slice_weight_gpu = from_ref(_slice_weight_gpu);
    _slice_potential_gpu = from_ref(slice_potential_gpu);
    ENG(svr_update_slice_weights(reconstructionGPU, local(this->_slice_weight_gpu)));
    return 0;
  }

  // RG.cc:3751-3757
  int ScaleGPU() {
    if (!sh.on) {
      ENG(svr_calculate_scale_vector(reconstructionGPU, nullptr));
      _scale_pending = true;
      return 0;
    }
    ENG(svr_calculate_scale_vector(reconstructionGPU, nullptr));   // sharded: fetched with the M-step's sums (or by settle)
    _scale_pending = true;
    _scale_stale = true;                               // read next in the E-step, whose exchange completes it
    return 0;
  }

  // RG.cc:4024-4036
  int SuperresolutionGPU(int iter) {
    // (the slice weights of a device-side EM are already where the scatter reads them: NULL = keep the device's)
    const float *sw = _em_on_host ? local(_slice_weight_gpu) : nullptr;
    if (!sw) ENG(svr_slice_em_apply_weights(reconstructionGPU));   // (whatever anybody sent the engine in between: the EM's weights, device to device)
    if (!sh.on) {
      ENG(svr_superresolution(reconstructionGPU, iter, sw, _adaptive, (float)_alpha,
                              (float)_min_intensity, (float)_max_intensity, (float)_delta, (float)_lambda,
                              _global_bias_correction, _sigma_bias, _low_intensity_cutoff));
    } else {
      ENG(sh.superresolution(sw, _adaptive, (float)_alpha, (float)_min_intensity, (float)_max_intensity, (float)_delta,
                             (float)_lambda));
    }
    return 0;
  }

  // RG.cc:4214-4223 + Reconstruction::MStep host part (reconstruction_cuda2.cu:3016-3071)
  int MStepGPU(int iter) {
    if (!sh.on) {
      if (_mstep_pending) { if (int rc = settle()) return rc; }   // (an M-step after an M-step; the scale vector and slice_inside stay pending for the fused fetch)
      if (iter > 0) {
        _mstep_pending = iter;                         // runs with the E-step that follows (reconstruction.cc:1093-1108), or in settle
        return 0;
      }
      if (int rc = settle()) return rc;                // iter == 0: the host's sigma / mix are the M-step's inputs -- what the device holds comes down first
      ENG(svr_mstep(reconstructionGPU, iter, (float)_step, &_sigma_gpu, &_mix_gpu, &_m_gpu));
      return 0;
    }
    if (device_em && sh.coll.allgather_device && iter > 0) {
      if (_mstep_pending) { if (int rc = settle()) return rc; }
      _mstep_pending = iter;                           // runs with the E-step that follows, its sums meeting on the device (EStepGPU), or in settle
      return 0;
    }
    return mstep_exchange(iter);
  }
  // the M-step of a sharded run through the hosts: this rank's five sums, one exchange, the scalars on the host
  bool device_em = getenv("SVR_DEVICE_EM") ? atoi(getenv("SVR_DEVICE_EM")) != 0 : true;
  int mstep_exchange(int iter) {
    double s5[5];
    {
      std::vector<float> sc(_scale_pending ? hi - lo : 0);
      std::vector<unsigned char> inside(_inside_pending ? hi - lo : 0);
      ENG(svr_mstep_sums_fetch(reconstructionGPU, s5, _scale_pending ? sc.data() : nullptr, _inside_pending ? inside.data() : nullptr));
      if (_scale_pending) std::copy(sc.begin(), sc.end(), _scale_gpu.begin() + lo);
      if (_inside_pending) for (int i = 0; i < hi - lo; ++i) _slice_inside_gpu[lo + i] = inside[i] != 0;
      _scale_pending = _inside_pending = false;
    }
    std::vector<double> all;
    ENG(exchange(s5, 5, all, nullptr));           // three sums, a minimum, a maximum: one collective
    s5[0] = s5[1] = s5[2] = 0;
    for (int r = 0; r < sh.coll.world; ++r) {
      for (int k = 0; k < 3; ++k) s5[k] += all[5 * r + k];
      s5[3] = r ? std::min(s5[3], all[5 * r + 3]) : all[3];
      s5[4] = r ? std::max(s5[4], all[5 * r + 4]) : all[4];
    }
    float sigma = (float)s5[0], mix = (float)s5[1], num = (float)s5[2];
    float min_ = std::min(3.402823466e+38f, (float)s5[3]);
    float max_ = std::max(1.175494351e-38f, (float)s5[4]);
    float step = (float)_step;
    if (mix > 0) _sigma_gpu = sigma / mix;
    if (_sigma_gpu < step * step / 6.28f) _sigma_gpu = step * step / 6.28f;
    if (iter > 1) _mix_gpu = mix / num;
    _m_gpu = 1.0f / (max_ - min_);
    return 0;
  }

  // RG.cc:3904-3913, 4653-4655
  int BiasGPU() { ENG(svr_correct_bias(reconstructionGPU, _sigma_bias, _global_bias_correction)); return 0; }
  int NormaliseBiasGPU(int iter) {
    if (!sh.on) { ENG(svr_normalise_bias(reconstructionGPU, iter, _sigma_bias)); return 0; }
    ENG(svr_normalise_bias_local(reconstructionGPU));
    // the bias volume is a single float[Nv] message
    ENG(sh.allreduce_pair(SVR_BUF_BIAS_VOLUME, svr_volume_voxels(reconstructionGPU)));
    ENG(svr_normalise_bias_finish(reconstructionGPU, _sigma_bias));
    return 0;
  }

  int MaskVolumeGPU() { ENG(svr_mask_volume(reconstructionGPU)); return 0; }   // RG.cc:5319-5323

  int ScaleVolumeGPU() {
    if (!sh.on) { ENG(svr_scale_volume(reconstructionGPU)); return 0; }
    double s2[2];
    ENG(svr_scale_volume_sums(reconstructionGPU, s2));
    ENG(sh.coll.allreduce_host(sh.coll.user, s2, 2, 0));
    ENG(svr_scale_volume_apply(reconstructionGPU, (float)(s2[0] / s2[1])));
    return 0;
  }

  // reconstruction.cc:1013-1108 with bias correction off
  int sr_iteration(int i) {
    int rc;
    if (_intensity_matching) {                                               // reconstruction.cc:1018-1045
      if (!_disableBiasC && _sigma_bias > 0)                                 // reconstruction.cc:1032-1037
        if ((rc = BiasGPU())) return rc;
      if ((rc = ScaleGPU())) return rc;
    }
    if ((rc = SuperresolutionGPU(i + 1))) return rc;
    if (_intensity_matching && !_disableBiasC && _sigma_bias > 0 && !_global_bias_correction)   // reconstruction.cc:1062-1076
      if ((rc = NormaliseBiasGPU(i))) return rc;
    if ((rc = SimulateSlicesGPU())) return rc;
    if ((rc = MStepGPU(i + 1))) return rc;
    return EStepGPU();
  }

  // ---- GPU slice-to-volume registration, host side ------------------------------------------
  struct M4 { double m[16]; };
  static M4 mul(const M4 &a, const M4 &b) {
    M4 c;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        double t = 0;
        for (int k = 0; k < 4; ++k) t += a.m[4 * i + k] * b.m[4 * k + j];
        c.m[4 * i + j] = t;
      }
    return c;
  }
  static M4 ident() { M4 c; for (int i = 0; i < 16; ++i) c.m[i] = (i % 5 == 0) ? 1.0 : 0.0; return c; }
  // irtkBaseImage::GetImageToWorldMatrix / GetWorldToImageMatrix (irtkBaseImage.cc:79-147)
  static M4 image_to_world(const svr_image_attr &a) {
    M4 t1 = ident(), sc = ident(), rot = ident(), t2 = ident();
    t1.m[3] = -(a.nx - 1) / 2.0; t1.m[7] = -(a.ny - 1) / 2.0; t1.m[11] = -(a.nz - 1) / 2.0;
    sc.m[0] = a.dx; sc.m[5] = a.dy; sc.m[10] = a.dz;
    for (int k = 0; k < 3; ++k) { rot.m[4 * k] = a.xaxis[k]; rot.m[4 * k + 1] = a.yaxis[k]; rot.m[4 * k + 2] = a.zaxis[k]; }
    for (int k = 0; k < 3; ++k) t2.m[4 * k + 3] = a.origin[k];
    return mul(t2, mul(rot, mul(sc, t1)));
  }
  static M4 world_to_image(const svr_image_attr &a) {
    M4 t1 = ident(), rot = ident(), sc = ident(), t2 = ident();
    for (int k = 0; k < 3; ++k) t1.m[4 * k + 3] = -a.origin[k];
    for (int k = 0; k < 3; ++k) { rot.m[k] = a.xaxis[k]; rot.m[4 + k] = a.yaxis[k]; rot.m[8 + k] = a.zaxis[k]; }
    sc.m[0] = 1.0 / a.dx; sc.m[5] = 1.0 / a.dy; sc.m[10] = 1.0 / a.dz;
    t2.m[3] = (a.nx - 1) / 2.0; t2.m[7] = (a.ny - 1) / 2.0; t2.m[11] = (a.nz - 1) / 2.0;
    return mul(t2, mul(sc, mul(rot, t1)));
  }
  static int irtk_round(double x) { return x > 0 ? (int)(x + 0.5) : (int)(x - 0.5); }   // irtkCommon.h:85-88

  std::vector<svr_image_attr> _slices_resampled_attr;   // attributes of `_slices_resampled` (RG.cc:2104-2119)
  std::vector<float> _reg_combined;                     // combinedStacks (RG.cc:2134-2160)
  int _reg_size[3] = {0, 0, 0};

  // irtkResamplingWithPadding<irtkRealPixel>(d, d, d, -1).Run() on one slice, plane 0 of the result
  // (IRTKSimple2/image++/src/irtkResamplingWithPadding.cc:198-252, 254-443)
  static void resample_plane0(const float *img, int row_pitch, const svr_image_attr &a, double d, svr_image_attr &oa,
                              std::vector<double> &out) {
    oa = a;
    oa.nx = irtk_round(a.nx * a.dx / d); oa.ny = irtk_round(a.ny * a.dy / d); oa.nz = irtk_round(a.nz * a.dz / d);
    oa.dx = oa.dy = oa.dz = d;
    if (oa.nx < 1) { oa.nx = 1; oa.dx = a.dx; }
    if (oa.ny < 1) { oa.ny = 1; oa.dy = a.dy; }
    if (oa.nz < 1) { oa.nz = 1; oa.dz = a.dz; }
    const M4 m = mul(world_to_image(a), image_to_world(oa));
    out.assign((size_t)oa.nx * oa.ny, -1.0);
    for (int j = 0; j < oa.ny; ++j)
      for (int i = 0; i < oa.nx; ++i) {
        const double x = m.m[0] * i + m.m[1] * j + m.m[3], y = m.m[4] * i + m.m[5] * j + m.m[7],
                     z = m.m[8] * i + m.m[9] * j + m.m[11];               // output k = 0
        const int u = (int)floor(x), v = (int)floor(y), w = (int)floor(z);
        const double fx = x - u, fy = y - v, fz = z - w;
        double val = 0, sum = 0;
        int pad = 8;
        for (int du = 0; du < 2; ++du)                                     // the reference's order w1..w8
          for (int dv = 0; dv < 2; ++dv)
            for (int dw = 0; dw < 2; ++dw) {
              const double wt = (du ? fx : 1 - fx) * (dv ? fy : 1 - fy) * (dw ? fz : 1 - fz);
              const int p = u + du, q = v + dv, r = w + dw;
              if (p >= 0 && p < a.nx && q >= 0 && q < a.ny && r >= 0 && r < a.nz) {
                const double g = img[(size_t)q * row_pitch + p];            // nz == 1
                if (g != -1.0) { --pad; val += g * wt; sum += wt; }
              } else {
                --pad;
              }
            }
        if (pad < 4 && sum > 0) out[(size_t)j * oa.nx + i] = val / sum;
      }
  }

  // PrepareRegistrationSlices RG.cc:2104-2181
  int PrepareRegistrationSlices(const float *slices, int sx, int sy, const svr_image_attr *attrs, double d) {
    const int n = hi - lo;
    _slices_resampled_attr.resize(n);
    std::vector<std::vector<double>> res(n);
    int mx = 0, my = 0;
    for (int i = 0; i < n; ++i) {
      if (attrs[i].nz != 1) { err = "PrepareRegistrationSlices: slices must have one plane"; return SVR_E_ARG; }
      resample_plane0(slices + (size_t)i * sx * sy, sx, attrs[i], d, _slices_resampled_attr[i], res[i]);
      mx = std::max(mx, _slices_resampled_attr[i].nx);
      my = std::max(my, _slices_resampled_attr[i].ny);
    }
    _reg_size[0] = mx; _reg_size[1] = my; _reg_size[2] = n;
    _reg_combined.assign((size_t)n * mx * my, -1.0f);                       // combinedStacks = -1 (RG.cc:2143)
    std::vector<float> i2w(16 * (size_t)n);
    for (int i = 0; i < n; ++i) {
      const svr_image_attr &a = _slices_resampled_attr[i];
      for (int y = 0; y < a.ny; ++y)
        for (int x = 0; x < a.nx; ++x)
          _reg_combined[((size_t)i * my + y) * mx + x] = (float)res[i][(size_t)y * a.nx + x];
      const M4 m = image_to_world(a);
      for (int k = 0; k < 16; ++k) i2w[16 * (size_t)i + k] = (float)m.m[k];
    }
    const uint32_t size[3] = {(uint32_t)mx, (uint32_t)my, (uint32_t)n};
    const float dim[3] = {(float)d, (float)d, (float)d};
    ENG(svr_init_reg_storage_volumes(reconstructionGPU, size, dim));
    ENG(svr_fill_reg_slices(reconstructionGPU, _reg_combined.data(), i2w.data()));
    return 0;
  }

  // SliceToVolumeRegistrationGPU RG.cc:2214-2290
  int SliceToVolumeRegistrationGPU(double *transformations) {
    const int n = hi - lo;
    if ((int)_slices_resampled_attr.size() != n) { err = "SliceToVolumeRegistrationGPU: PrepareRegistrationSlices first"; return SVR_E_STATE; }
    std::vector<float> transf(16 * (size_t)n), ofs(16 * (size_t)n);
    std::vector<M4> mos(n);
    for (int i = 0; i < n; ++i) {
      svr_image_attr a0 = _slices_resampled_attr[i];
      M4 mo = ident();
      for (int k = 0; k < 3; ++k) { mo.m[4 * k + 3] = a0.origin[k]; a0.origin[k] = 0.0; }   // RG.cc:2226-2236
      mos[i] = mo;
      M4 t;
      for (int k = 0; k < 16; ++k) t.m[k] = transformations[16 * (size_t)i + k];
      const M4 tm = mul(t, mo), o = image_to_world(a0);
      for (int k = 0; k < 16; ++k) { transf[16 * (size_t)i + k] = (float)tm.m[k]; ofs[16 * (size_t)i + k] = (float)o.m[k]; }
    }
    ENG(svr_update_resampled_slices_i2w(reconstructionGPU, ofs.data()));
    ENG(svr_prepare_slice_to_volume_reg(reconstructionGPU));
    ENG(svr_register_slices_to_volume(reconstructionGPU, transf.data()));
    for (int i = 0; i < n; ++i) {                                            // mat * mo^-1 (RG.cc:2262-2267)
      M4 t, moi = ident();
      for (int k = 0; k < 16; ++k) t.m[k] = (double)transf[16 * (size_t)i + k];
      for (int k = 0; k < 3; ++k) moi.m[4 * k + 3] = -mos[i].m[4 * k + 3];
      const M4 rres = mul(t, moi);
      for (int k = 0; k < 16; ++k) transformations[16 * (size_t)i + k] = rres.m[k];
    }
    return 0;
  }

  // reconstruction.cc:930-1140 (one outer iteration after registration)
  int reconstruct_iteration(int rec_iterations) {
    int rc;
    if ((rc = InitializeEMValuesGPU())) return rc;
    if ((rc = GaussianReconstructionGPU())) return rc;
    if ((rc = SimulateSlicesGPU())) return rc;
    if ((rc = InitializeRobustStatisticsGPU())) return rc;
    if ((rc = EStepGPU())) return rc;
    for (int i = 0; i < rec_iterations; ++i)
      if ((rc = sr_iteration(i))) return rc;
    return MaskVolumeGPU();
  }
#undef ENG
};

}  // namespace svr

struct svrh_recon {
  svr::irtkReconstruction impl;
  svrh_recon(svr_ctx *e, int n, int lo, int hi, const svr_collectives *c) : impl(e, n, lo, hi, c) {}
};

extern "C" {

svrh_recon *svrh_create(svr_ctx *engine, int n_slices_global, int slice_lo, int slice_hi,
                        const svr_collectives *coll) {
  if (!engine || n_slices_global <= 0 || slice_lo < 0 || slice_hi > n_slices_global || slice_lo > slice_hi) return nullptr;
  if (coll && coll->struct_size < SVR_COLLECTIVES_MIN_SIZE) return nullptr;      // (a launcher built against the struct of rounds 1-4)
  if (coll && coll->world > 1 && (!coll->allreduce_volume_pair || !coll->allreduce_host || !coll->allgather_slices))
    return nullptr;
  return new svrh_recon(engine, n_slices_global, slice_lo, slice_hi, coll);
}
void svrh_destroy(svrh_recon *r) { delete r; }
const char *svrh_last_error(const svrh_recon *r) { return r ? r->impl.err.c_str() : "null"; }
void svrh_set_intensity_range(svrh_recon *r, double mn, double mx) { r->impl._min_intensity = mn; r->impl._max_intensity = mx; }
void svrh_set_smoothing_parameters(svrh_recon *r, double delta, double lambda) { r->impl.SetSmoothingParameters(delta, lambda); }
void svrh_set_force_excluded(svrh_recon *r, const int *idx, int n) { (void)r->impl.settle(); r->impl._force_excluded.assign(idx, idx + n); }
int svrh_set_bias_correction(svrh_recon *r, int enable, double sigma_bias) {
  r->impl._disableBiasC = !enable;
  r->impl._sigma_bias = (float)sigma_bias;
  return svr_set_flags(r->impl.reconstructionGPU, !enable, 0);
}
int svrh_bias_gpu(svrh_recon *r) { return r->impl.BiasGPU(); }
int svrh_normalise_bias_gpu(svrh_recon *r, int iter) { return r->impl.NormaliseBiasGPU(iter); }
int svrh_initialize_em_values_gpu(svrh_recon *r) { return r->impl.InitializeEMValuesGPU(); }
int svrh_gaussian_reconstruction_gpu(svrh_recon *r) { return r->impl.GaussianReconstructionGPU(); }
int svrh_simulate_slices_gpu(svrh_recon *r) { return r->impl.SimulateSlicesGPU(); }
int svrh_initialize_robust_statistics_gpu(svrh_recon *r) { return r->impl.InitializeRobustStatisticsGPU(); }
int svrh_estep_gpu(svrh_recon *r) { return r->impl.EStepGPU(); }
int svrh_scale_gpu(svrh_recon *r) { return r->impl.ScaleGPU(); }
int svrh_superresolution_gpu(svrh_recon *r, int iter) { return r->impl.SuperresolutionGPU(iter); }
void svrh_set_intensity_matching(svrh_recon *r, int on) { if (r) r->impl._intensity_matching = on != 0; }
int svrh_mstep_gpu(svrh_recon *r, int iter) { return r->impl.MStepGPU(iter); }
int svrh_mask_volume_gpu(svrh_recon *r) { return r->impl.MaskVolumeGPU(); }
int svrh_scale_volume_gpu(svrh_recon *r) { return r->impl.ScaleVolumeGPU(); }
int svrh_sr_iteration(svrh_recon *r, int i) { return r->impl.sr_iteration(i); }
int svrh_reconstruct_iteration(svrh_recon *r, int n) { return r->impl.reconstruct_iteration(n); }

int svrh_prepare_registration_slices(svrh_recon *r, const float *slices, int sx, int sy, const svr_image_attr *attrs,
                                     double recon_voxel) {
  if (!r || !slices || !attrs || sx <= 0 || sy <= 0 || !(recon_voxel > 0)) return SVR_E_ARG;
  return r->impl.PrepareRegistrationSlices(slices, sx, sy, attrs, recon_voxel);
}
int svrh_slice_to_volume_registration_gpu(svrh_recon *r, double *transformations) {
  if (!r || !transformations) return SVR_E_ARG;
  return r->impl.SliceToVolumeRegistrationGPU(transformations);
}
int svrh_get_registration_slices(svrh_recon *r, int size3[3], float *data_or_null) {
  if (!r || !size3) return SVR_E_ARG;
  for (int k = 0; k < 3; ++k) size3[k] = r->impl._reg_size[k];
  if (data_or_null) std::copy(r->impl._reg_combined.begin(), r->impl._reg_combined.end(), data_or_null);
  return 0;
}

int svrh_set_unit_order(svrh_recon *r, const int *order_or_null) {
  if (!r) return SVR_E_ARG;
  svr::irtkReconstruction &m = r->impl;
  if (int rc = m.settle()) return rc;                  // (the device's copy of the slice-level state, if it is the current one, in the old numbering)
  m._sem_ready = false;                                // the device-side EM learns the new numbering at its next use
  if (!order_or_null) { m.order.clear(); return SVR_OK; }
  std::vector<char> seen(m.ns, 0);
  for (int k = 0; k < m.ns; ++k) {
    const int i = order_or_null[k];
    if (i < 0 || i >= m.ns || seen[i]) { m.err = "svrh_set_unit_order: not a permutation of the slices"; return SVR_E_ARG; }
    seen[i] = 1;
  }
  m.order.assign(order_or_null, order_or_null + m.ns);
  return SVR_OK;
}
void svrh_force_collectives(svrh_recon *r, int on) { if (r) { (void)r->impl.settle(); r->impl.sh.force(on != 0); } }
void svrh_set_slab_update(svrh_recon *r, int on) { if (r) r->impl.sh.slabs = on != 0; }

int svrh_get_state(svrh_recon *r, float *scale, float *slice_weight, float *slice_potential,
                   unsigned char *slice_inside, double s[8]) {
  if (!r) return SVR_E_ARG;
  svr::irtkReconstruction &m = r->impl;
  if (int rc = m.flush()) return rc;      // sharded: the scale vector / slice_inside of the other ranks may still be on their way
  if (scale) std::copy(m._scale_gpu.begin(), m._scale_gpu.end(), scale);
  if (slice_weight) std::copy(m._slice_weight_gpu.begin(), m._slice_weight_gpu.end(), slice_weight);
  if (slice_potential) std::copy(m._slice_potential_gpu.begin(), m._slice_potential_gpu.end(), slice_potential);
  if (slice_inside) std::copy(m._slice_inside_gpu.begin(), m._slice_inside_gpu.end(), slice_inside);
  if (s) {
    s[0] = m._sigma_gpu; s[1] = m._mix_gpu; s[2] = m._m_gpu; s[3] = m._mean_s_gpu; s[4] = m._mean_s2_gpu;
    s[5] = m._sigma_s_gpu; s[6] = m._sigma_s2_gpu; s[7] = m._mix_s_gpu;
  }
  return SVR_OK;
}

}  // extern "C"
