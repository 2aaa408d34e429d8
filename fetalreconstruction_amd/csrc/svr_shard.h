// svr_shard.h -- what the two host objects (svr::irtkReconstruction, svr::irtkPatchBasedReconstruction) share when the
// slices / patches are sharded over ranks: the unit range [lo, hi) of this rank, the launcher's collectives, the timed
// all-reduce of a volume pair and the ONE host-side exchange per step.
//
// The reference fans every step out over `devicesToUse` inside one process and adds the per-device volumes up on GPU 0
// (reconstruction_cuda2.cu:1413-1457, 2225-2239); its patch-based path is single-GPU (patchBasedReconMain.cpp:78,177-179,
// irtkPatchBasedReconstruction.cpp:402).  Here a rank keeps the whole volume and a contiguous range of units.
#ifndef SVR_SHARD_H
#define SVR_SHARD_H

#include <chrono>
#include <cstdlib>
#include <vector>

#include "../../include/svr_host.h"

namespace svr {

struct Shard {
  svr_ctx *e = nullptr;
  int n = 0, lo = 0, hi = 0;          // units (slices / patches): global count, this rank's range
  svr_collectives coll;
  bool given = false, on = false;     // collectives supplied / in use (world > 1, or forced for tests at world 1)

  void init(svr_ctx *engine, int n_global, int lo_, int hi_, const svr_collectives *c) {
    e = engine; n = n_global; lo = lo_; hi = hi_;
    given = c != nullptr;
    if (c) coll = *c;
    else { coll.user = nullptr; coll.rank = 0; coll.world = 1; coll.allreduce_volume_pair = nullptr;
           coll.allreduce_host = nullptr; coll.allgather_slices = nullptr; coll.reduce_scatter_device = nullptr; coll.allgather_device = nullptr; coll.on_engine_stream = 0; }
    on = given && coll.world > 1;
    if (const char *v = getenv("SVR_SLAB_UPDATE")) slabs = atoi(v) != 0;     // (tests: the two forms of the volume update side by side)
  }
  void force(bool f) { on = given && (coll.world > 1 || f); }   // test hook: world 1 through the callbacks

  // a collective that does not run on the engine's stream must not start before what the engine has queued there (the pack
  // kernel, the scatter) is done
  int before_device_collective() { return coll.on_engine_stream ? 0 : svr_stream_sync(e); }

  // in-place sum of a device buffer of the engine over the ranks, on the engine's stream; HIP events around it when the
  // engine's timers are on (SVR_T_ALLREDUCE: what a rank waits for = its own wait for the slowest rank + the ring)
  int allreduce_pair(int buffer, size_t n_floats) {
    int rc = svr_timer_begin(e, SVR_T_ALLREDUCE);
    if (rc) return rc;
    // only the mask's bounding box travels (the scatter writes mask voxels only: 47 % of the S8 volume; two device copies of
    // the box against half of a 163 MB ring all-reduce); the whole buffer where that gains nothing
    void *packed = nullptr;
    size_t n_packed = 0;
    if ((rc = svr_pair_pack(e, buffer, n_floats, &packed, &n_packed))) return rc;
    if ((rc = before_device_collective())) return rc;
    if (packed) {
      if ((rc = coll.allreduce_volume_pair(coll.user, packed, n_packed))) return rc;
      if ((rc = svr_pair_unpack(e, buffer, n_floats))) return rc;
    } else {
      rc = coll.allreduce_volume_pair(coll.user, svr_device_ptr(e, buffer), n_floats);
      if (rc) return rc;
    }
    return svr_timer_end(e, SVR_T_ALLREDUCE);
  }

  // The volume update of an SR iteration (after svr_superresolution_backproject).  By z-slabs when the launcher supplies the two device
  // collectives: reduce-scatter of addon | cmap at the mask's voxels -> the rank's slab -> all-gather of the new volume
  // (csrc/svr_slab.inc: less than half the bytes of the all-reduce, and the update's time divides by the ranks).  Otherwise
  // all-reduce of the pair + the update replicated on every rank.  Same bits either way.
  bool slabs = true;       // (svrh_set_slab_update / pvrh_set_slab_update: tests compare the two forms)
  // scatter + update of a sharded SR iteration; nothing waits for the device when the collectives run on the engine's stream
  int superresolution(const float *unit_weights, int adaptive, float alpha, float min_i, float max_i, float delta, float lambda) {
    const int async = coll.on_engine_stream ? 1 : 0;
    int rc = svr_set_option(e, "sr_no_wait", async);
    if (!rc) rc = svr_superresolution_backproject(e, unit_weights);
    if (!rc) rc = update(adaptive, alpha, min_i, max_i, delta, lambda);
    const int rc2 = svr_set_option(e, "sr_no_wait", 0);
    return rc ? rc : rc2;
  }
  int update(int adaptive, float alpha, float min_i, float max_i, float delta, float lambda) {
    int rc;
    if (!(slabs && coll.reduce_scatter_device && coll.allgather_device)) {
      if ((rc = allreduce_pair(SVR_BUF_ADDON, 2 * svr_volume_voxels(e)))) return rc;
      return svr_superresolution_update(e, adaptive, alpha, min_i, max_i, delta, lambda);
    }
    size_t nrs = 0, nag = 0;
    void *send = nullptr, *recv = nullptr;
    if ((rc = svr_slab_plan(e, coll.world, coll.rank, &nrs, &nag))) return rc;
    if ((rc = svr_timer_begin(e, SVR_T_REDUCE_SCATTER))) return rc;
    if ((rc = svr_slab_rs_pack(e, &send, &recv))) return rc;
    if ((rc = before_device_collective())) return rc;
    if ((rc = coll.reduce_scatter_device(coll.user, send, recv, nrs))) return rc;
    if ((rc = svr_timer_end(e, SVR_T_REDUCE_SCATTER))) return rc;
    if ((rc = svr_slab_update(e, adaptive, alpha, min_i, max_i, delta, lambda, &send, &recv))) return rc;
    if ((rc = svr_timer_begin(e, SVR_T_ALLGATHER))) return rc;
    if ((rc = before_device_collective())) return rc;
    if ((rc = coll.allgather_device(coll.user, send, recv, nag))) return rc;
    if ((rc = svr_slab_finish(e))) return rc;
    return svr_timer_end(e, SVR_T_ALLGATHER);
  }

  // ONE host collective per exchange.  Every host-side collective is a stream synchronisation plus a small collective
  // launch (~60-100 us), so everything a step has to share travels as one SUM all-reduce of a vector in which a rank fills
  // only its own entries (x + 0 is exact: an all-gather); sums over ranks are then taken by the caller in rank order, the
  // same bits on every rank and every run.  Layout (its length depends on the call site only, never on a rank's state):
  //   [world][n_mine] the ranks' scalars | [world] which vectors a rank sent | up to three n-sized vectors.
  // vec[k] = global vector (complete on return) or NULL; a rank sends its own range [lo, hi) of it.  The ranks must agree
  // on which vectors travel: a rank whose operator sequence differs from the others' is an error, not a silent mix-up.
  int exchange(const double *mine, int n_mine, std::vector<double> &all, std::vector<float> *vec[3]) {
    const auto t0 = std::chrono::steady_clock::now();
    const int W = coll.world, R = coll.rank;
    const int o_flag = n_mine * W, o_vec = o_flag + W;
    const int total = o_vec + 3 * n;
    std::vector<double> v((size_t)total, 0.0);
    for (int k = 0; k < n_mine; ++k) v[(size_t)R * n_mine + k] = mine[k];
    int flags = 0;
    for (int k = 0; k < 3; ++k)
      if (vec[k]) {
        flags |= 1 << k;
        for (int i = lo; i < hi; ++i) v[(size_t)o_vec + (size_t)k * n + i] = (*vec[k])[i];
      }
    v[o_flag + R] = (double)(flags + 1);
    const int rc = coll.allreduce_host(coll.user, v.data(), total, 0);
    if (rc) return rc;
    for (int r = 0; r < W; ++r)
      if ((int)v[o_flag + r] != flags + 1) return SVR_E_STATE;            // the ranks are not in the same step
    all.assign(v.begin(), v.begin() + o_flag);
    for (int k = 0; k < 3; ++k)
      if (vec[k])
        for (int i = 0; i < n; ++i) (*vec[k])[i] = (float)v[(size_t)o_vec + (size_t)k * n + i];
    (void)svr_timer_add(e, SVR_T_EXCHANGE, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return 0;
  }
};

}  // namespace svr
#endif
