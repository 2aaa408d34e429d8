// svr_shard.h -- what the two host objects (svr::irtkReconstruction, svr::irtkPatchBasedReconstruction) share when the
// slices / patches are sharded over ranks: the unit range [lo, hi) of this rank, the launcher's collectives, the timed
// all-reduce of a volume pair and the ONE host-side exchange per step.
//
// The reference fans every step out over `devicesToUse` inside one process and adds the per-device volumes up on GPU 0
// (reconstruction_cuda2.cu:1413-1457, 2225-2239); its patch-based path is single-GPU (patchBasedReconMain.cpp:78,177-179,
// irtkPatchBasedReconstruction.cpp:402).  Here a rank keeps the whole volume and a contiguous range of units.
#ifndef SVR_SHARD_H
#define SVR_SHARD_H

#include <math.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/svr_host.h"

namespace svr {

// The numbering of a sharded run and every rank's range of it (the launchers: csrc/svr_cli.cpp, csrc/pvr_cli.cpp; bench.py does the
// same through fetalreconstruction_amd/sharding.py shard_units, operation for operation).  work[i], stack[i]: estimated PSF work and
// stack of unit i in the reference's order (stack after stack).  Rank r takes the r-th of `world` work-balanced segments of EVERY
// stack: a rank's units are neighbours in space, so it touches about 1 / world of the volume's (cell, plane) items per stack
// orientation instead of all the items of the stacks it holds (less to stage, less to combine: csrc/svr_cell.inc), and every rank
// holds the same mix of orientations.  The cut points of a stack are placed against the running total over the stacks before it, so
// the ranks' totals differ by about one unit's work whatever the number of stacks.  order[k] = reference index of unit k of the
// sharded numbering (rank after rank; inside a rank stack after stack), [lo[r], hi[r]) = rank r's range of it.
// (The reference shards by slice count in slice order and drops the remainder: reconstruction_cuda2.cu:1413-1457.)
inline void spatial_order(const std::vector<double> &work, const std::vector<int> &stack, int world, std::vector<int> &order, std::vector<int> &lo,
                          std::vector<int> &hi) {
  const int n = (int)work.size();
  std::vector<std::vector<int>> parts(world);
  std::vector<double> done(world + 1, 0.0);
  double total = 0.0;
  int at = 0;
  while (at < n) {
    int end = at;
    while (end < n && stack[end] == stack[at]) ++end;                  // the units of one stack: [at, end)
    const int m = end - at;
    std::vector<double> cum(m + 1, 0.0);
    for (int i = 0; i < m; ++i) cum[i + 1] = cum[i] + (work[at + i] + 1e-9);
    total += cum[m];
    std::vector<int> cuts(1, 0);
    for (int r = 1; r < world; ++r) {
      const double want = total * r / world - done[r];
      int b = 0;
      double best = fabs(cum[0] - want);
      for (int i = 1; i <= m; ++i) { const double d = fabs(cum[i] - want); if (d < best) { best = d; b = i; } }
      b = std::min(std::max(b, cuts.back()), m);
      cuts.push_back(b);
    }
    cuts.push_back(m);
    for (int r = 0; r < world; ++r) {
      for (int i = cuts[r]; i < cuts[r + 1]; ++i) parts[r].push_back(at + i);
      done[r + 1] += cum[cuts[r + 1]];
    }
    at = end;
  }
  order.clear(); lo.assign(world, 0); hi.assign(world, 0);
  for (int r = 0; r < world; ++r) {
    lo[r] = (int)order.size();
    order.insert(order.end(), parts[r].begin(), parts[r].end());
    hi[r] = (int)order.size();
  }
}
// rows of `width` elements of a per-unit array, from the reference's order into the sharded numbering: out[k] = in[order[k]]
template <class T> void permute_rows(std::vector<T> &v, size_t width, const std::vector<int> &order) {
  std::vector<T> out(v.size());
  for (size_t k = 0; k < order.size(); ++k) std::copy(v.begin() + (size_t)order[k] * width, v.begin() + ((size_t)order[k] + 1) * width, out.begin() + k * width);
  v.swap(out);
}
template <class T> void unpermute_rows(std::vector<T> &v, size_t width, const std::vector<int> &order) {      // back: out[order[k]] = in[k]
  std::vector<T> out(v.size());
  for (size_t k = 0; k < order.size(); ++k) std::copy(v.begin() + k * width, v.begin() + (k + 1) * width, out.begin() + (size_t)order[k] * width);
  v.swap(out);
}

struct Shard {
  svr_ctx *e = nullptr;
  int n = 0, lo = 0, hi = 0;          // units (slices / patches): global count, this rank's range
  svr_collectives coll;
  bool given = false, on = false;     // collectives supplied / in use (world > 1, or forced for tests at world 1)

  void init(svr_ctx *engine, int n_global, int lo_, int hi_, const svr_collectives *c) {
    e = engine; n = n_global; lo = lo_; hi = hi_;
    given = c != nullptr;
    // the launcher's struct may be shorter than this build's (svr_collectives::struct_size): members it does not have are NULL / 0
    memset(&coll, 0, sizeof(coll));
    coll.world = 1;
    if (c) memcpy(&coll, c, std::min(c->struct_size, sizeof(coll)));
    coll.struct_size = sizeof(coll);
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
  // all-reduce of the pair + the update replicated on every rank (also when the engine's update is not the fused kernel, reg_mode 0, which
  // only updates whole volumes).  Identical on every rank either way; bit-equal to each other with rank-ordered collectives.
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
    int reg_mode = 1;
    (void)svr_get_option(e, "reg_mode", &reg_mode);
    if (!(slabs && coll.reduce_scatter_device && coll.allgather_device && reg_mode == 1)) {
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
