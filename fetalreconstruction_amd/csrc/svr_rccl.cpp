// svr_rccl.cpp -- the collectives of the slice-sharded reconstruction on RCCL, bound directly (no torch in between).
//
// The reference fans every step out over `devicesToUse` inside one process, one worker thread per GPU, and adds the
// per-device volumes up on GPU 0 (reconstruction_cuda2.cu:1413-1457 sharding, :2225-2239 reduce).  Here every rank -- a
// process of its own (bench.py, one per GPU) or a thread of the command line (`-d 0 1 2 ..`, csrc/svr_cli.cpp) -- owns
// one engine context and one RCCL communicator; the exchanges of an SR iteration are
//   * one in-place ncclAllReduce(sum) of the float[2 Nv] pair addon|cmap (recon|volw in the Gaussian pass), enqueued on
//     the engine's own stream, so it is ordered with the scatter before it and the regulariser after it without a host
//     synchronisation,
//   * a handful of scalars (robust statistics, M-step: sum / min / max of <= 8 doubles) and the per-slice vectors
//     (scales, potentials: <= n_slices floats), which go through a small device scratch buffer.
// librccl is opened with dlopen at the first svr_comm_* call: a single-GPU user of libsvr_hip.so does not need it, and in
// a process that already holds an RCCL (PyTorch's) the same copy is used.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/svr_host.h"
#include "svr_rccl_abi.h"

namespace {

// the slice of rccl.h this file uses (ABI of NCCL 2.x / RCCL: enums, the opaque handles, the prototypes): csrc/svr_rccl_abi.h, checked
// against /opt/rocm/include/rccl/rccl.h at compile time by tests/rccl_abi_check.cpp
using namespace svr_rccl_abi;

struct Rccl {
  void *h = nullptr;
  GetUniqueId_fn GetUniqueId = nullptr;
  CommInitRank_fn CommInitRank = nullptr;
  CommDestroy_fn CommDestroy = nullptr;
  CommCount_fn CommCount = nullptr;
  AllReduce_fn AllReduce = nullptr;
  AllGather_fn AllGather = nullptr;
  ReduceScatter_fn ReduceScatter = nullptr;
  GroupStart_fn GroupStart = nullptr;
  GroupEnd_fn GroupEnd = nullptr;
  GetErrorString_fn GetErrorString = nullptr;
  std::string err;
};
Rccl g_rccl;
std::once_flag g_once;

bool rccl_load() {
  std::call_once(g_once, [] {
    // an RCCL the process already holds (PyTorch loads its own copy as "librccl.so") comes first: two copies of the
    // library in one process each run their own teardown over shared state
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
      g_rccl.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
      if (g_rccl.h) break;
    }
    for (const char *n : names) {
      if (g_rccl.h) break;
      g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!g_rccl.h) { g_rccl.err = std::string("dlopen(librccl): ") + dlerror(); return; }
#define SYM(field, name)                                                       \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.h, name)); \
  if (!g_rccl.field) { g_rccl.err = std::string("librccl lacks ") + name; g_rccl.h = nullptr; return; }
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
    SYM(CommCount, "ncclCommCount") SYM(AllReduce, "ncclAllReduce") SYM(AllGather, "ncclAllGather") SYM(ReduceScatter, "ncclReduceScatter")
    SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
  });
  return g_rccl.h != nullptr;
}

}  // namespace

struct svr_comm {
  int rank = 0, world = 1, device = 0;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;       // the engine's stream: collectives are ordered with its kernels
  void *d_scratch = nullptr;          // 8-byte units
  size_t scratch_units = 0;
  std::vector<int> counts;            // per-rank slice counts of the last allgather_slices (rank order = slice order)
  svr_collectives coll;
  std::string err;
};

namespace {

int cfail(svr_comm *c, const std::string &m) { c->err = m; fprintf(stderr, "svr_comm[%d]: %s\n", c->rank, m.c_str()); return 1; }
#define NCCLCHK(c, call)                                                                              \
  do { int r_ = (call); if (r_ != ncclSuccess) return cfail(c, std::string(#call) + ": " + g_rccl.GetErrorString(r_)); } while (0)
#define HIPCHK_(c, call)                                                                              \
  do { hipError_t e_ = (call); if (e_ != hipSuccess) return cfail(c, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

int scratch(svr_comm *c, size_t units) {
  if (units <= c->scratch_units) return 0;
  if (c->d_scratch) (void)hipFree(c->d_scratch);
  c->d_scratch = nullptr;
  c->scratch_units = 0;
  HIPCHK_(c, hipMalloc(&c->d_scratch, units * 8));
  c->scratch_units = units;
  return 0;
}

int cb_allreduce_volume_pair(void *user, void *device_ptr, size_t n_floats) {
  svr_comm *c = static_cast<svr_comm *>(user);
  HIPCHK_(c, hipSetDevice(c->device));
  NCCLCHK(c, g_rccl.AllReduce(device_ptr, device_ptr, n_floats, ncclFloat32, ncclSum, c->comm, c->stream));
  return 0;                            // no host synchronisation: the next engine call runs on the same stream
}

// the two collectives of the slab update (csrc/svr_slab.inc): on the engine's stream, no host synchronisation
int cb_reduce_scatter_device(void *user, const void *send, void *recv, size_t n_per_rank) {
  svr_comm *c = static_cast<svr_comm *>(user);
  HIPCHK_(c, hipSetDevice(c->device));
  NCCLCHK(c, g_rccl.ReduceScatter(send, recv, n_per_rank, ncclFloat32, ncclSum, c->comm, c->stream));
  return 0;
}
int cb_allgather_device(void *user, const void *send, void *recv, size_t n_per_rank) {
  svr_comm *c = static_cast<svr_comm *>(user);
  HIPCHK_(c, hipSetDevice(c->device));
  NCCLCHK(c, g_rccl.AllGather(send, recv, n_per_rank, ncclFloat32, c->comm, c->stream));
  return 0;
}

int cb_allreduce_host(void *user, double *data, int n, int op) {
  svr_comm *c = static_cast<svr_comm *>(user);
  if (n <= 0) return 0;
  HIPCHK_(c, hipSetDevice(c->device));
  if (scratch(c, (size_t)n)) return 1;
  HIPCHK_(c, hipMemcpyAsync(c->d_scratch, data, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
  NCCLCHK(c, g_rccl.AllReduce(c->d_scratch, c->d_scratch, (size_t)n, ncclFloat64, op == 0 ? ncclSum : (op == 1 ? ncclMin : ncclMax),
                              c->comm, c->stream));
  HIPCHK_(c, hipMemcpyAsync(data, c->d_scratch, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK_(c, hipStreamSynchronize(c->stream));
  return 0;
}

int cb_allgather_slices(void *user, const float *local, int n_local, float *global_out, int n_global) {
  svr_comm *c = static_cast<svr_comm *>(user);
  HIPCHK_(c, hipSetDevice(c->device));
  if (scratch(c, (size_t)std::max(n_global, c->world) + 8)) return 1;
  // where this rank's slices start: the counts of all ranks, gathered at the first call.  Whether they must be gathered again
  // (a caller that changed its range) is decided COLLECTIVELY -- a rank that entered the count gather alone would wait for
  // ever: every call starts with a one-word all-reduce of "my counts are stale".
  {
    int total = 0;
    for (int v : c->counts) total += v;
    const int stale = ((int)c->counts.size() != c->world || total != n_global || c->counts[c->rank] != n_local) ? 1 : 0;
    int *d = static_cast<int *>(c->d_scratch);
    int any = stale;
    if (!c->counts.empty()) {                                // (the first call gathers on every rank anyway)
      HIPCHK_(c, hipMemcpyAsync(d, &stale, sizeof(int), hipMemcpyHostToDevice, c->stream));
      NCCLCHK(c, g_rccl.AllReduce(d, d, 1, ncclInt32, ncclMax, c->comm, c->stream));
      HIPCHK_(c, hipMemcpyAsync(&any, d, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      HIPCHK_(c, hipStreamSynchronize(c->stream));
    }
    if (any) {
      HIPCHK_(c, hipMemcpyAsync(d + c->rank, &n_local, sizeof(int), hipMemcpyHostToDevice, c->stream));
      NCCLCHK(c, g_rccl.AllGather(d + c->rank, d, 1, ncclInt32, c->comm, c->stream));
      c->counts.assign(c->world, 0);
      HIPCHK_(c, hipMemcpyAsync(c->counts.data(), d, c->world * sizeof(int), hipMemcpyDeviceToHost, c->stream));
      HIPCHK_(c, hipStreamSynchronize(c->stream));
      total = 0;
      for (int v : c->counts) total += v;
      if (total != n_global) return cfail(c, "allgather_slices: the ranks' slice counts do not add up to n_global");   // (on every rank alike)
    }
  }
  int lo = 0;
  for (int r = 0; r < c->rank; ++r) lo += c->counts[r];
  // ragged all-gather of <= a few thousand floats: every rank places its piece in a zeroed vector, one all-reduce
  float *d = static_cast<float *>(c->d_scratch);
  HIPCHK_(c, hipMemsetAsync(d, 0, (size_t)n_global * sizeof(float), c->stream));
  if (n_local) HIPCHK_(c, hipMemcpyAsync(d + lo, local, (size_t)n_local * sizeof(float), hipMemcpyHostToDevice, c->stream));
  NCCLCHK(c, g_rccl.AllReduce(d, d, (size_t)n_global, ncclFloat32, ncclSum, c->comm, c->stream));
  HIPCHK_(c, hipMemcpyAsync(global_out, d, (size_t)n_global * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK_(c, hipStreamSynchronize(c->stream));
  return 0;
}

}  // namespace

extern "C" {

int svr_comm_unique_id(char id128[128]) {
  if (!id128) return 1;
  if (!rccl_load()) { fprintf(stderr, "svr_comm: %s\n", g_rccl.err.c_str()); return 1; }
  ncclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != ncclSuccess) return 1;
  memcpy(id128, id.internal, 128);
  return 0;
}

svr_comm *svr_comm_create(int rank, int world, const char id128[128], svr_ctx *engine) {
  if (!id128 || !engine || world < 1 || rank < 0 || rank >= world) return nullptr;
  if (!rccl_load()) { fprintf(stderr, "svr_comm: %s\n", g_rccl.err.c_str()); return nullptr; }
  svr_comm *c = new svr_comm();
  c->rank = rank; c->world = world;
  c->device = svr_device(engine);
  c->stream = static_cast<hipStream_t>(svr_get_stream(engine));
  if (hipSetDevice(c->device) != hipSuccess) { delete c; return nullptr; }
  ncclUniqueId id;
  memcpy(id.internal, id128, 128);
  const int r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    fprintf(stderr, "svr_comm[%d]: ncclCommInitRank: %s\n", rank, g_rccl.GetErrorString(r));
    delete c;
    return nullptr;
  }
  c->coll.struct_size = sizeof(svr_collectives);
  c->coll.user = c;
  c->coll.rank = rank;
  c->coll.world = world;
  c->coll.allreduce_volume_pair = cb_allreduce_volume_pair;
  c->coll.allreduce_host = cb_allreduce_host;
  c->coll.allgather_slices = cb_allgather_slices;
  c->coll.reduce_scatter_device = cb_reduce_scatter_device;
  c->coll.allgather_device = cb_allgather_device;
  c->coll.on_engine_stream = 1;
  return c;
}

/* the same communicator for another engine context on the SAME device: its collectives run on that engine's stream from now on
 * (bench.py measures a second workload in the same launch, on the same ranks and the same communicator) */
int svr_comm_rebind(svr_comm *c, svr_ctx *engine) {
  if (!c || !engine) return 1;
  if (svr_device(engine) != c->device) return cfail(c, "svr_comm_rebind: the engine lives on another device");
  if (hipSetDevice(c->device) != hipSuccess) return cfail(c, "svr_comm_rebind: hipSetDevice");
  // (the previous engine may be gone already, and with it its stream: nothing of the communicator's is pending there -- every host-vector
  // collective waits for its own result, and the volume collectives are followed by a wait of the host objects before they return)
  c->stream = static_cast<hipStream_t>(svr_get_stream(engine));
  c->counts.clear();
  return 0;
}

const svr_collectives *svr_comm_collectives(svr_comm *c) { return c ? &c->coll : nullptr; }

int svr_comm_world(svr_comm *c) {        // the size RCCL itself reports for the communicator
  int n = 0;
  if (!c || g_rccl.CommCount(c->comm, &n) != ncclSuccess) return -1;
  return n;
}

/* host-vector reduction for the launcher (timing: max over ranks; counters: sum); op: 0 sum, 1 min, 2 max */
int svr_comm_allreduce_host(svr_comm *c, double *data, int n, int op) { return c ? cb_allreduce_host(c, data, n, op) : 1; }

const char *svr_comm_last_error(const svr_comm *c) { return c ? c->err.c_str() : "null communicator"; }

void svr_comm_destroy(svr_comm *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();         // (not the engine's stream: the engine -- and its stream -- may be gone before its communicator)
  if (c->comm) (void)g_rccl.CommDestroy(c->comm);
  if (c->d_scratch) (void)hipFree(c->d_scratch);
  delete c;
}

}  // extern "C"

// ---- rank groups inside one process: the command line's `-d 0 1 2 ..` --------------------------------------------------
// One thread per rank, one engine context per rank.  Distinct devices: every rank makes an RCCL communicator from one
// shared ncclUniqueId (svr_group_join blocks until all have joined).  The same device named more than once (RCCL refuses
// two ranks on one GPU): the ranks exchange through host memory under a thread barrier -- a test mode for one-GPU boxes,
// same sharding, same call sequence, no claim on speed.
namespace {
struct Barrier {
  std::mutex m;
  std::condition_variable cv;
  int n, waiting = 0;
  unsigned long gen = 0;
  explicit Barrier(int n_) : n(n_) {}
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    const unsigned long g = gen;
    if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g; });
  }
};
}  // namespace

struct svr_group {
  int world = 1;
  bool rccl = true;
  char id[128];
  std::vector<svr_comm *> comms;                 // rccl mode
  // shared-memory mode
  Barrier bar;
  std::vector<std::vector<float>> fstage;        // per rank
  std::vector<std::vector<double>> dstage;
  std::vector<int> counts;
  std::vector<float> fsum;
  std::vector<double> dsum;
  struct Member { svr_group *g; int rank; svr_ctx *engine; svr_collectives coll; };
  std::vector<Member> members;
  // a rank that fails still reaches every barrier of the exchange and raises this flag; all ranks return it afterwards
  // (an early return left the peers waiting at the barrier for ever).  Sticky: a group that failed once stays failed.
  std::atomic<int> failed{0};
  explicit svr_group(int w) : world(w), bar(w) {}
};

namespace {

int g_pair(void *user, void *device_ptr, size_t n) {
  auto *m = static_cast<svr_group::Member *>(user);
  svr_group *g = m->g;
  hipStream_t st = static_cast<hipStream_t>(svr_get_stream(m->engine));
  std::vector<float> &mine = g->fstage[m->rank];
  mine.resize(n);
  if (hipSetDevice(svr_device(m->engine)) != hipSuccess ||
      hipMemcpyAsync(mine.data(), device_ptr, n * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    g->failed.store(1);
  g->bar.wait();
  if (m->rank == 0 && !g->failed.load()) {
    g->fsum.assign(n, 0.0f);
    for (int r = 0; r < g->world; ++r)                     // rank order: the same sum on every run
      for (size_t i = 0; i < n && i < g->fstage[r].size(); ++i) g->fsum[i] += g->fstage[r][i];
  }
  g->bar.wait();
  if (!g->failed.load() &&
      (hipMemcpyAsync(device_ptr, g->fsum.data(), n * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess ||
       hipStreamSynchronize(st) != hipSuccess))
    g->failed.store(1);
  g->bar.wait();                                           // fsum is free again; the flag is final for this exchange
  return g->failed.load();
}
// reduce-scatter / all-gather through host memory (the same device named more than once): sums in rank order like g_pair, so the
// slab update gives the replicated update's bits
int g_rs(void *user, const void *send, void *recv, size_t n) {
  auto *m = static_cast<svr_group::Member *>(user);
  svr_group *g = m->g;
  const int W = g->world;
  hipStream_t st = static_cast<hipStream_t>(svr_get_stream(m->engine));
  std::vector<float> &mine = g->fstage[m->rank];
  mine.resize(n * W);
  if (hipSetDevice(svr_device(m->engine)) != hipSuccess ||
      hipMemcpyAsync(mine.data(), send, n * W * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    g->failed.store(1);
  g->bar.wait();
  std::vector<float> sum(n, 0.0f);
  if (!g->failed.load()) {
    for (int r = 0; r < W; ++r) {
      if (g->fstage[r].size() != n * W) { g->failed.store(1); break; }
      const float *src = g->fstage[r].data() + (size_t)m->rank * n;
      for (size_t i = 0; i < n; ++i) sum[i] += src[i];
    }
  }
  if (!g->failed.load() &&
      (hipMemcpyAsync(recv, sum.data(), n * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess))
    g->failed.store(1);
  g->bar.wait();                                           // every rank has read the stages
  return g->failed.load();
}
int g_ag(void *user, const void *send, void *recv, size_t n) {
  auto *m = static_cast<svr_group::Member *>(user);
  svr_group *g = m->g;
  const int W = g->world;
  hipStream_t st = static_cast<hipStream_t>(svr_get_stream(m->engine));
  std::vector<float> &mine = g->fstage[m->rank];
  mine.resize(n);
  if (hipSetDevice(svr_device(m->engine)) != hipSuccess ||
      hipMemcpyAsync(mine.data(), send, n * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    g->failed.store(1);
  g->bar.wait();
  if (!g->failed.load()) {
    for (int r = 0; r < W; ++r) {
      if (g->fstage[r].size() != n) { g->failed.store(1); break; }
      if (hipMemcpyAsync(static_cast<float *>(recv) + (size_t)r * n, g->fstage[r].data(), n * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess) { g->failed.store(1); break; }
    }
    if (hipStreamSynchronize(st) != hipSuccess) g->failed.store(1);
  }
  g->bar.wait();
  return g->failed.load();
}
int g_rs_rccl(void *user, const void *send, void *recv, size_t n) {
  auto *m = static_cast<svr_group::Member *>(user);
  return cb_reduce_scatter_device(m->g->comms[m->rank], send, recv, n);
}
int g_ag_rccl(void *user, const void *send, void *recv, size_t n) {
  auto *m = static_cast<svr_group::Member *>(user);
  return cb_allgather_device(m->g->comms[m->rank], send, recv, n);
}
int g_pair_rccl(void *user, void *device_ptr, size_t n) {
  auto *m = static_cast<svr_group::Member *>(user);
  return cb_allreduce_volume_pair(m->g->comms[m->rank], device_ptr, n);
}
int g_host(void *user, double *data, int n, int op) {
  auto *m = static_cast<svr_group::Member *>(user);
  svr_group *g = m->g;
  g->dstage[m->rank].assign(data, data + n);
  g->bar.wait();
  if (m->rank == 0)
    for (int r = 0; r < g->world; ++r)
      if ((int)g->dstage[r].size() != n) g->failed.store(1);      // the ranks are not in the same exchange
  if (m->rank == 0 && !g->failed.load()) {
    g->dsum = g->dstage[0];
    for (int r = 1; r < g->world; ++r)
      for (int i = 0; i < n; ++i) {
        const double v = g->dstage[r][i];
        g->dsum[i] = op == 0 ? g->dsum[i] + v : (op == 1 ? std::min(g->dsum[i], v) : std::max(g->dsum[i], v));
      }
  }
  g->bar.wait();
  if (!g->failed.load()) std::copy(g->dsum.begin(), g->dsum.begin() + n, data);
  g->bar.wait();
  return g->failed.load();
}
int g_gather(void *user, const float *local, int n_local, float *global_out, int n_global) {
  auto *m = static_cast<svr_group::Member *>(user);
  svr_group *g = m->g;
  g->fstage[m->rank].assign(local, local + n_local);
  g->bar.wait();
  int o = 0;
  for (int r = 0; r < g->world; ++r) {
    if (o + (int)g->fstage[r].size() > n_global) { g->failed.store(1); break; }
    std::copy(g->fstage[r].begin(), g->fstage[r].end(), global_out + o);
    o += (int)g->fstage[r].size();
  }
  if (o != n_global) g->failed.store(1);
  g->bar.wait();
  return g->failed.load();
}

}  // namespace

extern "C" {

svr_group *svr_group_create(int world, const int *devices) {
  if (world < 1 || !devices) return nullptr;
  svr_group *g = new svr_group(world);
  std::vector<int> d(devices, devices + world);
  std::sort(d.begin(), d.end());
  g->rccl = std::adjacent_find(d.begin(), d.end()) == d.end();
  if (g->rccl && world > 1 && svr_comm_unique_id(g->id)) { delete g; return nullptr; }
  g->comms.assign(world, nullptr);
  g->fstage.resize(world);
  g->dstage.resize(world);
  g->members.resize(world);
  return g;
}

int svr_group_uses_rccl(const svr_group *g) { return g && g->rccl && g->world > 1; }

/* called by every rank from its own thread; returns the rank's collectives (NULL on failure; NULL also for world 1) */
const svr_collectives *svr_group_join(svr_group *g, int rank, svr_ctx *engine) {
  if (!g || rank < 0 || rank >= g->world || !engine || g->world == 1) return nullptr;
  svr_group::Member &m = g->members[rank];
  m.g = g; m.rank = rank; m.engine = engine;
  m.coll.struct_size = sizeof(svr_collectives);
  m.coll.user = &m; m.coll.rank = rank; m.coll.world = g->world;
  if (g->rccl) {
    // the volume pairs over RCCL on the engine's stream; the small host vectors through the memory the rank threads share
    // (a thread barrier instead of a stream synchronisation plus a collective launch per exchange).
    // What can fail on ONE rank before ncclCommInitRank (the library, the device) is checked first and agreed on under the
    // barrier: a rank that gave up alone would leave its peers inside ncclCommInitRank for ever.
    if (!rccl_load() || hipSetDevice(svr_device(engine)) != hipSuccess) g->failed.store(1);
    g->bar.wait();
    if (g->failed.load()) return nullptr;
    g->comms[rank] = svr_comm_create(rank, g->world, g->id, engine);
    if (!g->comms[rank]) g->failed.store(1);
    g->bar.wait();
    if (g->failed.load()) return nullptr;
    m.coll.allreduce_volume_pair = g_pair_rccl;
    m.coll.allreduce_host = g_host;
    m.coll.allgather_slices = g_gather;
    m.coll.reduce_scatter_device = g_rs_rccl;
    m.coll.allgather_device = g_ag_rccl;
    m.coll.on_engine_stream = 1;
    return &m.coll;
  }
  m.coll.allreduce_volume_pair = g_pair;
  m.coll.allreduce_host = g_host;
  m.coll.allgather_slices = g_gather;
  m.coll.reduce_scatter_device = g_rs;
  m.coll.allgather_device = g_ag;
  m.coll.on_engine_stream = 1;          // (they copy through the engine's stream and wait for it themselves)
  return &m.coll;
}

void svr_group_destroy(svr_group *g) {
  if (!g) return;
  for (svr_comm *c : g->comms) svr_comm_destroy(c);
  delete g;
}

}  // extern "C"
