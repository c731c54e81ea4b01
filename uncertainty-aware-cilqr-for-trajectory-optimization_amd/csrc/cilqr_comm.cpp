// cilqr_comm.cpp — the one exchange step of a scene-sharded batch behind the C-ABI (SURVEY §8b "Entry point", §8e):
// per-rank argmin → ONE ncclAllGather of a 24-byte record per rank over RCCL/xGMI → lexicographic minimum on the device.
// The message is bytes, so the step is latency-bound: ring vs tree and the per-link xGMI bandwidth do not matter.
// Two host models over the same step:
//   * one process per GPU (torch.distributed.run, MPI, …): cilqr_comm_unique_id on one rank, the 128 bytes carried to the
//     others by whatever channel the host has, cilqr_comm_init_rank on every rank's handle, cilqr_argmin_global_device;
//   * one process driving every GPU of the node (the north_star's C++ host): cilqr_create_multi / cilqr_multi_solve_batch,
//     one handle, stream and communicator per device (ncclCommInitAll), batch sharded contiguously by scene.
// The reference has no collective of any kind; nothing here restates reference code.
#include <rccl/rccl.h>

#include <math.h>
#include <string.h>

#include <new>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "cilqr_handle.h"

using cilqr::fail;

#define NCCL_TRY(expr)                                                                                      \
  do {                                                                                                      \
    ncclResult_t r_ = (expr);                                                                               \
    if (r_ != ncclSuccess) return fail(CILQR_ERR_COMM, "%s failed: %s", #expr, ncclGetErrorString(r_));     \
  } while (0)

static_assert(CILQR_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "cilqr.h carries RCCL's unique id verbatim");

struct cilqr_multi {
  std::vector<cilqr_handle*> h;
  std::vector<int> devices;
  int max_batch_per_device;
  bool rccl;  // false: the device list names one device several times (shards share a GPU): RCCL cannot span a device twice, the
              // 24-byte records are then gathered by device copies ordered through events (cilqr_create_multi)
  std::vector<hipEvent_t> ev;  // !rccl: one per shard, "this shard's record is written"
};

namespace {

int set_gather(cilqr_handle* h, int n_ranks) {
  HIP_TRY(hipSetDevice(h->device));
  if (h->d_gather) (void)hipFree(h->d_gather);
  h->d_gather = nullptr;
  HIP_TRY(hipMalloc((void**)&h->d_gather, sizeof(double) * 3 * (size_t)n_ranks));
  return CILQR_OK;
}

}  // namespace

extern "C" {

int cilqr_shard_range(int B, int n_shards, int shard, int* first, int* count) {
  if (B < 0 || n_shards < 1 || shard < 0 || shard >= n_shards || !first || !count) return fail(CILQR_ERR_ARG, "cilqr_shard_range: bad argument");
  // contiguous and balanced: the first B mod n shards own one solve more (a launch lasts as long as its longest solve whatever
  // the shard's size, so no device should carry two solves more than another)
  const int base = B / n_shards, rem = B % n_shards;
  *first = shard * base + (shard < rem ? shard : rem);
  *count = base + (shard < rem ? 1 : 0);
  return CILQR_OK;
}

int cilqr_comm_unique_id(void* id_bytes) {
  if (!id_bytes) return fail(CILQR_ERR_ARG, "cilqr_comm_unique_id: null argument");
  ncclUniqueId id;
  NCCL_TRY(ncclGetUniqueId(&id));
  memcpy(id_bytes, id.internal, NCCL_UNIQUE_ID_BYTES);
  return CILQR_OK;
}

int cilqr_comm_init_rank(cilqr_handle* h, int n_ranks, int rank, const void* id_bytes) {
  if (!h || !id_bytes || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(CILQR_ERR_ARG, "cilqr_comm_init_rank: bad argument");
  if (h->comm) return fail(CILQR_ERR_ARG, "cilqr_comm_init_rank: the handle already has a communicator");
  HIP_TRY(hipSetDevice(h->device));
  ncclUniqueId id;
  memcpy(id.internal, id_bytes, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t c = nullptr;
  NCCL_TRY(ncclCommInitRank(&c, n_ranks, id, rank));
  int rc = set_gather(h, n_ranks);
  if (rc) { (void)ncclCommDestroy(c); return rc; }
  h->comm = c;
  h->comm_ranks = n_ranks;
  h->comm_rank = rank;
  return CILQR_OK;
}

int cilqr_comm_destroy(cilqr_handle* h) {
  if (!h) return fail(CILQR_ERR_ARG, "null handle");
  if (h->comm) {
    (void)hipSetDevice(h->device);
    ncclResult_t r = ncclCommDestroy(h->comm);
    h->comm = nullptr;
    h->comm_ranks = 1;
    h->comm_rank = 0;
    if (r != ncclSuccess) return fail(CILQR_ERR_COMM, "ncclCommDestroy failed: %s", ncclGetErrorString(r));
  }
  return CILQR_OK;
}

int cilqr_comm_size(const cilqr_handle* h) { return h ? (h->comm ? h->comm_ranks : 1) : 0; }

int cilqr_argmin_global_device(cilqr_handle* h, void* stream, int B, const double* J, int64_t index_offset, double* out_pair) {
  if (!h || !out_pair || B < 0 || (B > 0 && !J) || index_offset < 0) return fail(CILQR_ERR_ARG, "cilqr_argmin_global_device: bad argument");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  if (!h->comm && B > 0) {  // a lone rank: the local pick with its offset, one launch
    HIP_TRY(cilqr::launch_argmin(J, B, out_pair, nullptr, 0.0, s, (double)index_offset));
    return CILQR_OK;
  }
  if (B > 0) {
    HIP_TRY(cilqr::launch_argmin(J, B, nullptr, h->d_triple, (double)index_offset, s));
  } else {  // a rank without scenes takes part in the exchange with "no finite cost"
    const double none[3] = {HUGE_VAL, -1.0, 0.0};
    HIP_TRY(hipMemcpyAsync(h->d_triple, none, sizeof(none), hipMemcpyHostToDevice, s));
  }
  const double* gathered = h->d_triple;
  int n = 1;
  if (h->comm) {
    NCCL_TRY(ncclAllGather(h->d_triple, h->d_gather, 3, ncclDouble, h->comm, s));
    gathered = h->d_gather;
    n = h->comm_ranks;
  }
  HIP_TRY(cilqr::launch_select(gathered, n, out_pair, s));
  return CILQR_OK;
}

int cilqr_debug_select(cilqr_handle* h, int n, const double* triples, double* out_pair) {
  if (!h || n < 1 || !triples || !out_pair) return fail(CILQR_ERR_ARG, "cilqr_debug_select: bad argument");
  HIP_TRY(hipSetDevice(h->device));
  void* v = nullptr;
  int rc = cilqr::scratch_bytes(h, cilqr::SCR_DEBUG, sizeof(double) * 3 * (size_t)n, &v);
  if (rc) return rc;
  double* d = (double*)v;
  HIP_TRY(hipMemcpyAsync(d, triples, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(cilqr::launch_select(d, n, h->d_pair, h->stream));
  HIP_TRY(hipMemcpyAsync(out_pair, h->d_pair, sizeof(double) * 2, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CILQR_OK;
}

// ---- one process, every GPU of the node ----------------------------------------------------------------------------------
int cilqr_create_multi(const cilqr_params* p, int max_batch_per_device, int max_horizon, int max_obstacles, int n_devices,
                       const int* devices, cilqr_multi** out) {
  if (!p || !out || n_devices < 1) return fail(CILQR_ERR_ARG, "cilqr_create_multi: bad argument");
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(CILQR_ERR_NO_DEVICE, "no HIP device available; this library has no CPU path");
  if (!devices && n_devices > count) return fail(CILQR_ERR_ARG, "cilqr_create_multi: %d devices requested, %d present", n_devices, count);
  cilqr_multi* m = new (std::nothrow) cilqr_multi();
  if (!m) return fail(CILQR_ERR_ARG, "out of host memory");
  m->max_batch_per_device = max_batch_per_device;
  for (int i = 0; i < n_devices; ++i) m->devices.push_back(devices ? devices[i] : i);
  m->rccl = std::set<int>(m->devices.begin(), m->devices.end()).size() == m->devices.size();
  for (int i = 0; i < n_devices; ++i) {
    cilqr_handle* h = nullptr;
    int rc = cilqr_create(p, max_batch_per_device, max_horizon, max_obstacles, m->devices[i], &h);
    if (rc) { cilqr_multi_destroy(m); return rc; }
    m->h.push_back(h);
  }
  if (m->rccl) {
    std::vector<ncclComm_t> comms(n_devices, nullptr);
    ncclResult_t r = ncclCommInitAll(comms.data(), n_devices, m->devices.data());
    if (r != ncclSuccess) {
      cilqr_multi_destroy(m);
      return fail(CILQR_ERR_COMM, "ncclCommInitAll failed: %s", ncclGetErrorString(r));
    }
    for (int i = 0; i < n_devices; ++i) {
      m->h[i]->comm = comms[i];
      m->h[i]->comm_ranks = n_devices;
      m->h[i]->comm_rank = i;
    }
  }
  for (int i = 0; i < n_devices; ++i) {
    int rc = set_gather(m->h[i], n_devices);
    if (rc) { cilqr_multi_destroy(m); return rc; }
  }
  if (!m->rccl) {  // several shards on one device (a one-GPU box rehearsing the n-shard path): records gathered by device copies
    for (int i = 0; i < n_devices; ++i) {
      hipEvent_t e = nullptr;
      if (hipSetDevice(m->devices[i]) != hipSuccess || hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        cilqr_multi_destroy(m);
        return fail(CILQR_ERR_HIP, "cilqr_create_multi: hipEventCreate failed");
      }
      m->ev.push_back(e);
    }
  }
  *out = m;
  return CILQR_OK;
}

int cilqr_multi_destroy(cilqr_multi* m) {
  if (!m) return CILQR_OK;
  for (hipEvent_t e : m->ev) (void)hipEventDestroy(e);
  for (cilqr_handle* h : m->h) cilqr_destroy(h);  // destroys each handle's communicator too
  delete m;
  return CILQR_OK;
}

int cilqr_multi_device_count(const cilqr_multi* m) { return m ? (int)m->h.size() : 0; }

cilqr_handle* cilqr_multi_handle(cilqr_multi* m, int i) { return (m && i >= 0 && i < (int)m->h.size()) ? m->h[i] : nullptr; }

int cilqr_multi_uses_rccl(const cilqr_multi* m) { return m && m->rccl ? 1 : 0; }

}  // extern "C"

namespace {

// The one way out of cilqr_multi_solve_batch after anything was enqueued: every stream drained (asynchronous copies into the
// caller's memory may be in flight on any of them), every handle's in-flight mark cleared, then the first error reported.
int multi_abort(cilqr_multi* m, int rc, const std::string& msg) {
  for (cilqr_handle* h : m->h) {
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    h->pending.active = false;
  }
  return fail(rc, "%s", msg.c_str());
}

}  // namespace

extern "C" {

int cilqr_multi_solve_batch(cilqr_multi* m, int B, int N, int M, const double* x0, double* U, const double* poly,
                            const double* xplan_fl, const double* obs_pose, const double* obs_dim, const double* obs_weight,
                            double* X_out, double* J_out, int32_t* iters_out, int32_t* status_out, uint32_t flags,
                            int64_t* best_index, double* best_J) {
  if (!m || m->h.empty()) return fail(CILQR_ERR_ARG, "cilqr_multi_solve_batch: null handle");
  const int n = (int)m->h.size();
  if (B < 0 || (long)B > (long)n * m->max_batch_per_device)
    return fail(CILQR_ERR_ARG, "B=%d outside [0,%ld]", B, (long)n * m->max_batch_per_device);
  if (N < 1 || N > m->h[0]->max_horizon || M < 0 || M > m->h[0]->max_obstacles) return fail(CILQR_ERR_ARG, "N or M outside the sizes given at create");
  if (B > 0 && (!x0 || !U || !poly || !xplan_fl || !X_out || (M > 0 && (!obs_pose || !obs_dim))))
    return fail(CILQR_ERR_ARG, "cilqr_multi_solve_batch: null required pointer");
  std::vector<int> first(n), cnt(n);
  for (int d = 0; d < n; ++d) {
    int rc = cilqr_shard_range(B, n, d, &first[d], &cnt[d]);
    if (rc) return rc;
    if (cnt[d] > m->max_batch_per_device) return fail(CILQR_ERR_ARG, "shard of %d solves exceeds max_batch_per_device=%d", cnt[d], m->max_batch_per_device);
  }
  const size_t sN = N, sM = M;
  // One host thread per device for the enqueue: from pageable caller memory hipMemcpyAsync is a synchronous copy, so a single
  // thread walking the devices would hold device d+1's inputs back until device d's kernel has finished and its results are
  // home.  Each thread enqueues its shard's copies in, kernels, copies out and the shard's argmin on its device's stream.
  std::vector<int> rcs(n, CILQR_OK);
  std::vector<std::string> msgs(n);
  auto work = [&](int d) {
    cilqr_handle* h = m->h[d];
    auto body = [&]() -> int {
      HIP_TRY(hipSetDevice(h->device));
      if (cnt[d] > 0) {
        const size_t f = first[d];
        cilqr::HostBatch q{cnt[d], N, M, 0, x0 + f * 4, U + f * 2 * sN, poly + f * CILQR_POLY_COEFFS, xplan_fl + f * 2,
                           obs_pose ? obs_pose + f * sM * sN * 4 : nullptr, obs_dim ? obs_dim + f * sM * sN * 2 : nullptr,
                           obs_weight ? obs_weight + f * sM : nullptr, nullptr, 0.0, X_out + f * 4 * (sN + 1), J_out ? J_out + f : nullptr,
                           iters_out ? iters_out + f : nullptr, status_out ? status_out + f : nullptr, flags};
        int rc = cilqr::host_solve_enqueue(h, q);
        if (rc) return rc;
        HIP_TRY(cilqr::launch_argmin(h->d_J, cnt[d], nullptr, h->d_triple, (double)first[d], h->stream));
      } else {  // a device without scenes takes part in the exchange with "no finite cost"
        static const double none[3] = {HUGE_VAL, -1.0, 0.0};
        HIP_TRY(hipMemcpyAsync(h->d_triple, none, sizeof(none), hipMemcpyHostToDevice, h->stream));
      }
      if (!m->rccl) HIP_TRY(hipEventRecord(m->ev[d], h->stream));
      return CILQR_OK;
    };
    rcs[d] = body();
    if (rcs[d]) msgs[d] = cilqr::g_last_error;  // (thread-local: carried back to the caller's thread)
  };
  if (n == 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    for (int d = 0; d < n; ++d) pool.emplace_back(work, d);
    for (std::thread& t : pool) t.join();
  }
  for (int d = 0; d < n; ++d)
    if (rcs[d]) return multi_abort(m, rcs[d], msgs[d]);
  // the exchange step: one grouped all-gather of the 24-byte records, the pick on device 0
  if (m->rccl) {
    ncclResult_t r = ncclGroupStart();
    for (int d = 0; d < n && r == ncclSuccess; ++d) {
      cilqr_handle* h = m->h[d];
      r = ncclAllGather(h->d_triple, h->d_gather, 3, ncclDouble, h->comm, h->stream);
    }
    const ncclResult_t r_end = ncclGroupEnd();  // (closed on every path)
    if (r == ncclSuccess) r = r_end;
    if (r != ncclSuccess) return multi_abort(m, CILQR_ERR_COMM, std::string("RCCL all-gather failed: ") + ncclGetErrorString(r));
  } else {
    cilqr_handle* h0 = m->h[0];
    hipError_t e = hipSetDevice(h0->device);
    for (int d = 0; d < n && e == hipSuccess; ++d) {
      e = hipStreamWaitEvent(h0->stream, m->ev[d], 0);
      if (e == hipSuccess) e = hipMemcpyAsync(h0->d_gather + 3 * (size_t)d, m->h[d]->d_triple, 3 * sizeof(double), hipMemcpyDeviceToDevice, h0->stream);
    }
    if (e != hipSuccess) return multi_abort(m, CILQR_ERR_HIP, std::string("record gather failed: ") + hipGetErrorString(e));
  }
  double pair[2] = {HUGE_VAL, -1.0};
  {
    cilqr_handle* h = m->h[0];
    hipError_t e = hipSetDevice(h->device);
    if (e == hipSuccess) e = cilqr::launch_select(h->d_gather, n, h->d_pair, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(pair, h->d_pair, sizeof(pair), hipMemcpyDeviceToHost, h->stream);
    if (e != hipSuccess) return multi_abort(m, CILQR_ERR_HIP, std::string("min-cost pick failed: ") + hipGetErrorString(e));
  }
  for (int d = 0; d < n; ++d) {
    int rc = cilqr::host_solve_finish(m->h[d]);  // waits for the device's stream; unpacks a small shard's staging buffer
    if (rc == CILQR_OK && hipStreamSynchronize(m->h[d]->stream) != hipSuccess) rc = fail(CILQR_ERR_HIP, "hipStreamSynchronize failed");
    if (rc) return multi_abort(m, rc, cilqr::g_last_error);
  }
  if (best_J) *best_J = pair[0];
  if (best_index) *best_index = (int64_t)pair[1];
  return CILQR_OK;
}

}  // extern "C"
