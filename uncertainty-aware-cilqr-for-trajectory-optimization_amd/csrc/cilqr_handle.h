// cilqr_handle.h — the handle behind the C-ABI, shared by the translation units that implement include/cilqr.h
// (cilqr_api.cpp: solver / costmap entry points; cilqr_comm.cpp: the RCCL exchange step).  Not installed.
#pragma once

#include <string>

#include "cilqr_internal.h"

struct ncclComm;  // RCCL communicator (rccl.h), only named here

struct cilqr_handle;

namespace cilqr {
// Where a host-buffer call's arrays lie in the handle's device arena and pinned staging buffer (cilqr_host_io.cpp); bytes.
struct IoLayout {
  size_t x0, poly, xplan, obs_w, samp_off, obs_pose, obs_dim, U, X, J, iters, status, end;
};
IoLayout io_layout(size_t B, size_t N, size_t M, bool weights, size_t n_samples);
// One host-buffer solve call (M = obstacles, or nominal obstacles of the sampled form when n_samples > 0).
struct HostBatch {
  int B, N, M, n_samples;
  const double *x0;
  double* U;
  const double *poly, *xplan_fl, *obs_pose, *obs_dim, *obs_weight, *samp_off;
  double samp_w;
  double *X_out, *J_out;
  int32_t *iters_out, *status_out;
  uint32_t flags;
};
struct PendingOut {
  bool active, packed;
  IoLayout L;
  HostBatch q;
};
int host_solve_enqueue(cilqr_handle* h, const HostBatch& q);  // copies in, kernels, copies out: all enqueued on h->stream
int host_solve_finish(cilqr_handle* h);                       // waits; unpacks the staging buffer of a small call
enum { SCR_PLAN_PATH, SCR_PLAN_IO, SCR_COUNT, SCR_CONV_IN, SCR_CONV_OUT, SCR_DEBUG, SCR_SLOTS };
int scratch_bytes(cilqr_handle* h, int slot, size_t bytes, void** out);

extern thread_local std::string g_last_error;
int fail(int code, const char* fmt, ...);  // records the message for cilqr_last_error() and returns `code`
}  // namespace cilqr

#define HIP_TRY(expr)                                                                                      \
  do {                                                                                                     \
    hipError_t e_ = (expr);                                                                                \
    if (e_ != hipSuccess) return cilqr::fail(CILQR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

struct cilqr_handle {
  cilqr_params params;
  cilqr::KParams kp;
  int device;
  int simds;         // SIMDs of the device (4 per CU): 1024 on an MI355X
  int max_batch, max_horizon, max_obstacles;
  hipStream_t stream;
  // host-buffer entry points (cilqr_host_io.cpp): device arena and pinned staging sized at create, the call in flight
  char* d_arena;
  size_t arena_cap;
  char* stage;
  size_t stage_cap;
  cilqr::PendingOut pending;
  int debug_fail_enqueue;  // test hook: the n-th host-buffer enqueue from now fails after its input copies (0: off)
  double* d_J;  // the costs of the last host-buffer call, inside the arena
  void* scratch[cilqr::SCR_SLOTS];
  size_t scratch_cap[cilqr::SCR_SLOTS];
  // workspace
  double* d_obs_tab;
  double* d_ws;      // workspace of the G-lanes-per-solve kernel family
  int32_t* d_redo;
  // schedule hint of the one-wavefront-per-solve family: passes of the previous call and the dispatch order built from them
  int32_t* d_hint_passes;
  int32_t* d_order;
  int hint_B;          // batch size the order is valid for (0: none)
  void* hint_stream;   // stream it was built on (a call on another stream does not use it)
  int hint_off;        // environment CILQR_NO_SCHEDULE_HINT at create
  int steal_off;       // environment CILQR_NO_LANE_SHARING at create: the grouped family without phase L's lane sharing (A/B, bit-equality test)
  int split_w;         // 0 = automatic; else 2 or 4 (test hook: environment CILQR_SPLIT_W at create)
  int split_off;       // environment CILQR_NO_SPLIT_KERNEL at create: sampled obstacles on one wavefront per solve (A/B, tests)
  int share_off;       // environment CILQR_NO_SHARE_KERNEL at create: static obstacles on one wavefront per solve at every batch size (A/B, tests)
  int tab_budget_kb;   // 0 = automatic (cilqr_api.cpp, lds_table_budget); else KiB (environment CILQR_LDS_TABLE_KB at create: A/B)
  int share_w;         // 0 = automatic (three wavefronts up to one solve per SIMD, two beyond); else 2 or 3 (environment CILQR_SHARE_W at create)
  int share_max;       // -1: the largest batch on the shared-phase-L kernel follows the horizon (cilqr_api.cpp, pick_share); CILQR_SHARE_MAX_B overrides
  int pair_on;         // environment CILQR_PAIR_KERNEL at create: the two-wavefront kernel for batches up to one solve per SIMD
                       // (a measured negative result, DESIGN.md §5: kept for the A/B of tools/pair_ab.py and its tests, off by default)
  int force_g;       // 0 = automatic; else 1,2,4,8,16,32 or 64 (test hook: environment CILQR_FORCE_G at create)
  double* d_pair;
  // warp staging (grown on demand by the host-pointer warp entry point only)
  float *d_src, *d_dst, *d_bbox;
  size_t src_cap, dst_cap, bbox_cap;
  unsigned long long* d_oob;
  double* d_poses;       // 8 slots x 1024 poses x 4 doubles: pose tables of cilqr_warp_costmap_batch_device calls in flight
  unsigned pose_slot;
  float* d_occ_steps;  // 8 x 128 floats: step tables of the layer -> occupancy conversion (rebuilt per call on the call's stream)
  unsigned occ_slot;
  unsigned long long* diag;  // caller-owned device buffer or null
  int32_t* passes;           // caller-owned device buffer or null (cilqr_set_pass_count_buffer)
  // uncertainty map (cilqr_set_uncertainty_map*): what the kernels read, and the device copy made from a host layer
  cilqr::UncArgs unc;
  float* d_unc_layer;
  size_t unc_layer_cap;
  // cross-rank min-cost exchange (cilqr_comm.cpp)
  ncclComm* comm;            // null: this handle is its own world
  int comm_ranks, comm_rank;
  double* d_triple;          // {J_min, local index, index offset} of this rank
  double* d_gather;          // comm_ranks triples
};
