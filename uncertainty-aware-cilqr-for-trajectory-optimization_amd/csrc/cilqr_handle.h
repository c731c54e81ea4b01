// cilqr_handle.h — the handle behind the C-ABI, shared by the translation units that implement include/cilqr.h
// (cilqr_api.cpp: solver / costmap entry points; cilqr_comm.cpp: the RCCL exchange step).  Not installed.
#pragma once

#include <string>

#include "cilqr_internal.h"

struct ncclComm;  // RCCL communicator (rccl.h), only named here

namespace cilqr {
extern thread_local std::string g_last_error;
int fail(int code, const char* fmt, ...);  // records the message for cilqr_last_error() and returns `code`
int solve_batch_enqueue(cilqr_handle* h, int B, int N, int M, const double* x0, double* U, const double* poly,
                        const double* xplan_fl, const double* obs_pose, const double* obs_dim, const double* obs_weight,
                        double* X_out, double* J_out, int32_t* iters_out, int32_t* status_out, uint32_t flags);
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
  int max_batch, max_horizon, max_obstacles;
  hipStream_t stream;
  // device staging for the host-pointer entry points
  double *d_x0, *d_U, *d_poly, *d_xplan, *d_obs_pose, *d_obs_dim, *d_obs_w, *d_samp_off, *d_X, *d_J;
  int32_t *d_iters, *d_status;
  // workspace
  double* d_obs_tab;
  double* d_ws;      // workspace of the G-lanes-per-solve kernel family
  int32_t* d_redo;
  int force_g;       // 0 = automatic; else 1,2,4,8,16,32 or 64 (test hook: environment CILQR_FORCE_G at create)
  double* d_pair;
  // warp staging (grown on demand by the host-pointer warp entry point only)
  float *d_src, *d_dst, *d_bbox;
  size_t src_cap, dst_cap, bbox_cap;
  unsigned long long* d_oob;
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
