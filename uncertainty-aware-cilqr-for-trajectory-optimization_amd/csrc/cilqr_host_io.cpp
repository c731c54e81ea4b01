// cilqr_host_io.cpp — the host-buffer side of the solve entry points (cilqr_solve_batch, cilqr_solve_batch_sampled,
// cilqr_multi_solve_batch): how a batch given in host memory reaches the kernels and comes back.  SURVEY §8(b) "Ownership":
// the caller owns every host buffer, the library owns device buffers sized at create and allocates nothing per call.
//
// One device arena per handle holds a call's arrays packed in a fixed order,
//     [ x0 | poly | xplan | obs_weight | sample_offset | obs_pose | obs_dim | U ][ X | J | iters | status ]
// inputs first, the in/out U at the seam, outputs last: the inputs are one contiguous prefix and what returns (U … status) one
// contiguous suffix.
//   * small calls (everything ≤ the handle's pinned staging buffer: the drop-in B = 1 tick, run_candidates): the caller's arrays
//     are packed into pinned memory and travel as ONE host→device copy; the results return as ONE device→host copy and are
//     unpacked — 2 DMA transfers per tick instead of 7 + 5, each of which costs ≈10 µs of latency whatever its size;
//   * large calls: each array is copied straight from / to the caller's memory into its place in the arena; from memory of
//     cilqr_host_alloc (pinned) these are true asynchronous DMA transfers, from pageable memory the runtime stages them.
#include <string.h>

#include "cilqr_handle.h"

using cilqr::fail;

namespace cilqr {

namespace {
size_t up16(size_t v) { return (v + 15) & ~(size_t)15; }
}  // namespace

IoLayout io_layout(size_t B, size_t N, size_t M, bool weights, size_t n_samples) {
  IoLayout L;
  size_t o = 0;
  L.x0 = o; o = up16(o + B * 4 * sizeof(double));
  L.poly = o; o = up16(o + B * CILQR_POLY_COEFFS * sizeof(double));
  L.xplan = o; o = up16(o + B * 2 * sizeof(double));
  L.obs_w = o; o = up16(o + (weights ? B * M * sizeof(double) : 0));
  L.samp_off = o; o = up16(o + B * M * n_samples * 3 * sizeof(double));
  L.obs_pose = o; o = up16(o + B * M * N * 4 * sizeof(double));
  L.obs_dim = o; o = up16(o + B * M * N * 2 * sizeof(double));
  L.U = o; o = up16(o + B * 2 * N * sizeof(double));
  L.X = o; o = up16(o + B * 4 * (N + 1) * sizeof(double));
  L.J = o; o = up16(o + B * sizeof(double));
  L.iters = o; o = up16(o + B * sizeof(int32_t));
  L.status = o; o = up16(o + B * sizeof(int32_t));
  L.end = o;
  return L;
}

namespace {
int enqueue_steps(cilqr_handle* h, const HostBatch& q);
}

// Enqueues a call; on any failure nothing is left behind: the stream is drained first — asynchronous copies to or from the
// CALLER's memory may already be in flight when a later step fails — and the handle is free for the next call.
int host_solve_enqueue(cilqr_handle* h, const HostBatch& q) {
  if (h->pending.active) return fail(CILQR_ERR_ARG, "a host-buffer solve is already in flight on this handle");
  const int rc = enqueue_steps(h, q);
  if (rc != CILQR_OK) {
    const std::string msg = g_last_error;
    (void)hipStreamSynchronize(h->stream);
    h->pending.active = false;
    g_last_error = msg;
  }
  return rc;
}

namespace {
int enqueue_steps(cilqr_handle* h, const HostBatch& q) {
  const bool sampled = q.n_samples > 0;
  const size_t B = q.B, N = q.N, M = q.M;
  const bool have_w = !sampled && q.obs_weight && M > 0;
  const IoLayout L = io_layout(B, N, M, have_w, sampled ? q.n_samples : 0);
  if (L.end > h->arena_cap) return fail(CILQR_ERR_ARG, "batch does not fit the device buffers reserved at create");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  char* dv = h->d_arena;
  const bool packed = L.end <= h->stage_cap;
  const size_t n_x0 = B * 4 * sizeof(double), n_poly = B * CILQR_POLY_COEFFS * sizeof(double), n_fl = B * 2 * sizeof(double);
  const size_t n_w = have_w ? B * M * sizeof(double) : 0, n_off = B * M * (sampled ? (size_t)q.n_samples : 0) * 3 * sizeof(double);
  const size_t n_pose = B * M * N * 4 * sizeof(double), n_dim = B * M * N * 2 * sizeof(double), n_U = B * 2 * N * sizeof(double);
  if (packed) {
    char* st = h->stage;
    memcpy(st + L.x0, q.x0, n_x0);
    memcpy(st + L.poly, q.poly, n_poly);
    memcpy(st + L.xplan, q.xplan_fl, n_fl);
    if (n_w) memcpy(st + L.obs_w, q.obs_weight, n_w);
    if (n_off) memcpy(st + L.samp_off, q.samp_off, n_off);
    if (M > 0) {
      memcpy(st + L.obs_pose, q.obs_pose, n_pose);
      memcpy(st + L.obs_dim, q.obs_dim, n_dim);
    }
    memcpy(st + L.U, q.U, n_U);
    HIP_TRY(hipMemcpyAsync(dv, st, L.X, hipMemcpyHostToDevice, s));  // the whole input prefix, U included
  } else {
    HIP_TRY(hipMemcpyAsync(dv + L.x0, q.x0, n_x0, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(dv + L.poly, q.poly, n_poly, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(dv + L.xplan, q.xplan_fl, n_fl, hipMemcpyHostToDevice, s));
    if (n_w) HIP_TRY(hipMemcpyAsync(dv + L.obs_w, q.obs_weight, n_w, hipMemcpyHostToDevice, s));
    if (n_off) HIP_TRY(hipMemcpyAsync(dv + L.samp_off, q.samp_off, n_off, hipMemcpyHostToDevice, s));
    if (M > 0) {
      HIP_TRY(hipMemcpyAsync(dv + L.obs_pose, q.obs_pose, n_pose, hipMemcpyHostToDevice, s));
      HIP_TRY(hipMemcpyAsync(dv + L.obs_dim, q.obs_dim, n_dim, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(hipMemcpyAsync(dv + L.U, q.U, n_U, hipMemcpyHostToDevice, s));
  }
  if (h->debug_fail_enqueue > 0 && --h->debug_fail_enqueue == 0)  // test hook (cilqr_debug_fail_enqueue): fail with the copies in flight
    return fail(CILQR_ERR_HIP, "forced failure after the input copies were enqueued (cilqr_debug_fail_enqueue)");
  double* dU = (double*)(dv + L.U);
  double* dX = (double*)(dv + L.X);
  double* dJ = (double*)(dv + L.J);
  int32_t* dI = (int32_t*)(dv + L.iters);
  int32_t* dS = (int32_t*)(dv + L.status);
  int rc;
  if (sampled)
    rc = cilqr_solve_batch_sampled_device(h, s, q.B, q.N, q.M, q.n_samples, (const double*)(dv + L.x0), dU, (const double*)(dv + L.poly),
                                          (const double*)(dv + L.xplan), (const double*)(dv + L.obs_pose), (const double*)(dv + L.obs_dim),
                                          (const double*)(dv + L.samp_off), q.samp_w, dX, dJ, dI, dS, q.flags);
  else
    rc = cilqr_solve_batch_device(h, s, q.B, q.N, q.M, (const double*)(dv + L.x0), dU, (const double*)(dv + L.poly),
                                  (const double*)(dv + L.xplan), M > 0 ? (const double*)(dv + L.obs_pose) : nullptr,
                                  M > 0 ? (const double*)(dv + L.obs_dim) : nullptr, n_w ? (const double*)(dv + L.obs_w) : nullptr, dX, dJ, dI,
                                  dS, q.flags);
  if (rc) return rc;
  h->d_J = dJ;  // where this call's costs lie (the exchange step of cilqr_multi_solve_batch reads them)
  if (packed) {
    HIP_TRY(hipMemcpyAsync(h->stage + L.U, dv + L.U, L.end - L.U, hipMemcpyDeviceToHost, s));  // U … status in one transfer
  } else {
    HIP_TRY(hipMemcpyAsync(q.U, dU, n_U, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(q.X_out, dX, B * 4 * (N + 1) * sizeof(double), hipMemcpyDeviceToHost, s));
    if (q.J_out) HIP_TRY(hipMemcpyAsync(q.J_out, dJ, B * sizeof(double), hipMemcpyDeviceToHost, s));
    if (q.iters_out) HIP_TRY(hipMemcpyAsync(q.iters_out, dI, B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (q.status_out) HIP_TRY(hipMemcpyAsync(q.status_out, dS, B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  }
  h->pending.active = true;
  h->pending.packed = packed;
  h->pending.L = L;
  h->pending.q = q;
  return CILQR_OK;
}
}  // namespace

int host_solve_finish(cilqr_handle* h) {
  if (!h->pending.active) return CILQR_OK;
  h->pending.active = false;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->pending.packed) {
    const IoLayout& L = h->pending.L;
    const HostBatch& q = h->pending.q;
    const size_t B = q.B, N = q.N;
    const char* st = h->stage;
    memcpy(q.U, st + L.U, B * 2 * N * sizeof(double));
    memcpy(q.X_out, st + L.X, B * 4 * (N + 1) * sizeof(double));
    if (q.J_out) memcpy(q.J_out, st + L.J, B * sizeof(double));
    if (q.iters_out) memcpy(q.iters_out, st + L.iters, B * sizeof(int32_t));
    if (q.status_out) memcpy(q.status_out, st + L.status, B * sizeof(int32_t));
  }
  return CILQR_OK;
}

// Device scratch of the convenience entry points that take host pointers (local plan, blur counts, conversions, test hooks):
// slots owned by the handle, grown when a call needs more than any before it — never allocated and freed per call.
int scratch_bytes(cilqr_handle* h, int slot, size_t bytes, void** out) {
  *out = nullptr;
  if (bytes == 0) return CILQR_OK;
  if (bytes > h->scratch_cap[slot]) {
    HIP_TRY(hipStreamSynchronize(h->stream));  // nothing enqueued may still use the old block
    if (h->scratch[slot]) HIP_TRY(hipFree(h->scratch[slot]));
    h->scratch[slot] = nullptr;
    h->scratch_cap[slot] = 0;
    const size_t want = bytes + bytes / 4;  // head-room: a slowly growing request does not reallocate every call
    HIP_TRY(hipMalloc(&h->scratch[slot], want));
    h->scratch_cap[slot] = want;
  }
  *out = h->scratch[slot];
  return CILQR_OK;
}

}  // namespace cilqr

extern "C" {

void* cilqr_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
    fail(CILQR_ERR_HIP, "cilqr_host_alloc: hipHostMalloc(%zu) failed", bytes);
    return nullptr;
  }
  return p;
}

int cilqr_debug_fail_enqueue(cilqr_handle* h, int nth_call) {
  if (!h || nth_call < 0) return fail(CILQR_ERR_ARG, "cilqr_debug_fail_enqueue: bad argument");
  h->debug_fail_enqueue = nth_call;
  return CILQR_OK;
}

int cilqr_host_free(void* p) {
  if (p) HIP_TRY(hipHostFree(p));
  return CILQR_OK;
}

}  // extern "C"
