// local_plan.hip — batched LocalPlanner on the device: for each candidate ego pose the nearest global
// waypoint, the slice of at most num_of_local_wpts waypoints and the degree-poly_order fit y(x), i.e. the reference's
// LocalPlanner::{closest_point_index,get_local_wpts,get_local_plan,get_local_plan_coeffs,polyfit}
// (I/LocalPlanner.cpp:25-117), so that a batch solve can start from raw (global_path, ego) without B host fits per tick.
//
// Mapping: one lane per candidate (the fit is a 20×6 problem with data-dependent pivoting — nothing to spread over
// lanes without changing the order of the sums, and the order of the sums is what keeps the result equal to the host
// pre-step's bit for bit).  Each lane's working storage is an LDS slot, element-major so that the 32 lanes of a
// workgroup touch 32 different banks.  The Vandermonde entries x^j are formed in double-double and rounded once
// (correctly rounded except within ~1e-16 ulp of a tie) — the host libm the reference calls returns the same value
// except for about one call in a thousand (glibc ≥ 2.28: < 0.52 ulp, measured 11-23 misrounded of 20 000 per exponent;
// earlier glibc: always correctly rounded).
#include <hip/hip_runtime.h>

#include "cilqr_internal.h"
#include "vandermonde_qr.hpp"

namespace cilqr {
namespace {

constexpr int LPB = 32;  // lanes (= candidates) per workgroup

__device__ __forceinline__ double pow_small_int(double x, int j) {
#pragma clang fp contract(off)  // p + e below must not become fma(hi, x, e)
  if (j == 0) return 1.0;
  double hi = x, lo = 0.0;
  for (int t = 1; t < j; ++t) {  // (hi + lo) * x, kept as an unevaluated sum
    const double p = hi * x;
    double e = __builtin_fma(hi, x, -p);
    e = __builtin_fma(lo, x, e);
    const double s = p + e;
    lo = e - (s - p);
    hi = s;
  }
  return hi;
}

struct LdsStore {
  double* d;  // this lane's first double; element e at d[e * LPB]
  int* n;     // this lane's first int
  int rows_max, cols;
  __device__ double& m(int i, int j) { return d[(j * rows_max + i) * LPB]; }
  __device__ double& c(int i) { return d[(rows_max * cols + i) * LPB]; }
  __device__ double& col_norm(int j) { return d[(rows_max * (cols + 1) + j) * LPB]; }
  __device__ double& tau(int k) { return d[(rows_max * (cols + 1) + cols + k) * LPB]; }
  __device__ int& swap_with(int k) { return n[k * LPB]; }
  __device__ int& order(int j) { return n[(cols + j) * LPB]; }
};

// RM: rows of the fit known at compile time (num_of_local_wpts; 20 in the reference) → unrolled row loops; 0 → plain loops.
template <int RM>
__global__ __launch_bounds__(LPB) void local_plan_kernel(LocalPlanArgs a) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x;
  const int b = blockIdx.x * LPB + lane;
  if (b >= a.B) return;
  const int doubles_per_lane = a.n_wpts * (a.cols + 1) + 2 * a.cols;
  LdsStore s;
  s.d = lds + lane;
  s.n = reinterpret_cast<int*>(lds + (size_t)doubles_per_lane * LPB) + lane;
  s.rows_max = a.n_wpts;
  s.cols = a.cols;

  const double* path = a.path + (size_t)b * a.path_stride;
  const double ex = a.ego[4 * (size_t)b], ey = a.ego[4 * (size_t)b + 1];
  // closest_point_index (:25-41): strict-< first minimum over squared distance
  double dx = ex - path[0], dy = ey - path[1];
  double best = dx * dx + dy * dy;
  int first = 0;
#pragma unroll 4
  for (int i = 1; i < a.P; ++i) {
    dx = ex - path[2 * i];
    dy = ey - path[2 * i + 1];
    const double dist = dx * dx + dy * dy;
    if (dist < best) {
      best = dist;
      first = i;
    }
  }
  // get_local_wpts (:47-60)
  const int n = (a.P - first) < a.n_wpts ? (a.P - first) : a.n_wpts;
  for (int i = 0; i < n; ++i) {
    const double x = path[2 * (first + i)];
    for (int j = 0; j < a.cols; ++j) s.m(i, j) = pow_small_int(x, j);
    s.c(i) = path[2 * (first + i) + 1];
  }
  double coeffs[CILQR_POLY_COEFFS];
  {
    // the store's matrix stride is rows_max; the fit works on the leading n rows
    vandermonde_lstsq<RM>(s, n, a.cols, coeffs);
  }
  for (int j = 0; j < a.cols; ++j) a.poly[(size_t)b * CILQR_POLY_COEFFS + j] = coeffs[j];
  for (int j = a.cols; j < CILQR_POLY_COEFFS; ++j) a.poly[(size_t)b * CILQR_POLY_COEFFS + j] = 0.0;
  a.xplan_fl[2 * (size_t)b] = path[2 * first];
  a.xplan_fl[2 * (size_t)b + 1] = path[2 * (first + n - 1)];
  if (a.ref_traj) {  // get_local_plan (:66-85): row 0 = waypoint x, row 1 = fitted y, ascending powers
    double* r = a.ref_traj + (size_t)b * 2 * a.n_wpts;
    for (int i = 0; i < n; ++i) {
      const double x = path[2 * (first + i)];
      double fy = 0.0;
      for (int j = 0; j < a.cols; ++j) fy += coeffs[j] * pow_small_int(x, j);
      r[2 * i] = x;
      r[2 * i + 1] = fy;
    }
  }
  if (a.n_out) a.n_out[b] = n;
}

}  // namespace

size_t local_plan_lds_bytes(int n_wpts, int cols) {
  return (size_t)LPB * ((size_t)(n_wpts * (cols + 1) + 2 * cols) * sizeof(double) + (size_t)2 * cols * sizeof(int));
}

hipError_t launch_local_plan(const LocalPlanArgs& a, hipStream_t stream) {
  const size_t lds = local_plan_lds_bytes(a.n_wpts, a.cols);
  if (lds > 64 * 1024) return hipErrorInvalidValue;  // checked against the parameters at cilqr_create
  if (a.n_wpts == 20) local_plan_kernel<20><<<dim3((a.B + LPB - 1) / LPB), dim3(LPB), lds, stream>>>(a);
  else local_plan_kernel<0><<<dim3((a.B + LPB - 1) / LPB), dim3(LPB), lds, stream>>>(a);
  return hipGetLastError();
}

}  // namespace cilqr
