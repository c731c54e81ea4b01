// cilqr_select.hip — min-cost selection over a batch and across ranks (new in this build: the reference never
// compares candidates).  Cost = Constraints::get_J of the result (I/Constraints.cpp:534-561), tie-break = lowest index, the
// strict-< first-minimum convention of the reference's own argmins (I/Constraints.cpp:50, I/LocalPlanner.cpp:34).
//   argmin_kernel  one 1024-thread workgroup: lexicographic (J, index) minimum of J[B], NaN never wins; wavefront shuffles and
//                  one LDS stage; writes {J_min, index} and, for the cross-rank step, {J_min, index, index offset of the rank}.
//   select_kernel  one wavefront: lexicographic minimum over the gathered per-rank triples (after ONE ncclAllGather of 24 bytes
//                  per rank, cilqr_comm.cpp) → {J_min, global index}; a rank without a finite cost (index -1) never wins.
#include "cilqr_internal.h"

namespace cilqr {

namespace {

constexpr int AM_THREADS = 1024;

__device__ __forceinline__ void amin_merge(double& j0, long long& i0, double j1, long long i1) {
  if (j1 < j0 || (j1 == j0 && i1 < i0)) { j0 = j1; i0 = i1; }
}

// pair_offset: added to the index written to out_pair (a lone rank's global index without the gather and select steps).
__global__ __launch_bounds__(AM_THREADS) void argmin_kernel(const double* J, int B, double* out_pair, double* out_triple, double offset,
                                                            double pair_offset) {
  __shared__ double sj[AM_THREADS / 64];
  __shared__ long long si[AM_THREADS / 64];
  constexpr long long NONE = 0x7fffffffffffffffll;
  double bj = __builtin_huge_val();
  long long bi = NONE;
  for (int i = threadIdx.x; i < B; i += AM_THREADS) amin_merge(bj, bi, J[i], i);
  for (int o = 32; o > 0; o >>= 1) {
    const double oj = __shfl_xor(bj, o, 64);
    const long long oi = __shfl_xor(bi, o, 64);
    amin_merge(bj, bi, oj, oi);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sj[wave] = bj; si[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < AM_THREADS / 64; ++w) amin_merge(bj, bi, sj[w], si[w]);
    const double idx = (bi == NONE) ? -1.0 : (double)bi;
    if (out_pair) { out_pair[0] = bj; out_pair[1] = idx < 0.0 ? idx : idx + pair_offset; }
    if (out_triple) { out_triple[0] = bj; out_triple[1] = idx; out_triple[2] = offset; }
  }
}

__global__ __launch_bounds__(64) void select_kernel(const double* triples, int n_ranks, double* out_pair) {
  constexpr long long NONE = 0x7fffffffffffffffll;
  double bj = __builtin_huge_val();
  long long bi = NONE;
  for (int r = threadIdx.x; r < n_ranks; r += 64) {
    const double j = triples[3 * r], idx = triples[3 * r + 1], off = triples[3 * r + 2];
    if (idx >= 0.0) amin_merge(bj, bi, j, (long long)idx + (long long)off);  // NaN J never passes amin_merge's comparisons
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double oj = __shfl_xor(bj, o, 64);
    const long long oi = __shfl_xor(bi, o, 64);
    amin_merge(bj, bi, oj, oi);
  }
  if (threadIdx.x == 0) {
    out_pair[0] = bj;
    out_pair[1] = (bi == NONE) ? -1.0 : (double)bi;
  }
}

}  // namespace

hipError_t launch_argmin(const double* J, int B, double* out_pair, double* out_triple, double offset, hipStream_t stream,
                         double pair_offset) {
  hipLaunchKernelGGL(argmin_kernel, dim3(1), dim3(AM_THREADS), 0, stream, J, B, out_pair, out_triple, offset, pair_offset);
  return hipGetLastError();
}

hipError_t launch_select(const double* triples, int n_ranks, double* out_pair, hipStream_t stream) {
  hipLaunchKernelGGL(select_kernel, dim3(1), dim3(64), 0, stream, triples, n_ranks, out_pair);
  return hipGetLastError();
}

}  // namespace cilqr
