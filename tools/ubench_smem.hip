// Feasibility probe (diagnostic tool, not part of the product): a wavefront writes records to global memory with vector
// stores, then reads them back through the SCALAR path (s_load_dwordx16) after s_waitcnt vmcnt(0) + s_dcache_inv.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(64) void k(double* buf, double* out, unsigned long long* cyc, int rounds, int N) {
  double* mine = buf + (size_t)blockIdx.x * N * 16;
  const int lane = threadIdx.x;
  double acc = 0.0;
  unsigned long long t0 = 0, t1 = 0, tl = 0;
  for (int r = 0; r < rounds; ++r) {
    // lane t writes record t (16 doubles), values depend on round
    for (int t = lane; t < N; t += 64)
      for (int f = 0; f < 16; ++f) mine[t * 16 + f] = (double)(r * 1000 + t * 16 + f);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)\n s_dcache_inv\n s_waitcnt lgkmcnt(0)" ::: "memory");
    if (r == rounds - 1) t0 = __builtin_readcyclecounter();
    for (int j = N - 1; j >= 0; --j) {
      d8 a, b;
      const double* p = mine + j * 16;
      asm volatile("s_load_dwordx16 %0, %2, 0x0\n s_load_dwordx16 %1, %2, 0x40\n s_waitcnt lgkmcnt(0)" : "=s"(a), "=s"(b) : "s"(p) : "memory");
      acc += a[0] + a[7] + b[0] + b[7];
      // verify
      if (a[0] != (double)(r * 1000 + j * 16) || b[7] != (double)(r * 1000 + j * 16 + 15)) acc = -1e300;
    }
    if (r == rounds - 1) t1 = __builtin_readcyclecounter();
  }
  (void)tl;
  out[blockIdx.x * 64 + lane] = acc;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  const int blocks = 1024, N = 50, rounds = 8;
  double *buf, *out; unsigned long long* cyc;
  hipMalloc(&buf, sizeof(double) * blocks * N * 16); hipMalloc(&out, 8 * 64 * blocks); hipMalloc(&cyc, 8 * blocks);
  k<<<blocks, 64>>>(buf, out, cyc, rounds, N);
  hipDeviceSynchronize();
  std::vector<double> h(64 * blocks); hipMemcpy(h.data(), out, 8 * 64 * blocks, hipMemcpyDeviceToHost);
  std::vector<unsigned long long> c(blocks); hipMemcpy(c.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
  int bad = 0; for (double v : h) if (v < 0) bad++;
  double mean = 0; for (auto v : c) mean += v; mean /= blocks;
  printf("bad lanes %d of %d; ticks per step (2 x s_load_dwordx16 + wait, no prefetch): %.1f\n", bad, 64 * blocks, mean / N);
  return 0;
}
