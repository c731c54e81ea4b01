// Does gfx950 skip the quarter-wave passes of an fp64 VALU instruction whose lanes are switched off?  (diagnostic tool, not
// part of the product).  The serial phases of the one-wavefront-per-solve kernel keep one useful lane in 64: if a wave64
// v_fma_f64 with EXEC = lanes 0-15 (or lane 0 alone) issued faster than with all 64 lanes, running those phases under a narrow
// EXEC would shorten the chain.  Dependent and independent chains of 4096 v_fma_f64 with 64, 32, 16 and 1 lanes switched on (by an ordinary branch), one wavefront per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_IT 256
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
template <int DEP>
__global__ __launch_bounds__(64) void k(double* out, unsigned long long* cyc, double a, int lanes) {
  double x0 = a + threadIdx.x, x1 = a * 2, x2 = a * 3, x4 = a * 5, x5 = a * 6, x6 = a * 7, x7 = a * 8;
  unsigned long long t0 = 0, t1 = 0;
  if ((int)threadIdx.x < lanes) {  // EXEC narrowed by the compiler's own branch: nothing forces it behind its back
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N_IT; ++i) {
      if (DEP) { R16(asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x4) : "v"(x0), "v"(x1));) }
      else { R4(asm volatile("v_fma_f64 %0, %4, %5, %6\n v_fma_f64 %1, %5, %6, %4\n v_fma_f64 %2, %6, %4, %5\n v_fma_f64 %3, %4, %6, %5" : "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7) : "v"(x0), "v"(x1), "v"(x2));) }
    }
    t1 = __builtin_readcyclecounter();
  }
  out[threadIdx.x + blockIdx.x * 64] = x0 + x1 + x2 + x4 + x5 + x6 + x7;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int DEP> void run(const char* name, int mask) {
  const int blocks = 1024;
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 8 * 64 * blocks); (void)hipMalloc(&cyc, 8 * blocks);
  for (int r = 0; r < 2; ++r) k<DEP><<<blocks, 64>>>(out, cyc, 1.0000001, mask);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks); (void)hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += v; mean /= blocks;
  printf("%-28s %2d lanes on: %.2f ticks/instr\n", name, mask, mean / N_IT / 16);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  for (int m : {64, 32, 16, 1}) {
    run<1>("v_fma_f64 dependent chain", m);
    run<0>("v_fma_f64 independent", m);
  }
  return 0;
}
