// Where do the wavefronts of a multi-wavefront workgroup land, and what does a second resident wavefront cost the first?
// (diagnostic tool, not part of the product).  Two questions behind a two-wavefront workgroup per solve:
//   1. placement: with 1024 workgroups of W wavefronts on the 1024 SIMDs of an MI355X, does every SIMD get one wavefront 0
//      ("main": the serial phases) — or do the main wavefronts of a CU's four workgroups pile up on one SIMD?
//      Each wavefront records HW_ID (SIMD, CU, SE) and XCC_ID.
//   2. co-issue: a "main" wavefront runs a dependent fp64 chain (the shape of phases R and F); the "aux" wavefront of the same
//      workgroup either sleeps at the barrier or issues independent fp64 instructions (the shape of phase L) meanwhile.
//      Ticks per instruction of both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)

__device__ __forceinline__ unsigned hw_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  return v;
}
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v;
}

__global__ void place_kernel(unsigned* out, int spin) {
  const int w = threadIdx.x >> 6, W = blockDim.x >> 6;
  double x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = fma(x, 1.0000001, 0.5);  // keep every workgroup resident while the others arrive
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * W + w) * 2] = hw_id();
    out[(blockIdx.x * W + w) * 2 + 1] = xcc_id() | (x == 12345.678 ? 0x80000000u : 0u);
  }
}

// mode 0: aux sleeps at the barrier; 1: aux issues independent v_fma_f64 for about as long as main runs;
// 2: aux issues a dependent chain too (two serial chains on neighbouring SIMDs or one)
__global__ __launch_bounds__(128) void coissue_kernel(double* sink, unsigned long long* cyc, unsigned* where, double a, int mode, int n_it) {
  const int w = threadIdx.x >> 6;
  double x0 = a + threadIdx.x, x1 = a * 2, x2 = a * 3, x4 = a * 5, x5 = a * 6, x6 = a * 7, x7 = a * 8;
  unsigned long long t0 = __builtin_readcyclecounter(), t1 = t0;
  if (w == 0 || mode == 2) {
    for (int i = 0; i < n_it; ++i) { R16(asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x4) : "v"(x0), "v"(x1));) }
    t1 = __builtin_readcyclecounter();
  } else if (mode == 1) {
    for (int i = 0; i < n_it; ++i) {
      R4(asm volatile("v_fma_f64 %0, %4, %5, %6\n v_fma_f64 %1, %5, %6, %4\n v_fma_f64 %2, %6, %4, %5\n v_fma_f64 %3, %4, %6, %5"
                      : "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7) : "v"(x0), "v"(x1), "v"(x2));)
    }
    t1 = __builtin_readcyclecounter();
  }
  __syncthreads();
  sink[threadIdx.x + blockIdx.x * 128] = x0 + x1 + x2 + x4 + x5 + x6 + x7;
  if ((threadIdx.x & 63) == 0) {
    cyc[blockIdx.x * 2 + w] = t1 - t0;
    where[blockIdx.x * 2 + w] = hw_id();
  }
}

static void placement(int W, int blocks) {
  unsigned* d;
  (void)hipMalloc(&d, sizeof(unsigned) * 2 * blocks * W);
  for (int r = 0; r < 2; ++r) place_kernel<<<blocks, 64 * W>>>(d, 20000);
  (void)hipDeviceSynchronize();
  std::vector<unsigned> h(2 * blocks * W);
  (void)hipMemcpy(h.data(), d, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost);
  // per CU (xcc, se, sh, cu): wavefront-0 count per SIMD; per workgroup: distinct SIMDs of its wavefronts, one CU or not
  std::map<unsigned, std::vector<int>> mains, all;
  int split_cu = 0, distinct_hist[5] = {0, 0, 0, 0, 0};
  std::map<int, int> first_simd_hist;
  for (int b = 0; b < blocks; ++b) {
    unsigned cu0 = 0;
    int simd_mask = 0;
    for (int w = 0; w < W; ++w) {
      const unsigned id = h[(b * W + w) * 2], xcc = h[(b * W + w) * 2 + 1] & 0xF;
      const unsigned simd = (id >> 4) & 3, cu = (id >> 8) & 0xF, sh = (id >> 12) & 1, se = (id >> 13) & 7;
      const unsigned key = (xcc << 16) | (se << 8) | (sh << 4) | cu;
      if (w == 0) cu0 = key;
      else if (key != cu0) ++split_cu;
      simd_mask |= 1 << simd;
      auto& va = all[key];
      va.resize(4);
      ++va[simd];
      if (w == 0) {
        auto& vm = mains[key];
        vm.resize(4);
        ++vm[simd];
        ++first_simd_hist[(int)simd];
      }
    }
    ++distinct_hist[__builtin_popcount(simd_mask)];
  }
  int worst_main = 0, worst_all = 0;
  std::map<int, int> main_per_simd_hist, all_per_simd_hist;
  for (auto& kv : mains)
    for (int s = 0; s < 4; ++s) { ++main_per_simd_hist[kv.second[s]]; if (kv.second[s] > worst_main) worst_main = kv.second[s]; }
  for (auto& kv : all)
    for (int s = 0; s < 4; ++s) { ++all_per_simd_hist[kv.second[s]]; if (kv.second[s] > worst_all) worst_all = kv.second[s]; }
  printf("W=%d wavefronts per workgroup, %d workgroups: %zu CUs used; workgroups split over CUs %d\n", W, blocks, all.size(), split_cu);
  printf("  distinct SIMDs per workgroup:");
  for (int k = 1; k <= 4; ++k) printf(" %d:%d", k, distinct_hist[k]);
  printf("\n  SIMD of wavefront 0:");
  for (auto& kv : first_simd_hist) printf(" simd%d:%d", kv.first, kv.second);
  printf("\n  wavefront-0 count per (CU, SIMD) histogram:");
  for (auto& kv : main_per_simd_hist) printf(" %d:%d", kv.first, kv.second);
  printf("  (worst %d)\n  all-wavefront count per (CU, SIMD) histogram:", worst_main);
  for (auto& kv : all_per_simd_hist) printf(" %d:%d", kv.first, kv.second);
  printf("  (worst %d)\n", worst_all);
  // first few workgroups, raw
  for (int b = 0; b < 6; ++b) {
    printf("  wg %d:", b);
    for (int w = 0; w < W; ++w) {
      const unsigned id = h[(b * W + w) * 2], xcc = h[(b * W + w) * 2 + 1] & 0xF;
      printf(" [xcc%u se%u cu%u simd%u slot%u]", xcc, (id >> 13) & 7, (id >> 8) & 0xF, (id >> 4) & 3, id & 0xF);
    }
    printf("\n");
  }
  (void)hipFree(d);
}

static void coissue(int mode, int blocks) {
  const int n_it = 256;
  double* sink; unsigned long long* cyc; unsigned* where;
  (void)hipMalloc(&sink, 8 * 128 * blocks); (void)hipMalloc(&cyc, 16 * blocks); (void)hipMalloc(&where, 8 * blocks);
  for (int r = 0; r < 2; ++r) coissue_kernel<<<blocks, 128>>>(sink, cyc, where, 1.0000001, mode, n_it);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(2 * blocks);
  std::vector<unsigned> hw(2 * blocks);
  (void)hipMemcpy(h.data(), cyc, 16 * blocks, hipMemcpyDeviceToHost);
  (void)hipMemcpy(hw.data(), where, 8 * blocks, hipMemcpyDeviceToHost);
  double m0 = 0, m1 = 0, worst0 = 0; int same = 0;
  for (int b = 0; b < blocks; ++b) {
    m0 += h[2 * b]; m1 += h[2 * b + 1];
    if ((double)h[2 * b] > worst0) worst0 = (double)h[2 * b];
    if (((hw[2 * b] >> 4) & 3) == ((hw[2 * b + 1] >> 4) & 3)) ++same;
  }
  const char* names[] = {"aux asleep at the barrier", "aux issues independent fp64", "aux runs a dependent chain too"};
  printf("co-issue, %4d workgroups x 2 wavefronts, %s: main %.2f ticks/instr (worst workgroup %.2f), aux %.2f ticks/instr; both on one SIMD in %d workgroups\n",
         blocks, names[mode], m0 / blocks / n_it / 16, worst0 / n_it / 16, mode ? m1 / blocks / n_it / 16 : 0.0, same);
  (void)hipFree(sink); (void)hipFree(cyc); (void)hipFree(where);
}

int main() {
  for (int W : {1, 2, 4}) placement(W, 1024);
  placement(2, 512);
  for (int blocks : {256, 1024})
    for (int m = 0; m < 3; ++m) coissue(m, blocks);
  return 0;
}
