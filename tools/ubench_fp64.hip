// Micro-benchmarks of single-wave fp64 issue/latency on gfx950 (diagnostic tool, not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_IT 4096
template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, unsigned long long* cyc, double a, double b) {
  double x0 = a + threadIdx.x, x1 = a * 2, x2 = a * 3, x3 = a * 4, x4 = a * 5, x5 = a * 6, x6 = a * 7, x7 = a * 8;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N_IT; ++i) {
    if (MODE == 0) {  // dependent chain of 8 fma
#pragma unroll
      for (int j = 0; j < 8; ++j) x0 = __builtin_fma(x0, b, a);
    } else if (MODE == 1) {  // 8 independent fma
      x0 = __builtin_fma(x0, b, a); x1 = __builtin_fma(x1, b, a); x2 = __builtin_fma(x2, b, a); x3 = __builtin_fma(x3, b, a);
      x4 = __builtin_fma(x4, b, a); x5 = __builtin_fma(x5, b, a); x6 = __builtin_fma(x6, b, a); x7 = __builtin_fma(x7, b, a);
    } else if (MODE == 2) {  // 2 interleaved chains
#pragma unroll
      for (int j = 0; j < 4; ++j) { x0 = __builtin_fma(x0, b, a); x1 = __builtin_fma(x1, b, a); }
    } else if (MODE == 3) {  // dependent division chain (8 ops: 1 div + ... ) -> just div
      x0 = a / (x0 + b);
    } else if (MODE == 4) {  // sqrt
      x0 = __builtin_sqrt(x0 + b);
    } else if (MODE == 5) {  // dependent add chain
#pragma unroll
      for (int j = 0; j < 8; ++j) x0 = x0 + b;
    } else if (MODE == 6) {  // dependent mul chain f32 for comparison
      float f = (float)x0;
#pragma unroll
      for (int j = 0; j < 8; ++j) f = __builtin_fmaf(f, (float)b, (float)a);
      x0 = f;
    } else if (MODE == 8) {  // VOP3 v_fma_f64 with three distinct VGPR sources and a distinct destination
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(x4) : "v"(x0), "v"(x1), "v"(x2));
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(x5) : "v"(x1), "v"(x2), "v"(x3));
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(x6) : "v"(x2), "v"(x3), "v"(x0));
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(x7) : "v"(x3), "v"(x0), "v"(x1));
      }
    } else if (MODE == 9) {  // v_fmac_f64 (VOP2) all-VGPR
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(x4) : "v"(x0), "v"(x1));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(x5) : "v"(x1), "v"(x2));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(x6) : "v"(x2), "v"(x3));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(x7) : "v"(x3), "v"(x0));
      }
    } else if (MODE == 10) {  // v_mul_f64 all-VGPR
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(x4) : "v"(x0), "v"(x1));
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(x5) : "v"(x1), "v"(x2));
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(x6) : "v"(x2), "v"(x3));
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(x7) : "v"(x3), "v"(x0));
      }
    } else if (MODE == 11) {  // v_add_f64 all-VGPR
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        asm volatile("v_add_f64 %0, %1, %2" : "=v"(x4) : "v"(x0), "v"(x1));
        asm volatile("v_add_f64 %0, %1, %2" : "=v"(x5) : "v"(x1), "v"(x2));
        asm volatile("v_add_f64 %0, %1, %2" : "=v"(x6) : "v"(x2), "v"(x3));
        asm volatile("v_add_f64 %0, %1, %2" : "=v"(x7) : "v"(x3), "v"(x0));
      }
    } else if (MODE == 12) {  // v_mov_b64 / s_mov / v_cndmask mix: 8 v_mov_b32
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_mov_b64 %0, %1" : "=v"(x4) : "v"(x0));
    } else if (MODE == 13) {  // 8 s_mov_b32
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("s_mov_b32 s20, 0x3ff00000" ::: "s20");
    } else if (MODE == 14) {  // 8 ds_read_b128 uniform address, then wait
      __shared__ double buf[64];
      double2 t0v;
      int zero_addr = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(t0v) : "v"(zero_addr));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      x4 = t0v.x + buf[0];
    } else if (MODE == 7) {  // 4 interleaved chains
#pragma unroll
      for (int j = 0; j < 2; ++j) { x0 = __builtin_fma(x0, b, a); x1 = __builtin_fma(x1, b, a); x2 = __builtin_fma(x2, b, a); x3 = __builtin_fma(x3, b, a); }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x * 64] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// LDS store (lane 0) -> uniform load round trip
__global__ __launch_bounds__(64) void lds_rt(double* out, unsigned long long* cyc, double a) {
  __shared__ double buf[64];
  double x = a;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N_IT; ++i) {
    if (threadIdx.x == 0) buf[i & 31] = x;
    __syncthreads();
    x = buf[i & 31] + 1.0;
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, int per_it, int blocks) {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 8 * 64 * blocks); hipMalloc(&cyc, 8 * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 64>>>(out, cyc, 1.0000001, 0.9999999);
  hipEventRecord(e0);
  k<MODE><<<blocks, 64>>>(out, cyc, 1.0000001, 0.9999999);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks); hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += v; mean /= blocks;
  printf("%-28s blocks %5d: %.2f memtime-ticks/op  (kernel %.3f ms -> %.2f ns/op)\n", name, blocks, mean / N_IT / per_it, ms, ms * 1e6 / N_IT / per_it);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int blocks : {1024, 2048}) {
    run<0>("dep fma x8", 8, blocks);
    run<1>("indep fma x8", 8, blocks);
    run<2>("2 chains", 8, blocks);
    run<7>("4 chains", 8, blocks);
    run<5>("dep add x8", 8, blocks);
    run<3>("dep div (+add)", 1, blocks);
    run<4>("dep sqrt (+add)", 1, blocks);
    run<6>("dep fma f32 x8", 8, blocks);
    run<8>("v_fma_f64 VOP3 3xVGPR", 8, blocks);
    run<9>("v_fmac_f64 VOP2", 8, blocks);
    run<10>("v_mul_f64", 8, blocks);
    run<11>("v_add_f64", 8, blocks);
    run<12>("v_mov_b32", 8, blocks);
    run<13>("s_mov_b32", 8, blocks);
    run<14>("8x ds_read_b128 + wait", 8, blocks);
  }
  double* out; unsigned long long* cyc; hipMalloc(&out, 8 * 64); hipMalloc(&cyc, 8);
  lds_rt<<<1, 64>>>(out, cyc, 1.0); hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("LDS store->barrier->load->add round trip: %.1f ticks\n", (double)h / N_IT);
  return 0;
}
