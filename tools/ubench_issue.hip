// Issue cost of individual gfx950 instructions for ONE wavefront per SIMD (diagnostic tool, not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_IT 1024
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)
template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, unsigned long long* cyc, double a, double b) {
  double x0 = a + threadIdx.x, x1 = a * 2, x2 = a * 3, x3 = a * 4, x4 = a * 5, x5 = a * 6, x6 = a * 7, x7 = a * 8;
  __shared__ double buf[512];
  buf[threadIdx.x] = a;
  int addr = 0;
  double2 q0 = {0, 0}, q1 = {0, 0};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N_IT; ++i) {
    if (MODE == 0) { R16(asm volatile("v_fma_f64 %0, %1, %2, %3\n v_fma_f64 %4, %2, %3, %1\n v_fma_f64 %5, %3, %1, %2\n v_fma_f64 %6, %1, %3, %2" : "=&v"(x4), "+v"(x0), "+v"(x1), "+v"(x2), "=&v"(x5), "=&v"(x6), "=&v"(x7));) }
    if (MODE == 1) { R16(asm volatile("v_fmac_f64 %0, %1, %2\n v_fmac_f64 %3, %2, %1\n v_fmac_f64 %4, %1, %2\n v_fmac_f64 %5, %2, %1" : "+v"(x4), "+v"(x0), "+v"(x1), "+v"(x5), "+v"(x6), "+v"(x7));) }
    if (MODE == 2) { R16(asm volatile("v_mul_f64 %0, %1, %2\n v_mul_f64 %3, %2, %1\n v_mul_f64 %4, %1, %2\n v_mul_f64 %5, %2, %1" : "=&v"(x4), "+v"(x0), "+v"(x1), "=&v"(x5), "=&v"(x6), "=&v"(x7));) }
    if (MODE == 3) { R16(asm volatile("v_add_f64 %0, %1, %2\n v_add_f64 %3, %2, %1\n v_add_f64 %4, %1, %2\n v_add_f64 %5, %2, %1" : "=&v"(x4), "+v"(x0), "+v"(x1), "=&v"(x5), "=&v"(x6), "=&v"(x7));) }
    if (MODE == 4) { R16(asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %2, %1\n v_mov_b64 %3, %1\n v_mov_b64 %4, %1" : "=&v"(x4), "+v"(x0), "=&v"(x5), "=&v"(x6), "=&v"(x7));) }
    if (MODE == 5) { R16(asm volatile("s_mov_b32 s20, 0x3ff00000\n s_mov_b32 s21, 0x3ff00001\n s_mov_b32 s22, 0x3ff00002\n s_mov_b32 s23, 0x3ff00003" ::: "s20", "s21", "s22", "s23");) }
    if (MODE == 6) { R16(asm volatile("s_mov_b32 s20, 1\n s_mov_b32 s21, 2\n s_mov_b32 s22, 3\n s_mov_b32 s23, 4" ::: "s20", "s21", "s22", "s23");) }
    if (MODE == 7) { R16(asm volatile("v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %2, %1, %0\n v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %2, %1, %0" : "+v"(x4), "+v"(x0), "+v"(x1));) }
    if (MODE == 8) { R16(asm volatile("v_fma_f64 %0, %1, %2, %3\n v_fma_f64 %4, %2, %3, %1\n v_fma_f64 %5, %3, %1, %2\n v_fma_f64 %6, %1, %3, %2" : "=&v"(x4), "+v"(x0), "+s"(a), "+v"(x2), "=&v"(x5), "=&v"(x6), "=&v"(x7));) }
    if (MODE == 9) { R16(asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n ds_read_b128 %0, %2 offset:32\n ds_read_b128 %1, %2 offset:48\n" : "=&v"(q0), "=&v"(q1) : "v"(addr) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)"); }
    if (MODE == 10) { int i1 = addr + 1, i2 = addr + 2, i3, i4, i5; R16(asm volatile("v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %3, %2, %1, vcc\n v_xor_b32 %4, %1, %2\n v_and_b32 %5, %1, %2" : "=&v"(addr), "+v"(i1), "+v"(i2), "=&v"(i3), "=&v"(i4), "=&v"(i5) :: "vcc");) addr += i3 + i4 + i5; }
    if (MODE == 11) { R16(asm volatile("v_min_f64 %0, %1, %2\n v_max_f64 %3, %2, %1\n v_min_f64 %4, %1, %2\n v_max_f64 %5, %2, %1" : "=&v"(x4), "+v"(x0), "+v"(x1), "=&v"(x5), "=&v"(x6), "=&v"(x7));) }
    if (MODE == 12) { R16(asm volatile("v_rcp_f64 %0, %1\n v_rcp_f64 %2, %1\n v_rcp_f64 %3, %1\n v_rcp_f64 %4, %1" : "=&v"(x4), "+v"(x0), "=&v"(x5), "=&v"(x6), "=&v"(x7));) }
    if (MODE == 13) { R16(asm volatile("v_rndne_f64 %0, %1\n v_cvt_i32_f64 %2, %1\n v_rndne_f64 %3, %1\n v_cvt_i32_f64 %4, %1" : "=&v"(x4), "+v"(x0), "=&v"(addr), "=&v"(x6), "=&v"(addr));) }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x * 64] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + q0.x + q1.y + addr;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, int blocks) {
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 8 * 64 * blocks); (void)hipMalloc(&cyc, 8 * blocks);
  k<MODE><<<blocks, 64>>>(out, cyc, 1.0000001, 0.9999999);
  k<MODE><<<blocks, 64>>>(out, cyc, 1.0000001, 0.9999999);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks); (void)hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += v; mean /= blocks;
  printf("%-44s waves/SIMD %d: %.2f ticks/instr\n", name, blocks / 1024, mean / N_IT / 64);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  for (int blocks : {1024, 2048}) {
    run<0>("v_fma_f64 vD, vA, vB, vC (VOP3, indep)", blocks);
    run<7>("v_fma_f64 acc chain (dst=src2)", blocks);
    run<8>("v_fma_f64 with one SGPR source", blocks);
    run<1>("v_fmac_f64 (VOP2)", blocks);
    run<2>("v_mul_f64", blocks);
    run<3>("v_add_f64", blocks);
    run<11>("v_min/max_f64", blocks);
    run<4>("v_mov_b64", blocks);
    run<10>("v_cndmask/xor/and b32", blocks);
    run<12>("v_rcp_f64", blocks);
    run<13>("v_rndne_f64 / v_cvt_i32_f64", blocks);
    run<5>("s_mov_b32 literal (8 B)", blocks);
    run<6>("s_mov_b32 inline const (4 B)", blocks);
    run<9>("ds_read_b128 x64 then wait", blocks);
  }
  return 0;
}
