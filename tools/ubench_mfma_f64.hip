// v_mfma_f64_4x4x4_4b_f64 on gfx950 (diagnostic tool, not part of the product): which lane holds which entry of A, B and D,
// in what order the four products of an entry are accumulated, and what a lone wavefront pays for a chain of them.  The
// Riccati step of the solver is a chain of 4x4 fp64 matrix products done redundantly on 64 lanes; this instruction does four
// such products at once (one per 16-lane row), so the answers decide whether the step can be restated on it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0)

// (1) lane maps: A one-hot at lane la, B = lane + 1 everywhere -> D[ld] = (lb + 1) tells which B lane meets A lane la in D lane ld.
__global__ __launch_bounds__(64) void probe(double* out) {
  const int l = threadIdx.x;
  for (int la = 0; la < 64; ++la) {
    const double a = l == la ? 1.0 : 0.0, b = l + 1.0;
    out[la * 64 + l] = MFMA(a, b, 0.0);
  }
}

// (2) order of accumulation in one entry: entry (0, 0) of block 0 with c and four products chosen by the host.
__global__ __launch_bounds__(64) void order(const double* a, const double* b, const double* c, double* d) {
  const int l = threadIdx.x;
  d[l] = MFMA(a[l], b[l], c[l]);
}

// (3) time: 256 x 16 instructions, one wavefront per SIMD.  KIND 0: result feeds C; 1: result feeds A; 2: result feeds B;
// 3: four independent accumulators; 4: result -> one v_fma_f64 -> A of the next (the VALU round trip of a real chain).
#define V3 asm volatile("v_fma_f64 %0, %3, %4, %0\n v_fma_f64 %1, %3, %4, %1\n v_fma_f64 %2, %3, %4, %2" : "+v"(y0), "+v"(y1), "+v"(y2) : "v"(y3), "v"(y4));
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
template <int KIND>
__global__ __launch_bounds__(64) void chain(double* out, unsigned long long* cyc, double s) {
  const int l = threadIdx.x;
  double a = (l % 5 == 0) ? 1.0 : 0.0, b = (l % 5 == 0) ? 1.0 : 0.0, c = s * l;  // identity-like operands keep values bounded
  double c1 = c + 1, c2 = c + 2, c3 = c + 3;
  double y0 = s, y1 = s + 1, y2 = s + 2, y3 = 0.5, y4 = 0.25;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 256; ++i) {
    if (KIND == 0) { R16(c = MFMA(a, b, c);) }
    if (KIND == 1) { R16(a = MFMA(a, b, 0.0);) }
    if (KIND == 2) { R16(b = MFMA(a, b, 0.0);) }
    if (KIND == 3) { R4(c = MFMA(a, b, c); c1 = MFMA(a, b, c1); c2 = MFMA(a, b, c2); c3 = MFMA(a, b, c3);) }
    if (KIND == 4) { R16(a = fma(MFMA(a, b, 0.0), s, 0.0);) }
    // 5: independent matrix instructions, each with three independent v_fma_f64 beside it — does vector work issue under the
    //    16 ticks a matrix instruction holds its pipe?  6: the same around a C-dependent chain.
    if (KIND == 5) { R4(c = MFMA(a, b, c); V3 c1 = MFMA(a, b, c1); V3 c2 = MFMA(a, b, c2); V3 c3 = MFMA(a, b, c3); V3) }
    if (KIND == 6) { R16(c = MFMA(a, b, c); V3) }
    // 7: six independent v_fma_f64 beside each independent matrix instruction
    if (KIND == 7) { R4(c = MFMA(a, b, c); V3 V3 c1 = MFMA(a, b, c1); V3 V3 c2 = MFMA(a, b, c2); V3 V3 c3 = MFMA(a, b, c3); V3 V3) }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[l + 64 * blockIdx.x] = a + b + c + c1 + c2 + c3 + y0 + y1 + y2;
  if (l == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND> void run(const char* name, int per) {
  const int blocks = 1024;
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 8 * 64 * blocks); (void)hipMalloc(&cyc, 8 * blocks);
  for (int r = 0; r < 2; ++r) chain<KIND><<<blocks, 64>>>(out, cyc, 1.0);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks); (void)hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += v; mean /= blocks;
  printf("%-44s %.2f ticks per %s\n", name, mean / 256 / 16, per ? "mfma + v_fma pair" : "mfma");
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  double* d; (void)hipMalloc(&d, 8 * 64 * 64);
  probe<<<1, 64>>>(d); (void)hipDeviceSynchronize();
  std::vector<double> h(64 * 64); (void)hipMemcpy(h.data(), d, 8 * 64 * 64, hipMemcpyDeviceToHost);
  printf("A lane -> (D lane : B lane) pairs\n");
  for (int la = 0; la < 64; ++la) {
    printf("A %2d:", la);
    for (int ld = 0; ld < 64; ++ld) if (h[la * 64 + ld] != 0.0) printf(" (%d:%d)", ld, (int)h[la * 64 + ld] - 1);
    printf("\n");
  }
  // the map the solver relies on: A_b[i][k] at lane i + 4b + 16k, B_b[k][j] at lane j + 4b + 16k, D_b[i][j] at lane j + 4b + 16i
  // (block b = lane bits 2-3, the 16-lane row = k for the inputs and i for the result)
  int bad = 0;
  for (int la = 0; la < 64; ++la) {
    const int i = la % 4, bl = (la / 4) % 4, k = la / 16;
    for (int ld = 0; ld < 64; ++ld) {
      double want = 0.0;
      if ((ld / 4) % 4 == bl && ld / 16 == i) { const int j = ld % 4; want = (j + 4 * bl + 16 * k) + 1.0; }
      if (h[la * 64 + ld] != want) ++bad;
    }
  }
  printf("lane map A[i][k]@i+4b+16k, B[k][j]@j+4b+16k, D[i][j]@j+4b+16i: %s (%d mismatches)\n", bad ? "WRONG" : "holds", bad);

  // order of accumulation: entry (0,0) of block 0: a[0][k] at lane 16k, b[k][0] at lane 16k, c at lane 0
  double *pa, *pb, *pc, *pd;
  (void)hipMalloc(&pa, 512); (void)hipMalloc(&pb, 512); (void)hipMalloc(&pc, 512); (void)hipMalloc(&pd, 512);
  const double big = 9007199254740992.0;  // 2^53: c = 1 is absorbed if added to it first
  for (int kp = 0; kp < 4; ++kp)
    for (int km = 0; km < 4; ++km) {
      if (kp == km) continue;
      std::vector<double> a(64, 0.0), b(64, 0.0), c(64, 0.0), r(64);
      a[16 * kp] = big; b[16 * kp] = 1.0; a[16 * km] = -big; b[16 * km] = 1.0; c[0] = 1.0;
      (void)hipMemcpy(pa, a.data(), 512, hipMemcpyHostToDevice); (void)hipMemcpy(pb, b.data(), 512, hipMemcpyHostToDevice);
      (void)hipMemcpy(pc, c.data(), 512, hipMemcpyHostToDevice);
      order<<<1, 64>>>(pa, pb, pc, pd); (void)hipDeviceSynchronize();
      (void)hipMemcpy(r.data(), pd, 512, hipMemcpyDeviceToHost);
      printf("c = 1, +2^53 at k = %d, -2^53 at k = %d: D = %g\n", kp, km, r[0]);
    }
  {  // fused or not: a*b with a = b = 1 + 2^-30: the product's low bits survive only under a fused multiply-add with c = -1 - 2^-29
    std::vector<double> a(64, 0.0), b(64, 0.0), c(64, 0.0), r(64);
    const double e = ldexp(1.0, -30);
    a[0] = 1 + e; b[0] = 1 + e; c[0] = -(1 + 2 * e);
    (void)hipMemcpy(pa, a.data(), 512, hipMemcpyHostToDevice); (void)hipMemcpy(pb, b.data(), 512, hipMemcpyHostToDevice);
    (void)hipMemcpy(pc, c.data(), 512, hipMemcpyHostToDevice);
    order<<<1, 64>>>(pa, pb, pc, pd); (void)hipDeviceSynchronize();
    (void)hipMemcpy(r.data(), pd, 512, hipMemcpyDeviceToHost);
    printf("(1 + 2^-30)^2 - (1 + 2^-29) = %g (2^-60 = %g if fused)\n", r[0], ldexp(1.0, -60));
  }
  run<0>("result -> C of the next", 0);
  run<1>("result -> A of the next", 0);
  run<2>("result -> B of the next", 0);
  run<3>("four independent accumulators", 0);
  run<4>("result -> v_fma_f64 -> A of the next", 1);
  run<5>("independent, 3 independent v_fma_f64 beside each", 0);
  run<6>("C-dependent, 3 independent v_fma_f64 beside each", 0);
  run<7>("independent, 6 independent v_fma_f64 beside each", 0);
  return 0;
}
