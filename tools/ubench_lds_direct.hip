// Checks, on gfx950, the whole-wave direct global->LDS copy used by the grouped solve kernel: lane l of the (temporarily
// fully enabled) wavefront moves 16 bytes to LDS base + 16 l, also when called from divergent code.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_lds_direct.hip -o /tmp/t && /tmp/t
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void stage_piece(const void* src, unsigned lds_byte, unsigned long long mask) {
  unsigned long long save;
  unsigned vtmp;
  asm volatile(
      "s_mov_b64 %0, exec\n\t"
      "s_mov_b64 exec, %3\n\t"
      "v_mbcnt_lo_u32_b32 %1, -1, 0\n\t"
      "v_mbcnt_hi_u32_b32 %1, -1, %1\n\t"
      "v_lshlrev_b32 %1, 4, %1\n\t"
      "s_mov_b32 m0, %4\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b64 exec, %0"
      : "=&s"(save), "=&v"(vtmp)
      : "s"(src), "s"(mask), "s"(lds_byte)
      : "memory", "m0");
}
__global__ void k(const double* __restrict__ g, double* out) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if (threadIdx.x % 3 == 0) {   // divergent: only a third of the lanes are active here
    stage_piece(g, __builtin_amdgcn_groupstaticsize(), ~0ull);
    stage_piece(g + 128, __builtin_amdgcn_groupstaticsize() + 1024, (1ull << 40) - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<double> h(256);
  for (int i = 0; i < 256; ++i) h[i] = 1000.0 + i;
  double *g, *o;
  (void)hipMalloc(&g, 256 * 8); (void)hipMalloc(&o, 256 * 8);
  (void)hipMemcpy(g, h.data(), 256 * 8, hipMemcpyHostToDevice);
  (void)hipMemset(o, 0, 256 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, g, o);
  std::vector<double> r(256);
  (void)hipMemcpy(r.data(), o, 256 * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 128 + 80; ++i) bad += r[i] != h[i];
  printf("whole-wave direct-to-LDS copy from divergent code: %d mismatches in %d doubles\n", bad, 128 + 80);
  return bad != 0;
}
