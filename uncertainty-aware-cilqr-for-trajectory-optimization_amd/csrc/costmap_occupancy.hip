// costmap_occupancy.hip — nav_msgs/OccupancyGrid (int8, -1 = unknown) <-> grid_map float32 layer (NaN = unknown), the
// wire formats either side of the costmap path: GridMapRosConverter::fromOccupancyGrid / toOccupancyGrid
// (G/grid_map_ros/src/GridMapRosConverter.cpp:225-269, :271-307) as the map node calls them (M/src/local_costmap.cpp:169
// for the global map coming in, :298 for the uncertainty map going out with range 0..100).
//
// Both directions reverse the cell order (occupancy cell k <-> layer linear index n-1-k).  HBM-bound byte work: 5 bytes
// per cell.  Groups of four cells: a 16-byte access on the float side; on the byte side the four reversed cells are one
// aligned 32-bit word when n is a multiple of 4 (the usual case), single cells otherwise.
#include <hip/hip_runtime.h>

#include "cilqr_internal.h"
#include "costmap_cells.hpp"

namespace cilqr {
namespace {

// Vector path: a lane moves groups of 4 cells (16 B on the float side, one 32-bit word on the byte side, so every wave
// instruction touches whole contiguous lines on both sides) and keeps UNR independent groups, a grid-width apart, in flight.
constexpr int CPL = 4, UNR = 4;

__device__ __forceinline__ float4 word_to_layer(uint32_t w) {  // a word of 4 reversed cells, high byte first
  float4 o;
  o.x = cell_to_layer((int8_t)(w >> 24));
  o.y = cell_to_layer((int8_t)(w >> 16));
  o.z = cell_to_layer((int8_t)(w >> 8));
  o.w = cell_to_layer((int8_t)w);
  return o;
}

__device__ __forceinline__ uint32_t layer_to_word(float4 v, float data_min, float den) {
  const uint32_t b0 = (uint8_t)layer_to_cell(v.x, data_min, den), b1 = (uint8_t)layer_to_cell(v.y, data_min, den);
  const uint32_t b2 = (uint8_t)layer_to_cell(v.z, data_min, den), b3 = (uint8_t)layer_to_cell(v.w, data_min, den);
  return (b0 << 24) | (b1 << 16) | (b2 << 8) | b3;
}

__global__ __launch_bounds__(256) void occ_to_layer_kernel(const int8_t* __restrict__ occ, float* __restrict__ layer, long n, int vec) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x, width = (long)gridDim.x * blockDim.x;
  if (vec) {  // n % 4 == 0, occ 4-byte and layer 16-byte aligned (checked by the launcher)
    const long groups = n / CPL;
    uint32_t w[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {  // layer[4g .. 4g+3] <- occ[n-1-4g .. n-4-4g]: one aligned word, bytes taken high to low
      const long g = t + u * width;
      if (g < groups) w[u] = *reinterpret_cast<const uint32_t*>(occ + (n - CPL - g * CPL));
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long g = t + u * width;
      if (g < groups) *reinterpret_cast<float4*>(layer + g * CPL) = word_to_layer(w[u]);
    }
  } else {
    for (int u = 0; u < UNR * CPL; ++u) {
      const long i = t + u * width;
      if (i < n) layer[i] = cell_to_layer(occ[n - 1 - i]);
    }
  }
}

__global__ __launch_bounds__(256) void layer_to_occ_kernel(const float* __restrict__ layer, int8_t* __restrict__ occ, long n,
                                                           float data_min, float data_max, int vec) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x, width = (long)gridDim.x * blockDim.x;
  const float den = __fsub_rn(data_max, data_min);
  if (vec) {
    const long groups = n / CPL;
    float4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long g = t + u * width;
      if (g < groups) v[u] = *reinterpret_cast<const float4*>(layer + g * CPL);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {  // occ[n-1-4g] = first cell of the group … occ[n-4-4g] = last
      const long g = t + u * width;
      if (g < groups) *reinterpret_cast<uint32_t*>(occ + (n - CPL - g * CPL)) = layer_to_word(v[u], data_min, den);
    }
  } else {
    for (int u = 0; u < UNR * CPL; ++u) {
      const long i = t + u * width;
      if (i < n) occ[n - 1 - i] = layer_to_cell(layer[i], data_min, den);
    }
  }
}

// ---- layer -> occupancy through the step table --------------------------------------------------------------------------
// For data_max > data_min the conversion is a non-decreasing step function of the cell value with at most 100 steps (every
// operation in layer_to_cell is monotone).  occ_steps_kernel finds, by bisection over the float ordering with layer_to_cell
// itself, the smallest value T[k] that converts to at least k (k = 1..100); a cell then needs a multiply-based estimate of k
// and two comparisons against the table instead of an IEEE division (≈ 25 → ≈ 13 instructions per cell; the division kept
// the exact kernel at 57 % of the HBM rate its sibling reaches).  Same results by construction, checked bit for bit.
constexpr int NSTEP = 102;  // T[0] = -inf, T[1..100], T[101] = NaN (never reached)

__device__ __forceinline__ uint32_t float_key(float f) {  // order-preserving map of non-NaN floats to unsigned
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__global__ void occ_steps_kernel(float data_min, float data_max, float* __restrict__ steps) {
  const int k = threadIdx.x;
  if (k >= NSTEP) return;
  if (k == 0) { steps[0] = -__builtin_inff(); return; }
  if (k == NSTEP - 1) { steps[k] = __builtin_nanf(""); return; }
  const float den = __fsub_rn(data_max, data_min);
  uint32_t lo = float_key(-__builtin_inff()), hi = float_key(__builtin_inff());  // f(lo) = 0 < k <= 100 = f(hi)
  while (hi - lo > 1) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if ((int)layer_to_cell(key_float(mid), data_min, den) >= k) hi = mid; else lo = mid;
  }
  steps[k] = key_float(hi);
}

__device__ __forceinline__ uint32_t steps_cell(float v, float data_min, float rden, const float* T) {
  if (v != v) return 0xffu;  // NaN -> -1
  float q = __fmul_rn(__fsub_rn(v, data_min), rden);
  q = q > 0.0f ? q : 0.0f;
  q = q < 1.0f ? q : 1.0f;
  const int k0 = (int)__fmul_rn(q, 100.0f);  // within one step of the answer
  return (uint32_t)(k0 + (v >= T[k0 + 1] ? 1 : 0) - (v < T[k0] ? 1 : 0));
}

__global__ __launch_bounds__(256) void layer_to_occ_steps_kernel(const float* __restrict__ layer, int8_t* __restrict__ occ, long n,
                                                                 float data_min, float data_max, const float* __restrict__ steps) {
  __shared__ float T[NSTEP];
  if (threadIdx.x < NSTEP) T[threadIdx.x] = steps[threadIdx.x];
  __syncthreads();
  const float rden = __fdiv_rn(1.0f, __fsub_rn(data_max, data_min));
  const long groups = n / CPL, width = (long)gridDim.x * blockDim.x;
  for (long g0 = (long)blockIdx.x * blockDim.x + threadIdx.x; g0 < groups; g0 += UNR * width) {
    float4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long g = g0 + u * width;
      if (g < groups) v[u] = *reinterpret_cast<const float4*>(layer + g * CPL);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long g = g0 + u * width;
      if (g < groups) {
        const uint32_t w = (steps_cell(v[u].x, data_min, rden, T) << 24) | (steps_cell(v[u].y, data_min, rden, T) << 16) |
                           (steps_cell(v[u].z, data_min, rden, T) << 8) | steps_cell(v[u].w, data_min, rden, T);
        *reinterpret_cast<uint32_t*>(occ + (n - CPL - g * CPL)) = w;
      }
    }
  }
}

}  // namespace

hipError_t launch_occ_to_layer(const int8_t* occ, float* layer, long n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const long lanes = (n + CPL * UNR - 1) / (CPL * UNR);
  const int vec = (n % CPL) == 0 && ((uintptr_t)occ & 3) == 0 && ((uintptr_t)layer & 15) == 0;
  hipLaunchKernelGGL(occ_to_layer_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, stream, occ, layer, n, vec);
  return hipGetLastError();
}

hipError_t launch_layer_to_occ(const float* layer, int8_t* occ, long n, float data_min, float data_max, float* steps_ws,
                               hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const long lanes = (n + CPL * UNR - 1) / (CPL * UNR);
  const int vec = (n % CPL) == 0 && ((uintptr_t)occ & 3) == 0 && ((uintptr_t)layer & 15) == 0;
  const float den = data_max - data_min;
  // step table: worth its 101-lane bisection launch from about a million cells; needs a positive finite range
  if (vec && steps_ws && n >= (1 << 20) && den >= 1.0e-30f && den <= 3.0e38f && data_min == data_min) {  // 1/den finite and normal
    hipLaunchKernelGGL(occ_steps_kernel, dim3(1), dim3(128), 0, stream, data_min, data_max, steps_ws);
    const long blocks = (lanes + 255) / 256;
    hipLaunchKernelGGL(layer_to_occ_steps_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, layer, occ,
                       n, data_min, data_max, steps_ws);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(layer_to_occ_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, stream, layer, occ, n, data_min,
                     data_max, vec);
  return hipGetLastError();
}

}  // namespace cilqr
