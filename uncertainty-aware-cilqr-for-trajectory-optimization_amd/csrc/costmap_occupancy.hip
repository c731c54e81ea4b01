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

}  // namespace

hipError_t launch_occ_to_layer(const int8_t* occ, float* layer, long n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const long lanes = (n + CPL * UNR - 1) / (CPL * UNR);
  const int vec = (n % CPL) == 0 && ((uintptr_t)occ & 3) == 0 && ((uintptr_t)layer & 15) == 0;
  hipLaunchKernelGGL(occ_to_layer_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, stream, occ, layer, n, vec);
  return hipGetLastError();
}

hipError_t launch_layer_to_occ(const float* layer, int8_t* occ, long n, float data_min, float data_max, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const long lanes = (n + CPL * UNR - 1) / (CPL * UNR);
  const int vec = (n % CPL) == 0 && ((uintptr_t)occ & 3) == 0 && ((uintptr_t)layer & 15) == 0;
  hipLaunchKernelGGL(layer_to_occ_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, stream, layer, occ, n, data_min,
                     data_max, vec);
  return hipGetLastError();
}

}  // namespace cilqr
