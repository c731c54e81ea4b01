// costmap_occupancy.hip — nav_msgs/OccupancyGrid (int8, -1 = unknown) <-> grid_map float32 layer (NaN = unknown), the
// wire formats either side of the costmap path: GridMapRosConverter::fromOccupancyGrid / toOccupancyGrid
// (G/grid_map_ros/src/GridMapRosConverter.cpp:225-269, :271-307) as the map node calls them (M/src/local_costmap.cpp:169
// for the global map coming in, :298 for the uncertainty map going out with range 0..100).
//
// Both directions reverse the cell order (occupancy cell k <-> layer linear index n-1-k).  HBM-bound byte work: 5 bytes
// per cell.  Four cells per lane: a 16-byte access on the float side; on the byte side the four reversed cells are one
// aligned 32-bit word when n is a multiple of 4 (the usual case) and single bytes otherwise.
#include <hip/hip_runtime.h>

#include "cilqr_internal.h"
#include "costmap_cells.hpp"

namespace cilqr {
namespace {

__global__ __launch_bounds__(256) void occ_to_layer_kernel(const int8_t* __restrict__ occ, float* __restrict__ layer, long n, int vec) {
  const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= n) return;
  if (vec) {  // n % 4 == 0 and both pointers aligned (checked by the launcher)
    // layer[i4..i4+3] <- occ[n-1-i4 .. n-4-i4]: one aligned word, bytes taken high to low
    const uint32_t w = *reinterpret_cast<const uint32_t*>(occ + (n - 4 - i4));
    float4 o;
    o.x = cell_to_layer((int8_t)(w >> 24));
    o.y = cell_to_layer((int8_t)(w >> 16));
    o.z = cell_to_layer((int8_t)(w >> 8));
    o.w = cell_to_layer((int8_t)w);
    *reinterpret_cast<float4*>(layer + i4) = o;
  } else {
    for (long i = i4; i < n && i < i4 + 4; ++i) layer[i] = cell_to_layer(occ[n - 1 - i]);
  }
}

__global__ __launch_bounds__(256) void layer_to_occ_kernel(const float* __restrict__ layer, int8_t* __restrict__ occ, long n,
                                                           float data_min, float data_max, int vec) {
  const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= n) return;
  const float den = __fsub_rn(data_max, data_min);
  if (vec) {  // n % 4 == 0 and both pointers aligned (checked by the launcher)
    const float4 v = *reinterpret_cast<const float4*>(layer + i4);
    const uint32_t b0 = (uint8_t)layer_to_cell(v.x, data_min, den), b1 = (uint8_t)layer_to_cell(v.y, data_min, den);
    const uint32_t b2 = (uint8_t)layer_to_cell(v.z, data_min, den), b3 = (uint8_t)layer_to_cell(v.w, data_min, den);
    // occ[n-1-i4] = b0 … occ[n-4-i4] = b3
    *reinterpret_cast<uint32_t*>(occ + (n - 4 - i4)) = (b0 << 24) | (b1 << 16) | (b2 << 8) | b3;
  } else {
    for (long i = i4; i < n && i < i4 + 4; ++i) occ[n - 1 - i] = layer_to_cell(layer[i], data_min, den);
  }
}

}  // namespace

hipError_t launch_occ_to_layer(const int8_t* occ, float* layer, long n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const long lanes = (n + 3) / 4;
  const int vec = (n & 3) == 0 && ((uintptr_t)occ & 3) == 0 && ((uintptr_t)layer & 15) == 0;
  hipLaunchKernelGGL(occ_to_layer_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, stream, occ, layer, n, vec);
  return hipGetLastError();
}

hipError_t launch_layer_to_occ(const float* layer, int8_t* occ, long n, float data_min, float data_max, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const long lanes = (n + 3) / 4;
  const int vec = (n & 3) == 0 && ((uintptr_t)occ & 3) == 0 && ((uintptr_t)layer & 15) == 0;
  hipLaunchKernelGGL(layer_to_occ_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, stream, layer, occ, n, data_min,
                     data_max, vec);
  return hipGetLastError();
}

}  // namespace cilqr
