// costmap_cells.hpp — per-cell value conversions between an OccupancyGrid cell and a grid_map layer cell, shared by the
// conversion kernels (costmap_occupancy.hip) and the blur kernel's fused occupancy output (costmap_blur.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cilqr {

// fromOccupancyGrid (G/grid_map_ros/src/GridMapRosConverter.cpp:265)
__device__ __forceinline__ float cell_to_layer(int8_t v) { return v != -1 ? (float)v : __builtin_nanf(""); }

// toOccupancyGrid's per-cell arithmetic (G/grid_map_ros/src/GridMapRosConverter.cpp:293-303): float throughout, IEEE
// division, no contraction; den = dataMax - dataMin (a float subtraction)
__device__ __forceinline__ int8_t layer_to_cell(float at, float data_min, float den) {
  float value = __fdiv_rn(__fsub_rn(at, data_min), den);
  if (value != value) return (int8_t)-1;
  const float lo = (0.0f < value) ? value : 0.0f;  // std::max(0.0f, value)
  const float hi = (1.0f < lo) ? 1.0f : lo;        // std::min(lo, 1.0f)
  value = __fadd_rn(0.0f, __fmul_rn(hi, 100.0f));  // cellMin + x * cellRange
  return (int8_t)value;                            // truncation, as the implicit float → int8_t conversion
}

}  // namespace cilqr
