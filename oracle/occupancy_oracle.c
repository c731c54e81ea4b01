/*
 * occupancy_oracle.c — CPU restatement of the reference's OccupancyGrid <-> grid_map layer conversion.
 * TEST INFRASTRUCTURE ONLY (see cilqr_oracle.h).
 *
 * Reference: G/grid_map_ros/src/GridMapRosConverter.cpp:225-269 (fromOccupancyGrid) and :271-307 (toOccupancyGrid), as
 * called by the map node at M/src/local_costmap.cpp:169 (global map in) and :298 (uncertainty map out, range 0..100).
 * G/ = CILQR/src/grid_map/, M/ = CILQR/src/map_engine/.
 *
 * Pinning: GridMapRosConverter.cpp includes ROS message headers that this image does not have, so it cannot be built into
 * oracle/_ref.  The restatement is pinned by the properties the reference's own gtests assert
 * (G/grid_map_ros/test/GridMapRosTest.cpp:116-127 "withMove", first half: a layer of 1.0 with range (0,1) gives 100;
 * :140-184 "roundTrip": every int8 in [-1,100] survives from→to with range (-1,100)), restated in tests/test_oracle.py.
 * Only maps whose circular-buffer start index is zero are covered (the node re-creates the vehicle map with setGeometry
 * every frame, M/src/local_costmap.cpp:212, and never moves the global map).
 */
#include <math.h>
#include <stdint.h>

#include "cilqr_oracle.h"

/* :259-266 — reverse iteration, data(i) = *it != -1 ? *it : NAN, i = linear (column-major) index */
void oracle_occupancy_to_layer(const int8_t* occ, long n, float* layer) {
  for (long i = 0; i < n; ++i) {
    const int8_t v = occ[n - 1 - i];
    layer[i] = v != -1 ? (float)v : NAN;
  }
}

/* :293-306 — float arithmetic throughout; std::max(0.0f, v) / std::min(., 1.0f) written out; float → int8 truncates */
void oracle_layer_to_occupancy(const float* layer, long n, float data_min, float data_max, int8_t* occ) {
  const float cell_min = 0, cell_max = 100, cell_range = cell_max - cell_min;
  for (long i = 0; i < n; ++i) {
    float value = (layer[i] - data_min) / (data_max - data_min);
    if (isnan(value)) {
      value = -1;
    } else {
      const float lo = (0.0f < value) ? value : 0.0f;  /* std::max(0.0f, value) */
      const float hi = (1.0f < lo) ? 1.0f : lo;        /* std::min(lo, 1.0f)    */
      value = cell_min + hi * cell_range;
    }
    occ[n - i - 1] = (int8_t)value;
  }
}
