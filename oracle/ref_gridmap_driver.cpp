// ref_gridmap_driver.cpp — TEST INFRASTRUCTURE ONLY.
//
// Thin C entry points over the REFERENCE's own grid_map_core, compiled from the sources where they lie
// under /root/reference (see oracle/Makefile, target _ref/libref_gridmap.so).  Used to validate
// oracle/warp_oracle.c and to generate tests/golden/warp_*.json.  The loop body below is this driver's
// statement of the warp recipe M/src/local_costmap.cpp:242-264 over reference GridMap calls; nothing from
// the reference is copied into the repository.
#include <cmath>
#include <cstdint>
#include <stdexcept>

#include "grid_map_core/grid_map_core.hpp"

using grid_map::GridMap;
using grid_map::Index;
using grid_map::Length;
using grid_map::Position;

extern "C" {

// Returns rows/cols chosen by GridMap::setGeometry.
void ref_geometry(double lx, double ly, double res, double px, double py, int* rows, int* cols, double* out_lx,
                  double* out_ly) {
  GridMap m;
  m.setGeometry(Length(lx, ly), res, Position(px, py));
  *rows = m.getSize()(0);
  *cols = m.getSize()(1);
  *out_lx = m.getLength().x();
  *out_ly = m.getLength().y();
}

int ref_get_position(double lx, double ly, double res, double px, double py, int i, int j, double* ox, double* oy) {
  GridMap m;
  m.setGeometry(Length(lx, ly), res, Position(px, py));
  Position p;
  bool ok = m.getPosition(Index(i, j), p);
  *ox = p.x();
  *oy = p.y();
  return ok ? 1 : 0;
}

int ref_get_index(double lx, double ly, double res, double px, double py, double qx, double qy, int* i, int* j) {
  GridMap m;
  m.setGeometry(Length(lx, ly), res, Position(px, py));
  Index idx;
  bool ok = m.getIndex(Position(qx, qy), idx);
  *i = idx(0);
  *j = idx(1);
  return ok ? 1 : 0;
}

// src: rows×cols float32 column-major of a map with geometry (slx, sly, sres, spx, spy); dst likewise.
// Cells whose lookup throws std::out_of_range are written as NaN and counted.
long ref_warp(const float* src, double slx, double sly, double sres, double spx, double spy, float* dst, double dlx,
              double dly, double dres, double dpx, double dpy, double Vx, double Vy, double Vtheta,
              const float* bbox) {
  GridMap global_map, vehicle_map;
  global_map.add("global_map");
  global_map.setGeometry(Length(slx, sly), sres, Position(spx, spy));
  vehicle_map.add("vehicle_map");
  vehicle_map.add("bounding_box_map");
  vehicle_map.setGeometry(Length(dlx, dly), dres, Position(dpx, dpy));
  {
    grid_map::Matrix& g = global_map["global_map"];
    for (long k = 0; k < (long)g.size(); k++) g.data()[k] = src[k];
    grid_map::Matrix& b = vehicle_map["bounding_box_map"];
    for (long k = 0; k < (long)b.size(); k++) b.data()[k] = bbox ? bbox[k] : 0.0f;
  }
  const double s = std::sin(Vtheta), c = std::cos(Vtheta);
  long oob = 0;
  for (grid_map::GridMapIterator it(vehicle_map); !it.isPastEnd(); ++it) {
    Position position;
    vehicle_map.getPosition(*it, position);
    const double Cx = position.x(), Cy = position.y();
    const double x_og = (Cx * c - Cy * s) + Vx;
    const double y_og = (Cx * s + Cy * c) + Vy;
    try {
      vehicle_map.at("vehicle_map", *it) = global_map.atPosition("global_map", Position(x_og, y_og));
    } catch (const std::out_of_range&) {
      vehicle_map.at("vehicle_map", *it) = NAN;
      oob++;
    }
    if (vehicle_map.at("bounding_box_map", *it) > 90) {
      vehicle_map.at("vehicle_map", *it) = vehicle_map.at("bounding_box_map", *it);
    }
  }
  const grid_map::Matrix& v = vehicle_map["vehicle_map"];
  for (long k = 0; k < (long)v.size(); k++) dst[k] = v.data()[k];
  return oob;
}

}  // extern "C"
