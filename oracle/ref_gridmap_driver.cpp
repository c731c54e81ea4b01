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

// GridMap::atPosition(layer, position, INTER_LINEAR) (G/grid_map_core/src/GridMap.cpp:191-201 → :770-837) at n positions of a
// layer given as rows×cols float32 column-major.  ok[k] = 0 where the call throws.
void ref_linear(const float* src, double lx, double ly, double res, double px, double py, int n, const double* qx,
                const double* qy, float* out, int* ok) {
  GridMap m;
  m.add("layer");
  m.setGeometry(Length(lx, ly), res, Position(px, py));
  grid_map::Matrix& g = m["layer"];
  for (long k = 0; k < (long)g.size(); k++) g.data()[k] = src[k];
  for (int k = 0; k < n; k++) {
    try {
      out[k] = m.atPosition("layer", Position(qx[k], qy[k]), grid_map::InterpolationMethods::INTER_LINEAR);
      ok[k] = 1;
    } catch (const std::out_of_range&) {
      out[k] = NAN;
      ok[k] = 0;
    }
  }
}

}  // extern "C"

// ---- uncertainty propagation ("blur"), M/src/arbitrary_transformation.cu:8-157 + M/include/ARBIT.cuh:51-107 ------------
// This driver's statement of the recipe over the reference's own grid_map_core (EllipseIterator, GridMap) and the
// reference's vendored Eigen (EigenSolver<Matrix2f>).  The three thrust functors are a dozen arithmetic lines; they are
// restated here because ARBIT.cuh cannot be included without CUDA/thrust/OpenCV/ROS headers.
#include <Eigen/Eigenvalues>

#include "grid_map_core/iterators/EllipseIterator.hpp"

extern "C" void ref_blur(const float* src, double lx, double ly, double res, double px, double py, int index, double sin_t,
                         double cos_t, double sigma_x, double sigma_y, double sigma_theta, float* out, int* count_out,
                         double* ellipse_out /* 3 per cell or null */) {
  GridMap m;
  m.add("vehicle_map");
  m.add("uncertainty_map");
  m.setGeometry(Length(lx, ly), res, Position(px, py));
  {
    grid_map::Matrix& g = m["vehicle_map"];
    for (long k = 0; k < (long)g.size(); k++) g.data()[k] = src[k];
    grid_map::Matrix& u = m["uncertainty_map"];
    for (long k = 0; k < (long)u.size(); k++) u.data()[k] = NAN;
  }
  grid_map::GridMapIterator it(m);
  for (int i = 0; i < index; ++i) ++it;
  long lin = index;
  for (; !it.isPastEnd(); ++it, ++lin) {
    Position position;
    m.getPosition(*it, position);
    const double Cx = position.x(), Cy = position.y();
    // uncertainty_error_functor (ARBIT.cuh:59-68)
    const double u = (-sin_t * Cx - cos_t * Cy) * (-sin_t * Cx - cos_t * Cy);
    const double v = (cos_t * Cx - sin_t * Cy) * (cos_t * Cx - sin_t * Cy);
    const double t = sin_t * cos_t * (Cx * Cx - Cy * Cy) + Cx * Cy * (sin_t * sin_t - cos_t * cos_t);
    const double sigma_x_i = sqrt(sigma_x * sigma_x + sigma_theta * sigma_theta * u);
    const double sigma_y_i = sqrt(sigma_y * sigma_y + sigma_theta * sigma_theta * v);
    const double rho = sigma_theta * sigma_theta * t / (sigma_x_i * sigma_y_i);
    // abc_functor (:74-79)
    const double a = sigma_x_i * sigma_x_i, b = rho * sigma_x_i * sigma_y_i, c = sigma_y_i * sigma_y_i;
    // host loop arbitrary_transformation.cu:60-83
    Eigen::Matrix2f cov_i;
    double temp0 = a, temp1 = b, temp2 = c;
    cov_i << temp0, temp1, temp1, temp2;
    Eigen::EigenSolver<Eigen::Matrix2f> es(cov_i);
    Eigen::Matrix2f D = es.pseudoEigenvalueMatrix();
    Eigen::Matrix2f V = es.pseudoEigenvectors();
    int major_index, minor_index;
    if (D(0, 0) > D(1, 1)) { major_index = 0; minor_index = 1; } else { major_index = 1; minor_index = 0; }
    // ellipse_params_functor (ARBIT.cuh:85-97)
    const double chisquare_val = 2.4477;
    double angle = atan2(V(major_index, 1), V(major_index, 0));
    if (angle < 0) angle += 6.28318530718;
    const double half_major_axis = chisquare_val * sqrt(D(major_index, major_index));
    const double half_minor_axis = chisquare_val * sqrt(D(minor_index, minor_index));
    if (ellipse_out) { ellipse_out[3 * lin] = half_major_axis; ellipse_out[3 * lin + 1] = half_minor_axis; ellipse_out[3 * lin + 2] = angle; }
    // OpenMP loop body arbitrary_transformation.cu:104-138
    double numerator = 0, denominator = 0;
    int count = 0;
    for (grid_map::EllipseIterator iterator(m, position, Length(2 * half_major_axis, 2 * half_minor_axis), angle);
         !iterator.isPastEnd(); ++iterator) {
      Position position_j;
      m.getPosition(*iterator, position_j);
      const double x = position_j.x(), y = position_j.y(), mu1 = Cx, mu2 = Cy, sigma1 = sigma_x_i, sigma2 = sigma_y_i;
      // nomal2 (ARBIT.cuh:103-107)
      const double f_i = 1.0 / (sqrt(1 - rho * rho) * (2 * M_PI * sigma1 * sigma2)) *
                         exp((-1 / (2 * (1 - rho * rho))) * ((x - mu1) * (x - mu1) / (sigma1 * sigma1) -
                                                           2 * rho * (x - mu1) * (y - mu2) / (sigma1 * sigma2) +
                                                           (y - mu2) * (y - mu2) / (sigma2 * sigma2)));
      numerator += f_i * m.atPosition("vehicle_map", position_j);
      denominator += f_i;
      count++;
    }
    const double result = numerator / denominator;
    // LocalCostmap::propagateUncertainty (M/src/local_costmap.cpp:483-496)
    if (count == 0) m.at("uncertainty_map", *it) = m.at("vehicle_map", *it);
    else m.at("uncertainty_map", *it) = result;
    if (count_out) count_out[lin] = count;
  }
  const grid_map::Matrix& u = m["uncertainty_map"];
  for (long k = 0; k < (long)u.size(); k++) out[k] = u.data()[k];
}
