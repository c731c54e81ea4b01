// local_planner.cpp — host pre-step of the solve: the reference's LocalPlanner (I/LocalPlanner.cpp:25-117).
// Stays on the host (≈0.1 % of a planning tick): nearest global waypoint, a slice of at most
// num_of_local_wpts waypoints, and a degree-poly_order least-squares fit y(x).
//
// The fit itself (Eigen's `colPivHouseholderQr().solve(y)`, :114, written out) lives in vandermonde_qr.hpp, shared with
// the batched device pre-step (local_plan.hip).
#include <math.h>

#include <vector>

#include "cilqr.h"
#include "vandermonde_qr.hpp"

namespace {

struct HostStore {
  int rows;
  std::vector<double> a, y, norms, taus;
  std::vector<int> swaps, orders;
  HostStore(int r, int cols) : rows(r), a((size_t)r * cols, 0.0), y(r), norms(cols), taus(cols), swaps(cols), orders(cols) {}
  double& m(int i, int j) { return a[(size_t)j * rows + i]; }
  double& c(int i) { return y[i]; }
  double& col_norm(int j) { return norms[j]; }
  double& tau(int k) { return taus[k]; }
  int& swap_with(int k) { return swaps[k]; }
  int& order(int j) { return orders[j]; }
};

// Least squares min |V c - y| with V(i,j) = x_i^j, j = 0..degree.
void vandermonde_lstsq(const std::vector<double>& x, const std::vector<double>& y, int degree, double* coeffs) {
  const int rows = (int)x.size(), cols = degree + 1;
  HostStore s(rows, cols);
  for (int i = 0; i < rows; ++i) {
    for (int j = 0; j < cols; ++j) s.m(i, j) = pow(x[i], (double)j);
    s.c(i) = y[i];
  }
  cilqr::vandermonde_lstsq<0>(s, rows, cols, coeffs);
}

}  // namespace

extern "C" int cilqr_local_plan(const cilqr_params* p, const double* path, int P, const double* ego_state,
                                double* coeffs, double* ref_traj, int* n_out) {
  if (!p || !path || !ego_state || !coeffs || P < 1 || p->num_of_local_wpts < 1 || p->poly_order < 0) return CILQR_ERR_ARG;
  // closest_point_index (:25-41): strict-< first minimum over squared distance
  double best = pow(ego_state[0] - path[0], 2) + pow(ego_state[1] - path[1], 2);
  int first = 0;
  for (int i = 0; i < P; ++i) {
    const double d = pow(ego_state[0] - path[2 * i], 2) + pow(ego_state[1] - path[2 * i + 1], 2);
    if (d < best) { best = d; first = i; }
  }
  // get_local_wpts (:47-60)
  const int n = (P - first) < p->num_of_local_wpts ? (P - first) : p->num_of_local_wpts;
  std::vector<double> xs(n), ys(n);
  for (int i = 0; i < n; ++i) {
    xs[i] = path[2 * (first + i)];
    ys[i] = path[2 * (first + i) + 1];
  }
  vandermonde_lstsq(xs, ys, p->poly_order, coeffs);  // polyfit (:101-117)
  if (ref_traj) {  // get_local_plan (:66-85)
    for (int i = 0; i < n; ++i) {
      double fy = 0.0;
      for (int j = 0; j < p->poly_order + 1; ++j) fy += coeffs[j] * pow(xs[i], (double)j);
      ref_traj[2 * i] = xs[i];
      ref_traj[2 * i + 1] = fy;
    }
  }
  if (n_out) *n_out = n;
  return CILQR_OK;
}
