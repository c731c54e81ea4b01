// local_planner.cpp — host pre-step of the solve: the reference's LocalPlanner (I/LocalPlanner.cpp:25-117).
// Stays on the host (≈0.1 % of a planning tick): nearest global waypoint, a slice of at most
// num_of_local_wpts waypoints, and a degree-poly_order least-squares fit y(x).
//
// The reference solves the Vandermonde system with Eigen's `colPivHouseholderQr().solve(y)` (:114).  The same
// algorithm is written out here: Householder QR with column pivoting on the largest remaining column norm, rank
// cut-off at max-column-norm²·eps²/rows scaled by the remaining rows, Qᵀy, back-substitution on the leading
// nonzero-pivot block, zeros for the cut columns.
#include <math.h>

#include <cfloat>
#include <vector>

#include "cilqr.h"

namespace {

struct ColMajor {
  int rows, cols;
  std::vector<double> a;
  ColMajor(int r, int c) : rows(r), cols(c), a((size_t)r * c, 0.0) {}
  double& operator()(int i, int j) { return a[(size_t)j * rows + i]; }
  double operator()(int i, int j) const { return a[(size_t)j * rows + i]; }
};

double tail_sq_norm(const ColMajor& m, int col, int from) {
  double s = 0.0;
  for (int i = from; i < m.rows; ++i) s += m(i, col) * m(i, col);
  return s;
}

// Least squares min |V c - y| with V(i,j) = x_i^j, j = 0..degree.
void vandermonde_lstsq(const std::vector<double>& x, const std::vector<double>& y, int degree, double* coeffs) {
  const int rows = (int)x.size(), cols = degree + 1;
  const int diag = rows < cols ? rows : cols;
  ColMajor qr(rows, cols);
  for (int i = 0; i < rows; ++i)
    for (int j = 0; j < cols; ++j) qr(i, j) = pow(x[i], (double)j);

  std::vector<double> col_norm(cols), tau(diag, 0.0);
  std::vector<int> swap_with(diag, 0), order(cols);
  double max_norm = 0.0;
  for (int j = 0; j < cols; ++j) {
    col_norm[j] = tail_sq_norm(qr, j, 0);
    if (j == 0 || col_norm[j] > max_norm) max_norm = col_norm[j];
  }
  const double cut = max_norm * (DBL_EPSILON * DBL_EPSILON) / (double)rows;
  int rank = diag;
  for (int k = 0; k < diag; ++k) {
    int pivot = k;
    for (int j = k + 1; j < cols; ++j)
      if (col_norm[j] > col_norm[pivot]) pivot = j;
    const double exact = tail_sq_norm(qr, pivot, k);
    col_norm[pivot] = exact;
    if (rank == diag && exact < cut * (double)(rows - k)) rank = k;
    swap_with[k] = pivot;
    if (pivot != k) {
      for (int i = 0; i < rows; ++i) std::swap(qr(i, k), qr(i, pivot));
      std::swap(col_norm[k], col_norm[pivot]);
    }
    // Householder vector for column k (stored below the diagonal, unit leading entry implied)
    const double below = tail_sq_norm(qr, k, k + 1);
    const double head = qr(k, k);
    double beta;
    if (below == 0.0) {
      tau[k] = 0.0;
      beta = head;
      for (int i = k + 1; i < rows; ++i) qr(i, k) = 0.0;
    } else {
      beta = sqrt(head * head + below);
      if (head >= 0.0) beta = -beta;
      for (int i = k + 1; i < rows; ++i) qr(i, k) = qr(i, k) / (head - beta);
      tau[k] = (beta - head) / beta;
    }
    qr(k, k) = beta;
    // reflect the trailing columns
    for (int j = k + 1; j < cols; ++j) {
      if (rows - k == 1) {
        qr(k, j) *= (1 - tau[k]);
      } else {
        double dot = 0.0;
        for (int i = k + 1; i < rows; ++i) dot += qr(i, k) * qr(i, j);
        dot += qr(k, j);
        qr(k, j) -= tau[k] * dot;
        for (int i = k + 1; i < rows; ++i) qr(i, j) -= tau[k] * qr(i, k) * dot;
      }
      col_norm[j] -= qr(k, j) * qr(k, j);
    }
  }
  for (int j = 0; j < cols; ++j) order[j] = j;
  for (int k = 0; k < diag; ++k) std::swap(order[k], order[swap_with[k]]);

  for (int j = 0; j < cols; ++j) coeffs[j] = 0.0;
  if (rank == 0) return;
  std::vector<double> c(y);
  for (int k = 0; k < rank; ++k) {  // c ← H_k c
    if (rows - k == 1) { c[k] *= (1 - tau[k]); continue; }
    double dot = 0.0;
    for (int i = k + 1; i < rows; ++i) dot += qr(i, k) * c[i];
    dot += c[k];
    c[k] -= tau[k] * dot;
    for (int i = k + 1; i < rows; ++i) c[i] -= tau[k] * qr(i, k) * dot;
  }
  for (int i = rank - 1; i >= 0; --i) {
    double s = c[i];
    for (int j = i + 1; j < rank; ++j) s -= qr(i, j) * c[j];
    c[i] = s / qr(i, i);
  }
  for (int i = 0; i < rank; ++i) coeffs[order[i]] = c[i];
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
