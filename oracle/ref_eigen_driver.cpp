// ref_eigen_driver.cpp — TEST INFRASTRUCTURE ONLY.
//
// C entry points over the Eigen 3.2.10 that the reference vendors (M/include/map_engine/Eigen), compiled
// from where it lies under /root/reference (oracle/Makefile, target _ref/libref_eigen.so).  These are the
// two library algorithms on the solver path:
//   Eigen::EigenSolver<MatrixXd> on Q_uu            — call site I/iLQR.cpp:155-175
//   MatrixXd::colPivHouseholderQr().solve(y)        — call site I/LocalPlanner.cpp:101-117
// Used to validate oracle_quu_inverse / oracle_polyfit and to generate tests/golden/eigen_*.json.
#include <Eigen/Dense>
#include <cmath>

extern "C" {

// Statement of I/iLQR.cpp:155-175 over the reference's Eigen.  Column-major 2×2 in, 2×2 out.
int ref_quu_inverse(const double* Quu, double lamb, double* Qinv, double* eval2, double* evec4) {
  Eigen::MatrixXd Q_uu(2, 2);
  Q_uu << Quu[0], Quu[2], Quu[1], Quu[3];
  Eigen::EigenSolver<Eigen::MatrixXd> matrix_solver(Q_uu);
  matrix_solver.compute(Q_uu);
  if (matrix_solver.info() != Eigen::Success) return -1;
  Eigen::VectorXd Q_uu_val = matrix_solver.eigenvalues().real();
  Eigen::MatrixXd Q_uu_vec = matrix_solver.eigenvectors().real();
  if (eval2) { eval2[0] = Q_uu_val(0); eval2[1] = Q_uu_val(1); }
  if (evec4) { evec4[0] = Q_uu_vec(0, 0); evec4[1] = Q_uu_vec(1, 0); evec4[2] = Q_uu_vec(0, 1); evec4[3] = Q_uu_vec(1, 1); }
  Q_uu_val = Q_uu_val.array().max(0);
  Q_uu_val = Q_uu_val.array() + lamb;
  Q_uu_val = Q_uu_val.array().inverse();
  Eigen::MatrixXd D = Q_uu_val.asDiagonal();
  Eigen::MatrixXd Q_uu_inv = Q_uu_vec * (D * Q_uu_vec.transpose());
  Qinv[0] = Q_uu_inv(0, 0);
  Qinv[1] = Q_uu_inv(1, 0);
  Qinv[2] = Q_uu_inv(0, 1);
  Qinv[3] = Q_uu_inv(1, 1);
  return 0;
}

// Statement of I/LocalPlanner.cpp:101-117 over the reference's Eigen.
void ref_polyfit(const double* x, const double* y, int n, int degree, double* coeffs) {
  Eigen::MatrixXd X = Eigen::MatrixXd::Zero(n, degree + 1);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= degree; ++j) X(i, j) = std::pow(x[i], j);
  Eigen::VectorXd yy(n);
  for (int i = 0; i < n; ++i) yy(i) = y[i];
  Eigen::VectorXd c = X.colPivHouseholderQr().solve(yy);
  for (int j = 0; j <= degree; ++j) coeffs[j] = c(j);
}

}  // extern "C"
