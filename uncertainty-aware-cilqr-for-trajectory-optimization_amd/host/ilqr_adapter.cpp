// ilqr_adapter.cpp — see ilqr_adapter.h.  Pure host C++ over the C-ABI (no HIP here).
#include "ilqr_adapter.h"

#include <stdexcept>
#include <string>

namespace cilqr_host {

namespace {
void check(int rc, const char* what) {
  if (rc != CILQR_OK) throw std::runtime_error(std::string(what) + ": " + cilqr_last_error());
}
}  // namespace

Parameters default_parameters() {
  Parameters p;
  cilqr_params_default(&p);
  return p;
}

iLQR::iLQR(const Parameters& params, int device, int max_obstacles, int max_candidates)
    : params(params), device_(device), max_obstacles_(max_obstacles), max_candidates_(max_candidates < 1 ? 1 : max_candidates) {
  const int N = params.horizon;
  check(cilqr_create(&params, max_candidates_, N, max_obstacles_, device_, &h_), "cilqr_create");
  control_seq_ = Matrix(params.num_ctrls, N);  // I/iLQR.cpp:9-15
  check(cilqr_default_control_seq(N, control_seq_.a.data()), "cilqr_default_control_seq");
}

iLQR::~iLQR() { cilqr_destroy(h_); }

void iLQR::set_Obstacle(const std::vector<Obstacle>& obstacles) {
  if ((int)obstacles.size() > max_obstacles_) throw std::runtime_error("set_Obstacle: more obstacles than max_obstacles");
  for (const Obstacle& o : obstacles)
    if (o.dimension.rows != 2 || o.relative_pos_array.rows != 4 || o.dimension.cols < params.horizon ||
        o.relative_pos_array.cols < params.horizon)
      throw std::runtime_error("set_Obstacle: dimension must be 2×horizon and relative_pos_array 4×horizon");
  obstacles_ = obstacles;
  obs_pose_.clear();
  obs_dim_.clear();
}

void iLQR::clear_Obstacle() {
  obstacles_.clear();
  obs_pose_.clear();
  obs_dim_.clear();
}

void iLQR::set_uncertainty_map(const Uncertainty& u) {
  if (u.layer.size() != (size_t)u.geom.rows * u.geom.cols) throw std::runtime_error("set_uncertainty_map: layer size does not match its geometry");
  cilqr_uncertainty_map m{};
  m.layer = u.layer.data();
  m.geom = u.geom;
  m.pose_x = u.pose_x; m.pose_y = u.pose_y; m.pose_theta = u.pose_theta;
  m.poses = nullptr;
  m.layer_stride = 0;
  m.probes_l = u.probes_l; m.probes_w = u.probes_w;
  check(cilqr_set_uncertainty_map(h_, &m), "cilqr_set_uncertainty_map");
}

void iLQR::clear_uncertainty_map() { check(cilqr_clear_uncertainty_map(h_), "cilqr_clear_uncertainty_map"); }

void iLQR::set_global_plan(const Matrix& global_plan) {
  if (global_plan.rows != 2 || global_plan.cols < 1) throw std::runtime_error("set_global_plan: expected a 2×P matrix");
  global_plan_ = global_plan;
}

void iLQR::pack_obstacles(int copies) {
  const int N = params.horizon, M = (int)obstacles_.size();
  if (obs_pose_.size() == (size_t)copies * M * 4 * N) return;
  obs_pose_.resize((size_t)copies * M * 4 * N);
  obs_dim_.resize((size_t)copies * M * 2 * N);
  for (int c = 0; c < copies; ++c)
    for (int m = 0; m < M; ++m) {
      const Obstacle& o = obstacles_[m];
      for (int t = 0; t < N; ++t) {
        for (int r = 0; r < 4; ++r) obs_pose_[(((size_t)c * M + m) * N + t) * 4 + r] = o.relative_pos_array(r, t);
        for (int r = 0; r < 2; ++r) obs_dim_[(((size_t)c * M + m) * N + t) * 2 + r] = o.dimension(r, t);
      }
    }
}

void iLQR::get_optimal_control_seq(const double x_0[4], Matrix& U, const double poly_coeffs[6],
                                   const std::vector<double>& x_local_plan) {
  const int N = params.horizon, M = (int)obstacles_.size();
  if (U.rows != 2 || U.cols != N) throw std::runtime_error("get_optimal_control_seq: U must be 2×horizon");
  if (x_local_plan.empty()) throw std::runtime_error("get_optimal_control_seq: empty x_local_plan");
  pack_obstacles(1);
  const double fl[2] = {x_local_plan.front(), x_local_plan.back()};
  X_result = Matrix(4, N + 1);
  int32_t iters = 0, status = 0;
  check(cilqr_solve_batch(h_, 1, N, M, x_0, U.a.data(), poly_coeffs, fl, M ? obs_pose_.data() : nullptr,
                          M ? obs_dim_.data() : nullptr, nullptr, X_result.a.data(), &last_cost, &iters, &status, CILQR_FLAG_NONE),
        "cilqr_solve_batch");
  last_iterations = iters;
  last_exit = status;
  U_result = U;  // I/iLQR.cpp:244
}

void iLQR::run_step(const double ego_state[4]) {
  if (global_plan_.cols < 1) throw std::runtime_error("run_step: set_global_plan was not called");
  double coeffs[CILQR_POLY_COEFFS];
  std::vector<double> ref(2 * (size_t)params.num_of_local_wpts);
  int n = 0;
  check(cilqr_local_plan(&params, global_plan_.a.data(), global_plan_.cols, ego_state, coeffs, ref.data(), &n), "cilqr_local_plan");
  std::vector<double> x_local_plan(n);
  ref_traj_result = Matrix(2, n);
  for (int i = 0; i < n; ++i) {
    x_local_plan[i] = ref[2 * i];
    ref_traj_result(0, i) = ref[2 * i];
    ref_traj_result(1, i) = ref[2 * i + 1];
  }
  get_optimal_control_seq(ego_state, control_seq_, coeffs, x_local_plan);  // I/iLQR.cpp:253: warm start persists
}

Experiment flatten_experiment(const double start_pos[4], double planning_time, const Matrix& X, const Matrix& U) {
  Experiment e;
  e.planning_time = planning_time;
  e.start_pos.assign(start_pos, start_pos + 4);
  e.X.resize(4 * (size_t)(U.cols + 1));
  e.U.resize(2 * (size_t)U.cols);
  for (int i = 0; i < U.cols + 1; ++i)
    for (int r = 0; r < 4; ++r) e.X[4 * i + r] = X(r, i);
  for (int i = 0; i < U.cols; ++i)
    for (int r = 0; r < 2; ++r) e.U[2 * i + r] = U(r, i);
  return e;
}

int iLQR::run_candidates(const std::vector<double>& ego_states) {
  const int B = (int)(ego_states.size() / 4), N = params.horizon, M = (int)obstacles_.size();
  if (B < 1 || B > max_candidates_) throw std::runtime_error("run_candidates: candidate count outside [1, max_candidates]");
  if (global_plan_.cols < 1) throw std::runtime_error("run_candidates: set_global_plan was not called");
  std::vector<double> U((size_t)B * 2 * N), poly((size_t)B * CILQR_POLY_COEFFS), fl((size_t)B * 2);
  std::vector<double> X((size_t)B * 4 * (N + 1)), J(B);
  std::vector<int32_t> iters(B), status(B);
  // LocalPlanner pre-step for all candidates in one device launch (cilqr_local_plan_batch) instead of B host fits
  const int W = params.num_of_local_wpts;
  std::vector<double> ref((size_t)B * 2 * W);
  std::vector<int32_t> n_ref(B);
  check(cilqr_local_plan_batch(h_, B, global_plan_.cols, global_plan_.a.data(), 0, ego_states.data(), poly.data(), fl.data(),
                               ref.data(), n_ref.data()), "cilqr_local_plan_batch");
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < 2 * N; ++i) U[(size_t)b * 2 * N + i] = control_seq_.a[i];
  pack_obstacles(B);
  check(cilqr_solve_batch(h_, B, N, M, ego_states.data(), U.data(), poly.data(), fl.data(), M ? obs_pose_.data() : nullptr,
                          M ? obs_dim_.data() : nullptr, nullptr, X.data(), J.data(), iters.data(), status.data(), CILQR_FLAG_NONE),
        "cilqr_solve_batch");
  int best = 0;  // strict-< first minimum, NaN never wins (the convention of cilqr_argmin_device)
  bool have = false;
  for (int b = 0; b < B; ++b)
    if (J[b] == J[b] && (!have || J[b] < J[best])) { best = b; have = true; }
  X_result = Matrix(4, N + 1);
  for (int i = 0; i < 4 * (N + 1); ++i) X_result.a[i] = X[(size_t)best * 4 * (N + 1) + i];
  for (int i = 0; i < 2 * N; ++i) control_seq_.a[i] = U[(size_t)best * 2 * N + i];
  U_result = control_seq_;
  ref_traj_result = Matrix(2, n_ref[best]);
  for (int i = 0; i < 2 * n_ref[best]; ++i) ref_traj_result.a[i] = ref[(size_t)best * 2 * W + i];
  last_iterations = iters[best];
  last_exit = status[best];
  last_cost = J[best];
  return best;
}

}  // namespace cilqr_host
