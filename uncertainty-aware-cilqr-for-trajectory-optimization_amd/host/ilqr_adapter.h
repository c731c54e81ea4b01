// ilqr_adapter.h — host-side C++ mirror of the reference planner façade, above the C-ABI of include/cilqr.h.
//
// The reference's ROS node drives a stateful `iLQR` object (I/iLQR.h:16-59; I/ = CILQR/src/ilqr/include/ilqr/):
//     ilqrplanner.set_global_plan(global_path);  ilqrplanner.set_Obstacle(obstacles);
//     ilqrplanner.run_step(ego_state);           → X_result, U_result, ref_traj_result
// (I/ilqr_uncertainty_node.cpp:113-130).  This class keeps those names, argument meanings and the persistent, un-shifted
// warm start (`control_seq`, I/iLQR.cpp:9-15,253) so that call sequence is unchanged; every solve goes through
// cilqr_solve_batch with B = 1 (or B = candidates, see run_candidates) on the HIP device.  There is no CPU path.
//
// Differences from the reference interface, all forced by the boundary:
//   * Eigen types are replaced by the column-major `Matrix` below (Eigen is not a dependency of this library);
//   * device/HIP failures throw std::runtime_error (the reference solver cannot fail that way);
//   * the reference's Uncertainty class is not in its repository: the `Uncertainty` below carries what its
//     constructor is given at the call site (I/ilqr_uncertainty_node.cpp:111-112) and the cost is the one include/cilqr.h
//     defines at cilqr_set_uncertainty_map (parity unpinned); set_uncertainty_map / clear_uncertainty_map keep the
//     reference's names and effect (I/iLQR.cpp:28-35);
//   * nothing is printed to stdout (the reference prints three lines per solve, I/iLQR.cpp:240-242); the same facts are
//     available as last_iterations / last_exit / last_cost.
#pragma once

#include <vector>

#include "cilqr.h"

namespace cilqr_host {

// Dense column-major matrix of doubles: element (r, c) at a[r + rows*c] — Eigen::MatrixXd's default layout.
struct Matrix {
  int rows = 0, cols = 0;
  std::vector<double> a;
  Matrix() = default;
  Matrix(int r, int c) : rows(r), cols(c), a((size_t)r * c, 0.0) {}
  double& operator()(int r, int c) { return a[(size_t)c * rows + r]; }
  double operator()(int r, int c) const { return a[(size_t)c * rows + r]; }
};

using Parameters = cilqr_params;  // POD mirror of the reference's class Parameters (I/Parameters.h)
Parameters default_parameters();  // Parameters::Parameters(), I/Parameters.cpp:3-75

// Obstacle(Parameters p, MatrixXd dimension /*2×N*/, MatrixXd relative_pos_array /*4×N*/), I/Obstacle.h:13-25.
class Obstacle {
 public:
  Obstacle(const Parameters& p, const Matrix& dimension, const Matrix& relative_pos_array)
      : p(p), dimension(dimension), relative_pos_array(relative_pos_array) {}
  Parameters p;
  Matrix dimension;           // 2 × horizon: (length, width) per step
  Matrix relative_pos_array;  // 4 × horizon: (x, y, v, theta) per step
};

// What the reference node builds its Uncertainty object from, every tick (I/ilqr_uncertainty_node.cpp:111-112): the blurred
// occupancy layer of the map node (grid_map_msg "uncertainty_map"; rows×cols float32 column-major, 0..100, NaN unknown), its
// vehicle-frame geometry centred at (x_center, y_center) (map_param), and the vehicle pose the map was made at
// (map_msg.info.origin).  probes: footprint sampling of the safe_length × safe_width rectangle (include/cilqr.h).
struct Uncertainty {
  std::vector<float> layer;
  cilqr_map_geom geom{};
  double pose_x = 0.0, pose_y = 0.0, pose_theta = 0.0;
  int probes_l = 3, probes_w = 3;
};

// vehiclepub/Experiment as the node fills it (I/ilqr_uncertainty_node.cpp:243-284): start_pos[4], X flattened column by
// column (4 per step, horizon + 1 steps), U likewise (2 per step), planning_time in seconds.  (Its ros::Time start_time
// belongs to the caller.)  Column-major flattening is the memory order of X_result / U_result, so these are copies.
struct Experiment {
  double planning_time = 0.0;
  std::vector<double> start_pos, X, U;
};
Experiment flatten_experiment(const double start_pos[4], double planning_time, const Matrix& X, const Matrix& U);

class iLQR {
 public:
  // max_candidates > 1 reserves device buffers for run_candidates().
  explicit iLQR(const Parameters& params, int device = 0, int max_obstacles = 64, int max_candidates = 1);
  ~iLQR();
  iLQR(const iLQR&) = delete;
  iLQR& operator=(const iLQR&) = delete;

  void set_Obstacle(const std::vector<Obstacle>& obstacles);  // I/iLQR.cpp:20-23 (deep copy, like the reference)
  void clear_Obstacle();                                      // :24-27
  void set_uncertainty_map(const Uncertainty& uncertainty);    // :28-31 → Constraints::set_uncertainty_map (I/Constraints.cpp:520-524)
  void clear_uncertainty_map();                               // :32-35
  void set_global_plan(const Matrix& global_plan);            // :41-45, 2 × P waypoints

  // I/iLQR.cpp:201-245.  U is the warm start on entry and U_result on return; x_local_plan is the local plan's x row
  // (only its first and last entries are read, I/Constraints.cpp:31-33).
  void get_optimal_control_seq(const double x_0[4], Matrix& U, const double poly_coeffs[6],
                               const std::vector<double>& x_local_plan);
  void run_step(const double ego_state[4]);  // :247-255

  // Batched form of run_step for sampled ego states (e.g. the node's Gaussian pose noise, I/ilqr_uncertainty_node.cpp:82-110):
  // solves every candidate from the current warm start in ONE launch, keeps the minimum-cost one (cilqr_argmin_device)
  // as X_result/U_result/control_seq and returns its index.
  int run_candidates(const std::vector<double>& ego_states /* 4 per candidate */);

  Parameters params;
  Matrix X_result;         // 4 × (horizon + 1)
  Matrix U_result;         // 2 × horizon
  Matrix ref_traj_result;  // 2 × n_local_wpts
  int last_iterations = 0;
  int last_exit = 0;  // cilqr_exit
  double last_cost = 0.0;

 private:
  void pack_obstacles(int copies);
  cilqr_handle* h_ = nullptr;
  int device_, max_obstacles_, max_candidates_;
  Matrix control_seq_;  // I/iLQR.h:34
  Matrix global_plan_;
  std::vector<Obstacle> obstacles_;
  std::vector<double> obs_pose_, obs_dim_;  // packed [copy][obstacle][4N] / [2N]
};

}  // namespace cilqr_host
