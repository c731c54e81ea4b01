/*
 * cilqr_oracle.h — CPU restatement of the reference CILQR hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link, load or call
 * anything under oracle/.  The product (the HIP library behind include/cilqr.h) never does.
 *
 * Pinning: the reference solver sources cannot be compiled in the authoring
 * container without writing stand-ins for headers it lacks (mkl.h, unsupported/Eigen/CXX11/Tensor and
 * the reference's own missing Uncertainty.h), so this restatement is pinned by
 *   (1) the known-answer solves recorded in SURVEY.md §8(c) (tests/golden/survey_known_answers.json),
 *   (2) oracle/_ref: the reference's Parameters.cpp compiled as-is (defaults), and the reference's
 *       vendored Eigen 3.2.10 EigenSolver / ColPivHouseholderQR run on the same 2×2 / Vandermonde
 *       inputs (the only library algorithms on the path),
 *   (3) for the costmap warp: the reference's grid_map_core compiled as-is (oracle/_ref) and the
 *       known answers of its own gtests.
 *
 * Citations: I/ = /root/reference/CILQR/src/ilqr/include/ilqr/.
 */
#ifndef CILQR_ORACLE_H_
#define CILQR_ORACLE_H_

#include "cilqr.h"

#ifdef __cplusplus
extern "C" {
#endif

/* I/Parameters.cpp:3-75, I/iLQR.cpp:17-18 */
void oracle_params_default(cilqr_params* p);
/* I/iLQR.cpp:9-15 */
void oracle_default_control_seq(int N, double* U);

/* I/Model.cpp:17-30 */
void oracle_forward_simulate(const cilqr_params* p, const double* state, const double* control, double* next);
/* I/iLQR.cpp:51-62 */
void oracle_nominal_trajectory(const cilqr_params* p, int N, const double* x0, const double* U, double* X);
/* I/Constraints.cpp:24-59 */
void oracle_find_closest_point(const cilqr_params* p, const double* state, const double* coeffs,
                               double xplan_first, double xplan_last, double* out_xy);
/* I/Obstacle.cpp:39-112 (+ barrier :21-32).  pose = relative_pos_array column, dim = dimension column. */
void oracle_obstacle_cost(const cilqr_params* p, const double* pose, const double* dim,
                          const double* ego_state, double* vx4, double* mx16);
/* I/Constraints.cpp:145-227 without the uncertainty-map term (added by the *_unc entry points below) */
void oracle_state_cost(const cilqr_params* p, int N, const double* X, const double* coeffs,
                       double xplan_first, double xplan_last, int M, const double* obs_pose,
                       const double* obs_dim, const double* obs_weight, double* l_x, double* l_xx);
/* I/Constraints.cpp:86-137 */
void oracle_control_cost(const cilqr_params* p, int N, const double* X, const double* U, double* l_u,
                         double* l_uu);
/* I/Constraints.cpp:534-561 */
double oracle_get_J(const cilqr_params* p, int N, const double* X, const double* U, const double* coeffs,
                    double xplan_first, double xplan_last);
/* I/Model.cpp:100-155: A (4×4×N) and B (2×4×N), stored transposed, column-major tensors */
void oracle_AB(const cilqr_params* p, int N, const double* vel, const double* theta, const double* acc,
               double* A, double* B);
/* I/iLQR.cpp:155-175: V·diag(1/(max(eig,0)+lamb))·Vᵀ with Eigen::EigenSolver semantics for a real 2×2
 * (column-major Quu[4] → Qinv[4]).  Returns 0, or -1 where the reference's solver would not return
 * real eigenvectors. */
int oracle_quu_inverse(const double* Quu, double lamb, double* Qinv, double* eval2, double* evec4);
/* I/iLQR.cpp:91-195.  Returns 1 on success (reference `true`). */
int oracle_backward_pass(const cilqr_params* p, int N, const double* X, const double* U, const double* coeffs,
                         double xplan_first, double xplan_last, int M, const double* obs_pose,
                         const double* obs_dim, const double* obs_weight, double lamb, double* k, double* K);
/* I/iLQR.cpp:68-86 */
void oracle_forward_pass(const cilqr_params* p, int N, const double* X, const double* U, const double* k,
                         const double* K, double* X_new, double* U_new);

/* I/iLQR.cpp:201-245.  U in/out.  trace (optional, may be NULL): per executed iteration
 * {J_new, lamb after update, accepted(1/0)} → 3 doubles × max_iterations.  Returns iteration_times. */
int oracle_solve(const cilqr_params* p, int N, int M, const double* x0, double* U, const double* coeffs,
                 double xplan_first, double xplan_last, const double* obs_pose, const double* obs_dim,
                 const double* obs_weight, double* X_out, double* J_out, int* status_out, double* trace);

/* Costmap-lookup uncertainty cost (uncertainty_oracle.c): semantics defined by include/cilqr.h — the reference's class
 * Uncertainty is absent from its repository, PARITY UNPINNED.  *m holds HOST pointers.  The solve adds w_uncertainty·(vx, mx)
 * where Constraints::get_state_cost does (I/Constraints.cpp:188-201). */
int oracle_layer_bilinear(const float* layer, const cilqr_map_geom* g, double qx, double qy, double* value, double* d_dx,
                          double* d_dy);
void oracle_uncertainty_cost(const cilqr_params* p, const cilqr_uncertainty_map* m, int b, const double* state, double* cost,
                             double* vx4, double* mx16);
/* oracle_solve / oracle_solve_batch with the map set (m may be NULL: identical to the plain calls); b = the solve's index in
 * its batch (selects its layer and pose). */
int oracle_solve_unc(const cilqr_params* p, int N, int M, const double* x0, double* U, const double* coeffs,
                     double xplan_first, double xplan_last, const double* obs_pose, const double* obs_dim,
                     const double* obs_weight, const cilqr_uncertainty_map* m, int b, double* X_out, double* J_out,
                     int* status_out, double* trace);
int oracle_solve_batch_unc(const cilqr_params* p, int B, int N, int M, const double* x0, double* U,
                           const double* poly, const double* xplan_fl, const double* obs_pose,
                           const double* obs_dim, const double* obs_weight, const cilqr_uncertainty_map* m, double* X_out,
                           double* J_out, int* iters_out, int* status_out, int threads);

/* Batch driver (layouts of include/cilqr.h), OpenMP over the batch with `threads` threads. */
int oracle_solve_batch(const cilqr_params* p, int B, int N, int M, const double* x0, double* U,
                       const double* poly, const double* xplan_fl, const double* obs_pose,
                       const double* obs_dim, const double* obs_weight, double* X_out, double* J_out,
                       int* iters_out, int* status_out, int threads);
int oracle_max_threads(void);

/* I/LocalPlanner.cpp:101-117 with Eigen::ColPivHouseholderQR::solve semantics */
void oracle_polyfit(const double* x, const double* y, int n, int degree, double* coeffs);
/* I/LocalPlanner.cpp:25-96; returns number of local waypoints */
int oracle_local_plan(const cilqr_params* p, const double* path, int P, const double* ego_state,
                      double* coeffs, double* ref_traj);

/* Costmap warp: M/src/local_costmap.cpp:242-264 over G/grid_map_core index math
 * (GridMapMath.cpp:114-156, GridMap.cpp:45-62,160-166).  Returns the number of destination cells whose
 * source position is outside the source map (reference: std::out_of_range thrown); those are set NaN. */
void oracle_map_geom_set(cilqr_map_geom* g, double len_x, double len_y, double res, double pos_x, double pos_y);
int  oracle_map_get_position(const cilqr_map_geom* g, int i, int j, double* px, double* py);
int  oracle_map_get_index(const cilqr_map_geom* g, double px, double py, int* i, int* j);
long oracle_warp_costmap(const float* src, const cilqr_map_geom* sg, float* dst, const cilqr_map_geom* dg,
                         double vx, double vy, double vtheta, const float* bbox, int threads);

/* Uncertainty propagation ("blur"): M/src/arbitrary_transformation.cu:8-157, M/include/ARBIT.cuh:51-107,
 * M/src/local_costmap.cpp:474-498 over G/grid_map_core EllipseIterator.  src/out: float32 column-major layers of geometry g;
 * cells with linear index < index are left NaN (the reference never writes them).  Returns the number of cells whose
 * ellipse was empty (copy-through).  count_out (optional): cells inside each ellipse. */
void oracle_blur_ellipse(double a, double b, double c, double* half_major, double* half_minor, double* angle);
long oracle_blur(const float* src, const cilqr_map_geom* g, int index, double sin_t, double cos_t, double sigma_x,
                 double sigma_y, double sigma_theta, float* out, int* count_out, int threads);

/* OccupancyGrid (int8, -1 = unknown) <-> grid_map float32 layer (NaN = unknown), cell order reversed
 * (G/grid_map_ros/src/GridMapRosConverter.cpp:259-266 and :293-306); n = rows*cols; start index zero. */
void oracle_occupancy_to_layer(const int8_t* occ, long n, float* layer);
void oracle_layer_to_occupancy(const float* layer, long n, float data_min, float data_max, int8_t* occ);

#ifdef __cplusplus
}
#endif
#endif
