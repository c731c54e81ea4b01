// cilqr_internal.h — shared between the C-ABI translation unit and the HIP kernels (not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cilqr.h"

namespace cilqr {

// Scalars the kernels need, derived once per handle from cilqr_params on the host (so that the libm calls
// tan(steer_angle_*) are the host's, exactly as in the reference, I/Model.cpp:20 / I/Constraints.cpp:119-121).
struct KParams {
  double dt;             // timestep
  double desired_speed;
  double tolerance;
  double w_acc, w_yawrate, w_pos, w_vel, w_obstacle;
  double q1_acc, q2_acc, q1_yawrate, q2_yawrate;
  double q1_front, q2_front, q1_rear, q2_rear;
  double acc_max, acc_min;
  double yaw_hi, yaw_lo;  // tan(steer_angle_max)/wheelbase, tan(steer_angle_min)/wheelbase: yaw-rate bound per unit speed
  double half_dt2;        // timestep²/2
  double wheelbase, speed_max;
  double t_safe, s_safe_a, s_safe_b, ego_rad, ego_front, ego_rear;
  double lamb_factor, lamb_max;
  int32_t max_iterations;
  int32_t n_samples;  // num_of_local_wpts * 10 (I/Constraints.cpp:28)
};

// The uncertainty map as the kernels read it (cilqr_set_uncertainty_map*, include/cilqr.h); layer == nullptr: no map set.
struct UncArgs {
  const float* layer;     // rows*cols float32 column-major per solve (stride floats apart; 0: shared)
  const double* poses;    // null, or [B][3] per-solve (x, y, theta) of the vehicle frame in the planning frame
  long long stride;
  int32_t rows, cols, nl, nw;
  double x_first, y_first, inv_res;  // centre of cell (0, 0) and 1/resolution (G/grid_map_core/src/GridMapMath.cpp:114-127)
  double px, py, cp, sp;             // shared pose: position, cos and sin of its heading (host libm)
  double la0, la_step, wb0, wb_step; // footprint probe offsets along / across the heading: la0 + k*la_step, wb0 + l*wb_step
  double q1, q2, scale;              // barrier constants; scale = w_uncertainty / (nl*nw)
};

constexpr int DIAG_SLOTS = 16;  // uint64 per solve in the diagnostic buffer (include/cilqr.h, cilqr_set_diag_buffer)

struct SolveArgs {
  const double* x0;
  double* U;
  const double* poly;
  const double* xplan_fl;
  const double* obs_pose;
  const double* obs_dim;
  const double* obs_weight;  // may be null
  double* X_out;
  double* J_out;        // may be null
  int32_t* iters_out;   // may be null
  int32_t* status_out;  // may be null
  const double* samp_off;  // sampled obstacles only: [B][M][n_samples][3] = (dx, dy, dtheta); then obs_pose/obs_dim are the
  int32_t n_samples;       // nominal trajectories of the M obstacles, every sample weighs samp_w, and the wavefront family runs.
  double samp_w;           // n_samples == 0: ordinary obstacles
  double* obs_tab;      // workspace: [B][M][N][6] (sampled: [B][M][N][8])
  double* fwd;          // workspace of the one-wavefront-per-solve family: [B][N + 1][16] forward-pass records (cilqr_solve.hip)
  int32_t* redo;        // workspace: [B] hand-over flags from the fast kernel to the general kernel
  unsigned long long* diag;  // null, or [B][DIAG_SLOTS] phase cycle totals (diagnostic instantiation)
  int32_t* passes;      // null, or [B]: backward+forward passes each solve actually executed (cilqr_set_pass_count_buffer)
  // Dispatch order of the one-wavefront-per-solve family for batches beyond one solve per SIMD (cilqr_api.cpp, schedule hint):
  const int32_t* order;  // null, or [B]: workgroup i runs solve order[i] (a permutation: longest solves of the previous call first)
  int32_t* hint_passes;  // null, or [B]: passes of each solve of THIS call, from which the next call's order is built
  int32_t B, N, M;
  uint32_t flags;
  // The batch has at most one solve per SIMD — the launcher may give every solve a second wavefront (cilqr_solve.hip).
  // 1: cilqr_solve_pair_kernel (the next linearisation runs behind the forward pass instead of after it; an experiment);
  // 2, 3: cilqr_solve_share_kernel with that many wavefronts (all of them work on phase L at the same time; the default)
  int32_t pair;
  // one-wavefront family, static obstacles: LDS bytes a workgroup may take with its obstacle table inside (0: 32 KiB) — cilqr_api.cpp, lds_table_budget
  int32_t tab_budget;
  // grouped family: 1 = in phase L the lanes of a wavefront's finished solves take steps of the unfinished ones (cilqr_solve_groups.hip)
  int32_t steal;
  // sampled obstacles: 1 = two wavefronts per solve share the obstacle entries of phase L (cilqr_solve.hip, cilqr_solve_split_kernel)
  int32_t split;
  KParams kp;
  UncArgs unc;
};

// Launchers (defined in the .hip files). All are asynchronous on `stream`.
// One wavefront per solve, LDS-resident (cilqr_solve.hip).
hipError_t launch_solve_wave(const SolveArgs& a, hipStream_t stream);
hipError_t launch_schedule_order(const int32_t* passes, int B, int32_t* order, hipStream_t stream);  // passes descending
size_t solve_lds_bytes(int N, int n_samples);
constexpr size_t SOLVE_LDS_MAX = 160 * 1024;  // LDS of one CU: the horizon bound of the wavefront family
bool solve_table_in_lds(int N, int M, int n_samples, int budget);   // static obstacles: the [M][N] entry table lies in LDS (else in the workspace)
bool solve_share_applies(int N, int M, int n_samples, int budget);  // … and the shape can take cilqr_solve_share_kernel (wavefronts share phase L)
size_t solve_sampled_lds_bytes(int n_obs, int n_samples);   // additional LDS of the sampled-obstacle mode
size_t solve_sampled_tab_doubles(int n_obs, int N);         // its workspace need per solve (in obs_tab)
// G lanes per solve (G in {1,2,4,8,16,32}), workspace `ws` of solve_groups_ws_doubles(B, N) doubles (cilqr_solve_groups.hip).
hipError_t launch_solve_groups(const SolveArgs& a, int G, double* ws, hipStream_t stream);
size_t solve_groups_ws_doubles(int B, int N);
// Test hook: the map cost alone at n states [n][4] → cost[n], vx[n][2], mx[n][3] (solve 0's layer and pose).
hipError_t launch_unc_cost(const UncArgs& u, int n, const double* states, double* cost, double* vx, double* mx, hipStream_t stream);
hipError_t launch_quu_inverse(int n, const double* q, const double* lamb, double* out, int general, hipStream_t stream);
hipError_t launch_closest_sample(int n, int S, const double* in, int32_t* out, hipStream_t stream);  // test hook (cilqr_debug_closest_sample)

// Batched LocalPlanner (local_plan.hip): one lane per candidate.
struct LocalPlanArgs {
  const double* path;      // 2×P column-major; candidate b reads path + b*path_stride (0: one shared path)
  long long path_stride;   // in doubles
  const double* ego;       // [B][4]
  double* poly;            // [B][6]
  double* xplan_fl;        // [B][2]
  double* ref_traj;        // null or [B][2*n_wpts]
  int32_t* n_out;          // null or [B]
  int32_t B, P, n_wpts, cols;
};
hipError_t launch_local_plan(const LocalPlanArgs& a, hipStream_t stream);
size_t local_plan_lds_bytes(int n_wpts, int cols);

// Min-cost selection (cilqr_select.hip).  out_pair {J_min, index} and/or out_triple {J_min, index, offset} (either may be null).
hipError_t launch_argmin(const double* J, int B, double* out_pair, double* out_triple, double offset, hipStream_t stream,
                         double pair_offset = 0.0);
hipError_t launch_select(const double* triples, int n_ranks, double* out_pair, hipStream_t stream);

struct WarpArgs {
  const float* src;
  float* dst;
  const float* bbox;  // may be null
  unsigned long long* n_oob;  // may be null
  cilqr_map_geom sg, dg;
  double vx, vy, sin_t, cos_t;
};
hipError_t launch_warp(const WarpArgs& a, hipStream_t stream);
// K frames per launch (dst rows must be a multiple of 4): poses [K][4] = (vx, vy, sin theta, cos theta) on the device,
// destination frames back to back, n_oob [K] or null.
struct WarpBatchArgs {
  const float* src;
  float* dst;
  const float* bbox;          // may be null (one layer, shared by the frames)
  unsigned long long* n_oob;  // may be null
  const double* poses;        // null: ONE frame, its pose in pose0 (no table to upload)
  double pose0[4];
  cilqr_map_geom sg, dg;
};
hipError_t launch_warp_batch(const WarpBatchArgs& a, int K, hipStream_t stream);

struct BlurArgs {
  const float* src;
  float* out;
  int32_t* count_out;  // may be null
  int8_t* occ_out;     // may be null: the same cells as an OccupancyGrid (toOccupancyGrid of `out`, reversed order)
  float occ_min, occ_den;  // dataMin and dataMax - dataMin of that conversion
  cilqr_map_geom g;
  int32_t index;       // first linear cell index processed (cells before it are set NaN)
  double sin_t, cos_t, sigma_x, sigma_y, sigma_theta;
};
hipError_t launch_blur(const BlurArgs& a, hipStream_t stream);
// OccupancyGrid <-> layer (costmap_occupancy.hip)
hipError_t launch_occ_to_layer(const int8_t* occ, float* layer, long n, hipStream_t stream);
// steps_ws: 102 floats of device workspace for the step table of large conversions (null: always the per-cell division)
hipError_t launch_layer_to_occ(const float* layer, int8_t* occ, long n, float data_min, float data_max, float* steps_ws,
                               hipStream_t stream);
hipError_t launch_blur_ellipse(int n, const double* abc, double* out, hipStream_t stream);

}  // namespace cilqr
