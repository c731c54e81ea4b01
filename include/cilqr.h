/*
 * cilqr.h — C-ABI of the MI355X-native batched constrained-iLQR (CILQR) solver and costmap warp.
 *
 * Drop-in boundary for the hot path of Leo-Liao-Chao/Uncertainty-Aware-CILQR-for-Trajectory-Optimization.
 * All citations are relative to the reference tree, with I/ = CILQR/src/ilqr/include/ilqr/,
 * M/ = CILQR/src/map_engine/, G/ = CILQR/src/grid_map/.
 *
 * The reference has no FFI layer: the planner node calls C++ methods on a stateful `iLQR` object
 * (I/iLQR.h:16-59).  This header is what a binding for that path would bind instead:
 *
 *   reference interface                                      replaced by
 *   -------------------------------------------------------  ------------------------------------
 *   Parameters::Parameters()            I/Parameters.cpp:3-75   cilqr_params_default
 *   iLQR::iLQR(const Parameters&)       I/iLQR.cpp:3-19         cilqr_create (+ cilqr_default_control_seq)
 *   iLQR::get_optimal_control_seq       I/iLQR.cpp:201-245      cilqr_solve_batch / cilqr_solve_batch_device
 *   iLQR::set_Obstacle / clear_Obstacle I/iLQR.cpp:20-27        obs_* arguments of cilqr_solve_batch (M = 0 ⇒ cleared)
 *   iLQR::set_uncertainty_map / clear_uncertainty_map I/iLQR.cpp:28-35   cilqr_set_uncertainty_map(_device) / cilqr_clear_uncertainty_map
 *   Constraints::get_J                  I/Constraints.cpp:534-561   J_out of cilqr_solve_batch
 *   GridMapRosConverter::from/toOccupancyGrid G/grid_map_ros/src/GridMapRosConverter.cpp:225-307
 *                                                                cilqr_occupancy_to_layer / cilqr_layer_to_occupancy(_device)
 *   LocalCostmap::odomCallback (one frame) M/src/local_costmap.cpp:172-305 cilqr_costmap_frame_device
 *   LocalPlanner::get_local_plan(_coeffs) I/LocalPlanner.cpp:25-117 cilqr_local_plan (host pre-step),
 *                                                                cilqr_local_plan_batch(_device) (B candidates on the device)
 *   LocalCostmap::odomCallback warp loop M/src/local_costmap.cpp:242-264 cilqr_warp_costmap(_device)
 *   thrust_propagateUncertainty     M/src/arbitrary_transformation.cu:8-157  cilqr_blur_costmap(_device)
 *   (none: batch min-cost selection is new)      cilqr_argmin_device, cilqr_argmin_global_device (RCCL),
 *                                                                cilqr_create_multi / cilqr_multi_solve_batch
 *
 * Conventions
 *   - fp64 everywhere in the solver; float32 map payloads in the warp.
 *   - Per-solve layouts equal Eigen column-major as used by the reference and by
 *     vehiclepub/Experiment.msg flattening (I/ilqr_uncertainty_node.cpp:265-274):
 *       U  : [a0, w0, a1, w1, ...]               2*N doubles
 *       X  : [x0, y0, v0, th0, x1, ...]          4*(N+1) doubles
 *       obstacle pose (relative_pos_array, I/Obstacle.h:25) : 4*N doubles, column t = (x, y, v, theta)
 *       obstacle dimension (I/Obstacle.h:24)                : 2*N doubles, column t = (length, width)
 *     The batch index is outermost, then (for obstacle tables) the obstacle index.
 *   - Every function returns 0 on success and a negative cilqr_status on failure; the message is
 *     available from cilqr_last_error() (thread-local).  Nothing is printed to stdout.
 *   - One handle = one device = one host thread at a time (the reference solver is not re-entrant
 *     either: `static int iteration_times`, I/iLQR.cpp:208).
 *   - There is NO CPU fallback: every compute entry point fails with CILQR_ERR_NO_DEVICE when no
 *     gfx950 device is usable.
 */
#ifndef CILQR_H_
#define CILQR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CILQR_NX 4
#define CILQR_NU 2
#define CILQR_POLY_COEFFS 6   /* poly_order + 1, I/Parameters.cpp:7 */
#define CILQR_MAX_HORIZON 384  /* per-solve arrays of the LDS-resident family at the default sample count: 98 KiB of 160 */
#define CILQR_ABI_VERSION 2
#define CILQR_COMM_ID_BYTES 128 /* an RCCL ncclUniqueId, carried opaquely */

/* Field-for-field POD mirror of class Parameters (I/Parameters.h:5-91) — only the fields the
 * constructor initialises (I/Parameters.cpp:6-74) — plus the two constants iLQR::iLQR sets
 * (lamb_factor, lamb_max; I/iLQR.cpp:17-18). */
typedef struct cilqr_params {
  /* planning parameters */
  int32_t num_of_local_wpts;   /* 20  */
  int32_t poly_order;          /* 5   */
  /* iLQR parameters */
  int32_t horizon;             /* 40  */
  int32_t max_iterations;      /* 20  */
  int32_t num_states;          /* 4   */
  int32_t num_ctrls;           /* 2   */
  double desired_speed;        /* 5.0 */
  double timestep;             /* 0.1 */
  double tolerance;            /* 1e-4 */
  /* cost weights */
  double w_acc, w_yawrate;     /* 1.0, 4.0 */
  double w_pos, w_vel;         /* 0.65, 3.0 */
  double w_obstacle, w_uncertainty; /* 1.0, 1.0 */
  /* barrier q1/q2 */
  double q1_acc, q2_acc;
  double q1_yawrate, q2_yawrate;
  double q1_front, q2_front;
  double q1_rear, q2_rear;
  double q1_uncertainty, q2_uncertainty;
  /* limits */
  double acc_max, acc_min;
  double steer_angle_min, steer_angle_max;
  /* ego vehicle */
  double wheelbase, speed_max;
  double steer_control_max, steer_control_min;
  double throttle_control_max, throttle_control_min;
  /* obstacle parameters */
  double t_safe, s_safe_a, s_safe_b;
  double ego_rad, ego_front, ego_rear;
  double length, width;
  double safe_length, safe_width;
  /* iLQR::iLQR, I/iLQR.cpp:17-18 */
  double lamb_factor, lamb_max;
} cilqr_params;

typedef enum cilqr_status {
  CILQR_OK = 0,
  CILQR_ERR_ARG = -1,        /* bad argument (null pointer, size out of the range given at create) */
  CILQR_ERR_NO_DEVICE = -2,  /* no usable gfx950 device / HIP runtime error at create */
  CILQR_ERR_HIP = -3,        /* HIP runtime error during a call */
  CILQR_ERR_UNSUPPORTED = -4,/* parameter combination the kernels do not implement (e.g. num_states != 4) */
  CILQR_ERR_COMM = -5        /* RCCL error in the cross-GPU exchange step */
} cilqr_status;

/* Per-solve exit reason written to status_out (I/iLQR.cpp:211-239). */
typedef enum cilqr_exit {
  CILQR_EXIT_TOLERANCE = 0,   /* accepted step with |J_new - J_old| < tolerance  (:225-228) */
  CILQR_EXIT_LAMBDA_MAX = 1,  /* rejected step drove lamb above lamb_max         (:232-236) */
  CILQR_EXIT_MAX_ITER = 2,    /* loop ran max_iterations times                   (:211)     */
  CILQR_EXIT_NUMERIC = 3      /* non-finite value met in the backward pass (reference: EigenSolver
                                 failure → break, :214-215); X/U of the last accepted iterate */
} cilqr_exit;

/* Flags for cilqr_solve_batch*. */
#define CILQR_FLAG_NONE 0u
/* Execute the backward/forward passes of rejected iterations exactly as the reference loop does
 * instead of stopping at the first rejection (results are identical). */
#define CILQR_FLAG_FAITHFUL_ITERS 1u
/* Test hook: every solve is handed to the GENERAL kernels (branching passes, library-range sincos of every heading, the regularised
 * Q_uu inverse in its eigenvalue-clamping form and the value update as the reference's direct product K' Q_uu) — the path
 * that otherwise only solves take which the production kernels do not cover.  Same results to rounding, several times slower. */
#define CILQR_FLAG_GENERAL_ONLY 2u

typedef struct cilqr_handle cilqr_handle;

/* Geometry of a grid_map layer (G/grid_map_core/src/GridMap.cpp:45-62): float32, column-major
 * rows×cols, cell (i,j) centre = pos + (len/2 - res/2) - res*(i,j)  (GridMapMath.cpp:114-127). */
typedef struct cilqr_map_geom {
  int32_t rows, cols;     /* size_(0), size_(1)  */
  double res;             /* resolution_ */
  double len_x, len_y;    /* length_  (= size*res after setGeometry) */
  double pos_x, pos_y;    /* position_ (map centre in its parent frame) */
} cilqr_map_geom;

/* The costmap the uncertainty cost reads.  The reference constructs, every tick,
 *   Uncertainty vehicle_map(params, map_msg, grid_map_msg, x_center, y_center, SIGMA_X, SIGMA_Y, SIGMA_THETA, 0, 0, nh)
 * (I/ilqr_uncertainty_node.cpp:111-112) and hands it to iLQR::set_uncertainty_map (:113); class Uncertainty itself is ABSENT
 * from the reference repository.  What those arguments carry is mirrored here: the blurred occupancy layer
 * the map node publishes (grid_map_msg, layer "uncertainty_map" = output of cilqr_blur_costmap*, values 0..100, NaN unknown),
 * its vehicle-frame geometry with the centre at (x_center, y_center) (map_param, M/src/local_costmap.cpp:793-799), and the pose
 * of that vehicle frame in the planning frame (map_msg.info.origin = the vehicle pose at map time, :300).  The sigmas were
 * consumed upstream by the blur.  THE ARITHMETIC OF THE COST IS DEFINED BY THIS LIBRARY (below, at
 * cilqr_set_uncertainty_map): there is no reference arithmetic to match — parity unpinned. */
typedef struct cilqr_uncertainty_map {
  const float* layer;      /* rows*cols float32, column-major */
  cilqr_map_geom geom;     /* vehicle-frame geometry of the layer */
  double pose_x, pose_y, pose_theta; /* vehicle frame in the planning frame */
  const double* poses;     /* NULL, or [B][3] per-solve (pose_x, pose_y, pose_theta): then the three scalars are ignored */
  int64_t layer_stride;    /* floats from solve b's layer to solve b+1's; 0: one layer shared by the batch */
  int32_t probes_l, probes_w; /* footprint probe grid along / across the ego heading, each >= 1 */
} cilqr_uncertainty_map;

/* --- parameters ------------------------------------------------------------------------------- */
void cilqr_params_default(cilqr_params* p);             /* I/Parameters.cpp:3-75 + I/iLQR.cpp:17-18 */
int  cilqr_abi_version(void);
int  cilqr_device_count(void);  /* gfx950 devices this process can use (0: none; there is no CPU path) */
const char* cilqr_last_error(void);

/* Initial warm-start control sequence of a fresh planner (I/iLQR.cpp:9-15): row 0 = 0.5, row 1 = 0 for
 * the first N/2 steps then 0.1.  Writes 2*N doubles. */
int cilqr_default_control_seq(int N, double* U);

/* --- host pre-step (stays on the host) -------------------------------------- */
/* LocalPlanner::{closest_point_index,get_local_wpts,get_local_plan,get_local_plan_coeffs,polyfit}
 * (I/LocalPlanner.cpp:25-117).  path: 2×P column-major.  Outputs: coeffs[poly_order+1]; ref_traj 2×n_out
 * column-major (row 0 = waypoint x, row 1 = fitted y), n_out ≤ num_of_local_wpts written to *n_out. */
int cilqr_local_plan(const cilqr_params* p, const double* path, int P, const double* ego_state,
                     double* coeffs, double* ref_traj, int* n_out);

/* The same pre-step for B candidate ego poses at once, on the device, so that a batch solve can start
 * from raw (global_path, ego): candidate b reads its 2×P column-major path at path + b*path_stride doubles
 * (path_stride = 0: one path shared by all candidates, the reference's set_global_plan).  ego [B][4];
 * outputs poly [B][6] (coefficients past poly_order are 0), xplan_fl [B][2] (first and last x of the slice, the two
 * elements of x_local_plan the solve reads), ref_traj [B][2*num_of_local_wpts] or NULL (2×n column-major per candidate,
 * the tail past n untouched), n_out [B] or NULL.  The fit is the host pre-step's, sum for sum; the Vandermonde entries are
 * correctly rounded powers where the host uses libm's pow (see local_plan.hip).  Uses the handle's num_of_local_wpts and
 * poly_order (≤ 5).  The *_device form takes device pointers and is asynchronous on `stream`. */
int cilqr_local_plan_batch(cilqr_handle* h, int B, int P, const double* path, int64_t path_stride, const double* ego,
                           double* poly, double* xplan_fl, double* ref_traj, int32_t* n_out);
int cilqr_local_plan_batch_device(cilqr_handle* h, void* stream, int B, int P, const double* path, int64_t path_stride,
                                  const double* ego, double* poly, double* xplan_fl, double* ref_traj, int32_t* n_out);

/* --- solver ----------------------------------------------------------------------------------- */
/* Sizes are upper bounds; device workspaces are allocated once here, never in solve. device = HIP
 * ordinal. */
int cilqr_create(const cilqr_params* p, int max_batch, int max_horizon, int max_obstacles, int device,
                 cilqr_handle** out);
int cilqr_destroy(cilqr_handle* h);

/* Page-locked host memory for the buffers handed to the host-buffer entry points: from such memory their copies are
 * asynchronous DMA transfers; any other host memory works too (the HIP runtime then stages each copy).  Returns NULL on failure. */
void* cilqr_host_alloc(size_t bytes);
int   cilqr_host_free(void* p);

/* Batched iLQR::get_optimal_control_seq (I/iLQR.cpp:201-245) — host buffers, synchronous.
 * Nothing is allocated per call: the device arena and a pinned staging buffer are sized at cilqr_create.  A call whose arrays
 * total ≤ 1 MiB (the drop-in B = 1 tick, a handful of candidates) travels as one packed host→device and one device→host copy.
 *   x0        [B][4]            ego state (x, y, v, theta)
 *   U         [B][2*N]  in/out  warm start in, U_result out (I/iLQR.cpp:222,244)
 *   poly      [B][6]            poly_coeffs, ascending powers
 *   xplan_fl  [B][2]            first and last element of x_local_plan (the only ones read,
 *                               I/Constraints.cpp:31-33)
 *   obs_pose  [B][M][4*N], obs_dim [B][M][2*N]   (ignored when M == 0)
 *   obs_weight[B][M] or NULL    per-obstacle factor applied where the reference applies
 *                               Parameters::w_obstacle (I/Constraints.cpp:184-185); NULL ⇒ p.w_obstacle
 *   X_out     [B][4*(N+1)]      X_result
 *   J_out     [B]               Constraints::get_J(X_result, U_result)
 *   iters_out [B]               iteration_times of the reference loop (I/iLQR.cpp:212)
 *   status_out[B]               cilqr_exit
 * J_out / iters_out / status_out may be NULL. */
int cilqr_solve_batch(cilqr_handle* h, int B, int N, int M,
                      const double* x0, double* U, const double* poly, const double* xplan_fl,
                      const double* obs_pose, const double* obs_dim, const double* obs_weight,
                      double* X_out, double* J_out, int32_t* iters_out, int32_t* status_out,
                      uint32_t flags);

/* Same, with every pointer a DEVICE pointer on the handle's device and the work enqueued on `stream`
 * (a hipStream_t passed as void*; NULL = the HIP null stream, as everywhere in HIP).  Asynchronous: returns after launch.
 * The handle's device workspaces (obstacle table, grouped-family arrays, hand-over flags) serve ONE solve at a time: calls
 * enqueued on the same stream follow each other and are fine; solves that may overlap in time on different streams need
 * different handles.
 * Scheduling: a batch of more solves than the device has SIMDs is dispatched longest-first by the pass counts the solves of
 * the PREVIOUS call on this handle had (same B, same stream: a planner solves nearly the same scenes tick after tick).  This
 * only changes which solves start first — results are bit-identical for any order — and is switched off by
 * CILQR_NO_SCHEDULE_HINT in the environment at cilqr_create. */
int cilqr_solve_batch_device(cilqr_handle* h, void* stream, int B, int N, int M,
                             const double* x0, double* U, const double* poly, const double* xplan_fl,
                             const double* obs_pose, const double* obs_dim, const double* obs_weight,
                             double* X_out, double* J_out, int32_t* iters_out, int32_t* status_out,
                             uint32_t flags);

/* --- costmap-lookup uncertainty cost ------------------------------------------------------------------
 * iLQR::set_uncertainty_map / clear_uncertainty_map (I/iLQR.cpp:28-35 → I/Constraints.cpp:520-528): while a map is set, every
 * later solve on the handle adds  w_uncertainty · (vx, mx)  of the map cost to l_x, l_xx at every step, exactly where
 * Constraints::get_state_cost does (I/Constraints.cpp:188-201); get_J is unchanged (its uncertainty term is commented out in the
 * reference, :553-557).  The cost itself — Uncertainty::get_uncertainty_cost(state) → {x, vx(4), mx(4×4)} — has NO source in the
 * reference; this library defines it, using only the reference's own ingredients:
 *   footprint  probes_l × probes_w points on the rectangle safe_length × safe_width (Parameters, launch 1.1 / 0.9) centred on the
 *              state's (x, y) and turned by its heading: body offsets a_k = -safe_length/2 + k·safe_length/(probes_l-1)
 *              (0 when probes_l = 1), b_l likewise across;
 *   lookup     each probe → vehicle frame (rigid transform by the map pose) → bilinear interpolation of the layer over the four
 *              surrounding cell centres, as GridMap::atPositionLinearInterpolated does (G/grid_map_core/src/GridMap.cpp:770-837),
 *              evaluated in double, with the interpolant's own gradient; a probe whose four cells are not all inside the map
 *              and finite contributes nothing;
 *   barrier    the reference's exponential barrier and Gauss-Newton form (Obstacle::barrier_function, I/Obstacle.cpp:21-32) with
 *              c = occupancy/100 - 1:  x = q1·exp(q2·c),  vx = q2·x·∇c,  mx = q2²·x·∇c∇cᵀ,  q1 = q1_uncertainty,
 *              q2 = q2_uncertainty (I/Parameters.cpp:41-42); ∇c is taken with respect to (x, y) only — the heading's effect on the
 *              probe positions is ignored, as the reference ignores it for its ego circles (I/Obstacle.cpp:75-78);
 *   result     the mean over the probes.
 *   omitted    (stated so that nobody reads more into vx, mx than is there) vx has no heading entry: d/dθ of the cost through the
 *              turning footprint is dropped (vx[2] = vx[3] = 0); mx is the Gauss-Newton outer product only: the interpolant's own
 *              curvature (the bilinear cross term ∂²o/∂x∂y) and every θ row and column are dropped.  The x, y entries of vx ARE
 *              the exact derivative of the cost at fixed heading (checked by finite differences, tests/test_oracle.py and the
 *              -m gpu twin on the kernel's own value).
 * cilqr_set_uncertainty_map_device: every pointer in *map is a device pointer that must stay valid (and is read) during later
 * solves — e.g. the uncertainty_layer cilqr_costmap_frame_device wrote on the same stream.  cilqr_set_uncertainty_map: host
 * pointers; one shared layer (layer_stride = 0, poses = NULL) copied into a buffer the handle owns.  Both return
 * CILQR_ERR_ARG for probes < 1 or a bad geometry. */
int cilqr_set_uncertainty_map_device(cilqr_handle* h, const cilqr_uncertainty_map* map);
int cilqr_set_uncertainty_map(cilqr_handle* h, const cilqr_uncertainty_map* map);
int cilqr_clear_uncertainty_map(cilqr_handle* h);

/* Test hook: the map cost alone at n states (host buffers, [n][4]) against the map currently set (solve index 0's layer and
 * pose) → cost[n], vx[n][2] (the x, y entries; the others are zero), mx[n][3] (xx, xy, yy). */
int cilqr_debug_uncertainty_cost(cilqr_handle* h, int n, const double* states, double* cost, double* vx, double* mx);

/* Sampled obstacles (the "uncertainty-aware" batch of BASELINE config 3: n_obs moving obstacles × n_samples Gaussian pose
 * samples, every sample an Obstacle of its own with Parameters::w_obstacle = 1/n_samples, I/Constraints.cpp:177-187).
 * Exactly cilqr_solve_batch(_device) with M = n_obs·n_samples obstacles where obstacle m = o·n_samples + s has
 *   pose[t] = (x_o[t] + dx, y_o[t] + dy, v_o[t], theta_o[t] + dtheta),  dim[t] = dim_o[t],  weight = sample_weight,
 * (dx, dy, dtheta) = sample_offset[b][o][s], but taking the compact form: the materialised tables are n_samples times
 * larger and, not fitting on chip, would be streamed from HBM once per iteration.
 *   nom_pose [B][n_obs][4*N], nom_dim [B][n_obs][2*N], sample_offset [B][n_obs][n_samples][3]
 * n_obs·n_samples counts against max_obstacles of cilqr_create; n_samples ≥ 2.  Sample headings come from the angle-addition
 * formulas and the ellipse semi-axes from reciprocals refined to ≈1 ulp: results agree with the materialised call to ≈1e-12,
 * not bit for bit. */
int cilqr_solve_batch_sampled(cilqr_handle* h, int B, int N, int n_obs, int n_samples,
                              const double* x0, double* U, const double* poly, const double* xplan_fl,
                              const double* nom_pose, const double* nom_dim, const double* sample_offset,
                              double sample_weight, double* X_out, double* J_out, int32_t* iters_out,
                              int32_t* status_out, uint32_t flags);
int cilqr_solve_batch_sampled_device(cilqr_handle* h, void* stream, int B, int N, int n_obs, int n_samples,
                                     const double* x0, double* U, const double* poly, const double* xplan_fl,
                                     const double* nom_pose, const double* nom_dim, const double* sample_offset,
                                     double sample_weight, double* X_out, double* J_out, int32_t* iters_out,
                                     int32_t* status_out, uint32_t flags);

/* Local min-cost selection over a batch resident on the device (strict-< first-minimum tie-break, as in
 * I/Constraints.cpp:50): writes {J_min, (double)index} to out_pair (device, 2 doubles).  The cross-GPU step
 * is cilqr_argmin_global_device below. */
int cilqr_argmin_device(cilqr_handle* h, void* stream, int B, const double* J, double* out_pair);

/* --- the cross-GPU exchange step (SURVEY §8b "Entry point", §8e; new: the reference has no collective) ------------------
 * The batch shards by scene with no data-path collective; the ONE exchange is the min-cost pick: every rank's
 * {J_min, local index, index offset} (24 bytes) through one ncclAllGather (RCCL over xGMI), then the lexicographic minimum
 * on the device, lowest global index winning ties (the strict-< first-minimum convention of I/Constraints.cpp:50).
 *
 * One process per GPU: rank 0 calls cilqr_comm_unique_id (ncclGetUniqueId), the host carries the CILQR_COMM_ID_BYTES to every
 * rank by its own means (MPI_Bcast, a file, torch.distributed's store), every rank calls cilqr_comm_init_rank on its handle
 * (ncclCommInitRank: collective, blocks until all ranks arrive).  cilqr_destroy releases the communicator. */
int cilqr_comm_unique_id(void* id_bytes /* CILQR_COMM_ID_BYTES, out */);
int cilqr_comm_init_rank(cilqr_handle* h, int n_ranks, int rank, const void* id_bytes);
int cilqr_comm_destroy(cilqr_handle* h);
int cilqr_comm_size(const cilqr_handle* h); /* ranks of the handle's communicator; 1 without one */
/* Global min-cost selection, asynchronous on `stream`: argmin of this rank's J[B] (device; B = 0: the rank has no scenes) →
 * all-gather → out_pair (device, 2 doubles) = {J_min, (double)global index}, the same on every rank; global index =
 * index_offset + local index; index -1: no rank had a finite cost.  Without a communicator it is the local pick plus offset.
 * Every rank of the communicator must call it, in the same order relative to its other collective calls. */
int cilqr_argmin_global_device(cilqr_handle* h, void* stream, int B, const double* J, int64_t index_offset, double* out_pair);

/* Test hook: the cross-rank pick alone, over n gathered records {J_min, local index, index offset} given in host memory →
 * out_pair (host, 2 doubles).  Lets the device-side combine rule be checked for several ranks on a one-GPU box. */
int cilqr_debug_select(cilqr_handle* h, int n, const double* triples, double* out_pair);

/* One process driving n_devices GPUs (the host model of the reference: one C++ process): one handle, stream and RCCL
 * communicator per device (ncclCommInitAll); devices = NULL means ordinals 0..n_devices-1.  Mirrors
 * iLQR::iLQR / get_optimal_control_seq like cilqr_create / cilqr_solve_batch do, for a batch that spans devices. */
typedef struct cilqr_multi cilqr_multi;
int cilqr_create_multi(const cilqr_params* p, int max_batch_per_device, int max_horizon, int max_obstacles, int n_devices,
                       const int* devices, cilqr_multi** out);
int cilqr_multi_destroy(cilqr_multi* m);
int cilqr_multi_device_count(const cilqr_multi* m);
cilqr_handle* cilqr_multi_handle(cilqr_multi* m, int i); /* device i's handle, for the *_device entry points */
/* The shard of a batch of B solves that device (or rank) `shard` of `n_shards` owns: contiguous by scene and balanced — the first
 * B mod n shards own one solve more; shards may be empty when B < n.  Host arithmetic only (no device needed): the rule
 * cilqr_multi_solve_batch applies, exported so that a one-process-per-GPU host shards the same way. */
int cilqr_shard_range(int B, int n_shards, int shard, int* first, int* count);
/* cilqr_solve_batch over all devices: host buffers as there, solves sharded by cilqr_shard_range, followed by the exchange
 * step; best_index / best_J (may be NULL) receive the global min-cost pick.  Each device's shard is enqueued (copies in,
 * kernels, copies out, the shard's argmin) by its own host thread, so the devices overlap whatever the caller's memory is:
 * from pageable buffers a hipMemcpyAsync is a synchronous copy, and one thread walking the devices would serialise them; with
 * buffers from cilqr_host_alloc the copies are true DMA transfers on every device's own stream.  On any failure every
 * device's stream is drained before the error is returned — no copy into caller memory is left in flight — and the handles
 * are free for the next call.
 * A device list that names one device several times (shards sharing a GPU: how a one-GPU box rehearses the n-shard path) gets
 * no RCCL communicator — RCCL cannot span a device twice — and gathers the 24-byte records with device copies ordered by
 * events instead; cilqr_multi_uses_rccl tells which. */
int cilqr_multi_solve_batch(cilqr_multi* m, int B, int N, int M, const double* x0, double* U, const double* poly,
                            const double* xplan_fl, const double* obs_pose, const double* obs_dim, const double* obs_weight,
                            double* X_out, double* J_out, int32_t* iters_out, int32_t* status_out, uint32_t flags,
                            int64_t* best_index, double* best_J);
int cilqr_multi_uses_rccl(const cilqr_multi* m); /* 1: distinct devices, exchange by ncclAllGather; 0: shards share a device */
/* Test hook: the nth_call-th host-buffer solve enqueued on `h` from now on fails AFTER its input copies were enqueued
 * (nth_call = 1: the next one; 0 switches the hook off) — the error path of cilqr_solve_batch / cilqr_multi_solve_batch with
 * asynchronous copies in flight. */
int cilqr_debug_fail_enqueue(cilqr_handle* h, int nth_call);

/* Diagnostics (the reference's only tracing is std::chrono around run_step, I/ilqr_uncertainty_node.cpp:117-124): while
 * dev_buf != NULL, solves run a separately compiled, stamped instantiation of the kernel that writes, per solve, 16
 * uint64 to dev_buf[B][16] (device memory owned by the caller): shader-clock totals {prologue, linearise, Riccati, forward,
 * epilogue, #linearise, #Riccati, total}, then (one-wavefront-per-solve family; 0 elsewhere) totals inside the linearisation
 * {cos/sin columns, closest path sample, cost derivatives, record stores, cost reduction} and inside the Riccati steps
 * {products up to Q_uu in scalar registers, determinant and reciprocal, rest of the step}.  NULL restores the production
 * kernel.  Never use it when timing. */
int cilqr_set_diag_buffer(cilqr_handle* h, uint64_t* dev_buf);

/* Measurement hook: while dev_buf != NULL every solve also writes the number of backward + forward passes it actually
 * executed to dev_buf[B] (device int32, owned by the caller).  The production kernels stop at the first rejected iteration
 * (DESIGN.md §4.3), so this is smaller than iters_out, the reference loop's iteration count; bench.py prices its fp64 estimate
 * with it.  NULL switches it off. */
int cilqr_set_pass_count_buffer(cilqr_handle* h, int32_t* dev_buf);

/* Which kernel family a solve of this shape takes on this handle: the number of lanes per solve — 64 = one
 * wavefront per solve, LDS-resident (cilqr_solve.hip); 32 … 1 = the grouped family (cilqr_solve_groups.hip), 64/G solves per
 * wavefront.  The same rule cilqr_solve_batch(_device) applies (CILQR_FORCE_G in the environment at create overrides it);
 * measurement tools label their figures with it instead of restating the rule.  Negative: error code. */
int cilqr_solve_family(const cilqr_handle* h, int B, int N, int M);
/* Wavefronts per solve of cilqr_solve_batch(_device) on the one-wavefront family: 2 or 3 where further wavefronts take the obstacle,
 * Jacobian and control-barrier terms of phase L while the first searches the closest path samples (cilqr_solve_share_kernel, DESIGN.md
 * §4.2: three up to three quarters of a solve per SIMD, at least two obstacles and N ≤ 64, two up to two solves per SIMD and N ≤ 127; obstacle
 * table in LDS — a solve's LDS share grows where fewer solves share a CU; with an uncertainty map set, whose term then goes to the
 * last of the further wavefronts: three up to half a solve per SIMD, two up to one; results bit-identical to the one-wavefront kernel; CILQR_NO_SHARE_KERNEL in the environment at create switches it
 * off, CILQR_SHARE_W = 2 or 3 fixes the number), else 1 (also for every shape cilqr_solve_family sends to the grouped family).
 * CILQR_FLAG_FAITHFUL_ITERS always runs on one.  Negative: error code. */
int cilqr_solve_wavefronts(const cilqr_handle* h, int B, int N, int M);
/* The same for cilqr_solve_batch_sampled(_device): how many wavefronts share a solve's phase L on this handle — 1 (one wavefront
 * per solve: horizons beyond 64, CILQR_NO_SPLIT_KERNEL), 2 or 4 (cilqr_solve_split_kernel, DESIGN.md
 * §4.1c).  CILQR_FLAG_FAITHFUL_ITERS always runs on one.  Negative: error code. */
int cilqr_solve_sampled_wavefronts(const cilqr_handle* h, int B, int N, int n_obs);

/* Test hook: runs the kernels' own regularised Q_uu inverse (I/iLQR.cpp:155-175) on n column-major 2×2 matrices (host
 * buffers).  general = 0: the positive-semi-definite form of the production kernel; 1: the eigenvalue-clamping form of the
 * GENERAL kernel (NaN rows where it reports a non-finite matrix).  Lets the rarely taken branch be checked against the
 * reference's EigenSolver outputs (tests/golden/ref_quu.json) without having to provoke it through a whole solve. */
int cilqr_debug_quu_inverse(cilqr_handle* h, int n, const double* Quu, const double* lamb, double* Qinv, int general);
/* Test hook: the kernels' closest-path-sample search (Constraints::find_closest_point, I/Constraints.cpp:24-59: the strict-< first
 * minimum of the squared distance over all samples; csrc/cilqr_device.hpp::closest_sample finds it from two pruning windows and, where
 * the window is wide and the distance provably convex over it, by Newton) on n independent queries (host buffers).  queries[i] =
 * {poly[6], x_local_plan first, last, point x, y}; out[i] = {the search's index, the index a plain scan over ALL samples of the same
 * kernel finds, 1 if the Newton search decided}.  The two indices must be equal for every query. */
int cilqr_debug_closest_sample(cilqr_handle* h, int n, const double* queries, int32_t* out);

/* Test hook: the blur kernel's own covariance → confidence-ellipse step (float eigen-solve following Eigen::EigenSolver
 * <Matrix2f>, M/src/arbitrary_transformation.cu:60-83 + M/include/ARBIT.cuh:82-99) on n covariances {a, b, c} (host buffers);
 * out = {half_major, half_minor, angle} per row.  Checked bit for bit against the reference's Eigen (ref_blur.json). */
int cilqr_debug_blur_ellipse(cilqr_handle* h, int n, const double* abc, double* out);

/* Blocks until everything the host-pointer entry points enqueued on the handle's own stream has finished. */
int cilqr_wait(cilqr_handle* h);

/* --- costmap warp ----------------------------------------------------------------------------- */
/* Rigid global→vehicle-frame warp (M/src/local_costmap.cpp:242-264): for every destination cell, centre C
 * → g = Rot(theta)·C + (vx, vy) → nearest source cell (GridMap::atPosition, G/grid_map_core/src/GridMap.cpp:160-166);
 * if bbox != NULL and bbox(cell) > 90 the destination takes the bbox value (:260-263).  Where the reference
 * would throw std::out_of_range the destination is set to NaN and counted in *n_out_of_range (may be NULL).
 * src: src_geom.rows×cols float32 column-major; dst/bbox: dst_geom.rows×cols float32 column-major. */
int cilqr_warp_costmap(cilqr_handle* h, const float* src, const cilqr_map_geom* src_geom,
                       float* dst, const cilqr_map_geom* dst_geom,
                       double vx, double vy, double vtheta, const float* bbox, int64_t* n_out_of_range);
/* Device-pointer form; n_out_of_range_dev is a device int64 counter (zeroed by the call) or NULL. */
int cilqr_warp_costmap_device(cilqr_handle* h, void* stream, const float* src, const cilqr_map_geom* src_geom,
                              float* dst, const cilqr_map_geom* dst_geom,
                              double vx, double vy, double vtheta, const float* bbox,
                              int64_t* n_out_of_range_dev);
/* K frames in one launch: the same source map warped for K poses (a backlog of odometry ticks, the node's pose-noise candidates,
 * I/ilqr_uncertainty_node.cpp:82-110) into K destination layers stored back to back (frame k at dst + k*rows*cols).
 * poses: HOST array [K][3] = (vx, vy, vtheta), read before the call returns; bbox (device, one layer shared by the frames) and
 * n_out_of_range_dev (device, K int64 counters, zeroed by the call) may be NULL.  1 <= K <= 1024.  Every frame equals what
 * cilqr_warp_costmap_device gives for its pose, bit for bit. */
int cilqr_warp_costmap_batch_device(cilqr_handle* h, void* stream, const float* src, const cilqr_map_geom* src_geom, float* dst,
                                    const cilqr_map_geom* dst_geom, int K, const double* poses, const float* bbox,
                                    int64_t* n_out_of_range_dev);
/* --- pose-uncertainty propagation over the vehicle-frame costmap ("blur") ---------- */
/* thrust_propagateUncertainty (M/src/arbitrary_transformation.cu:8-157, functors M/include/ARBIT.cuh:51-107) together with
 * the copy-through of LocalCostmap::propagateUncertainty (M/src/local_costmap.cpp:483-496): for every cell with linear
 * (column-major) index ≥ index, the pose-uncertainty covariance at the cell → 95 % confidence ellipse → Gaussian-weighted
 * average of the `src` layer over the cells inside it; an empty ellipse copies the cell through; cells before `index` are
 * NaN.  src/out: float32 column-major layers of geometry g.  count_out (optional): cells inside each ellipse.
 * vtheta: vehicle heading (the reference passes its sine and cosine). */
int cilqr_blur_costmap(cilqr_handle* h, const float* src, const cilqr_map_geom* g, int index, double vtheta, double sigma_x,
                       double sigma_y, double sigma_theta, float* out, int32_t* count_out);
int cilqr_blur_costmap_device(cilqr_handle* h, void* stream, const float* src, const cilqr_map_geom* g, int index, double vtheta,
                              double sigma_x, double sigma_y, double sigma_theta, float* out, int32_t* count_out);

/* --- wire formats either side of the costmap path --------------------------------- */
/* GridMapRosConverter::fromOccupancyGrid's data loop (G/grid_map_ros/src/GridMapRosConverter.cpp:259-266, called at
 * M/src/local_costmap.cpp:169): layer[i] = occ[n-1-i] == -1 ? NaN : (float)occ[n-1-i], i = column-major linear index.
 * n_cells = width*height; the geometry side of the conversion is cilqr_map_geom_set(width*res, height*res, res,
 * origin + length/2). */
int cilqr_occupancy_to_layer(cilqr_handle* h, const int8_t* occ, int64_t n_cells, float* layer);
int cilqr_occupancy_to_layer_device(cilqr_handle* h, void* stream, const int8_t* occ, int64_t n_cells, float* layer);
/* GridMapRosConverter::toOccupancyGrid's data loop (:293-306, called at M/src/local_costmap.cpp:298 with range 0..100) for
 * a layer whose circular-buffer start index is zero: v = (layer[i]-data_min)/(data_max-data_min) in float; NaN → -1, else
 * the truncation of clamp(v,0,1)*100; stored at occ[n-1-i]. */
int cilqr_layer_to_occupancy(cilqr_handle* h, const float* layer, int64_t n_cells, float data_min, float data_max, int8_t* occ);
int cilqr_layer_to_occupancy_device(cilqr_handle* h, void* stream, const float* layer, int64_t n_cells, float data_min,
                                    float data_max, int8_t* occ);

/* One frame of the map node's odometry callback (M/src/local_costmap.cpp:172-305) on the device, asynchronous on `stream`:
 * warp of the global layer into the vehicle frame with the optional bounding-box override (cilqr_warp_costmap_device) →
 * pose-uncertainty blur (cilqr_blur_costmap_device, index 0) → the blurred layer as an OccupancyGrid with range 0..100
 * written by the blur kernel itself.  vehicle_layer and uncertainty_layer (device, rows*cols floats each) receive the two
 * float layers; occupancy_out (device, rows*cols int8) may be NULL. */
int cilqr_costmap_frame_device(cilqr_handle* h, void* stream, const float* global_layer, const cilqr_map_geom* global_geom,
                               const cilqr_map_geom* vehicle_geom, double vx, double vy, double vtheta, const float* bbox,
                               double sigma_x, double sigma_y, double sigma_theta, float* vehicle_layer,
                               float* uncertainty_layer, int8_t* occupancy_out, int64_t* n_out_of_range_dev);

/* setGeometry(Length(lx,ly), res, Position(px,py)) size/length rule (G/grid_map_core/src/GridMap.cpp:45-62). */
int cilqr_map_geom_set(cilqr_map_geom* g, double len_x, double len_y, double res, double pos_x, double pos_y);

#ifdef __cplusplus
}
#endif
#endif /* CILQR_H_ */
