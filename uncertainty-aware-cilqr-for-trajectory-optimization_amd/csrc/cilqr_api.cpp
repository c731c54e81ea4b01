// cilqr_api.cpp — the C-ABI of include/cilqr.h over the HIP kernels.  No CPU fallback: compute entry points
// fail with CILQR_ERR_NO_DEVICE / CILQR_ERR_HIP when the device path is unavailable.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>

#include "cilqr_internal.h"

#include "cilqr_handle.h"

namespace cilqr {
thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}
}  // namespace cilqr

using cilqr::fail;

namespace {

void derive(const cilqr_params& p, cilqr::KParams& k) {
  k.dt = p.timestep;
  k.desired_speed = p.desired_speed;
  k.tolerance = p.tolerance;
  k.w_acc = p.w_acc; k.w_yawrate = p.w_yawrate; k.w_pos = p.w_pos; k.w_vel = p.w_vel; k.w_obstacle = p.w_obstacle;
  k.q1_acc = p.q1_acc; k.q2_acc = p.q2_acc; k.q1_yawrate = p.q1_yawrate; k.q2_yawrate = p.q2_yawrate;
  k.q1_front = p.q1_front; k.q2_front = p.q2_front; k.q1_rear = p.q1_rear; k.q2_rear = p.q2_rear;
  k.acc_max = p.acc_max; k.acc_min = p.acc_min;
  k.yaw_hi = tan(p.steer_angle_max) / p.wheelbase;
  k.yaw_lo = tan(p.steer_angle_min) / p.wheelbase;
  k.half_dt2 = p.timestep * p.timestep / 2.0;
  k.wheelbase = p.wheelbase; k.speed_max = p.speed_max;
  k.t_safe = p.t_safe; k.s_safe_a = p.s_safe_a; k.s_safe_b = p.s_safe_b;
  k.ego_rad = p.ego_rad; k.ego_front = p.ego_front; k.ego_rear = p.ego_rear;
  k.lamb_factor = p.lamb_factor; k.lamb_max = p.lamb_max;
  k.max_iterations = p.max_iterations;
  k.n_samples = p.num_of_local_wpts * 10;
}

int check_sizes(const cilqr_handle* h, int B, int N, int M) {
  if (!h) return fail(CILQR_ERR_ARG, "null handle");
  if (B < 0 || B > h->max_batch) return fail(CILQR_ERR_ARG, "B=%d outside [0,%d]", B, h->max_batch);
  if (N < 1 || N > h->max_horizon) return fail(CILQR_ERR_ARG, "N=%d outside [1,%d]", N, h->max_horizon);
  if (M < 0 || M > h->max_obstacles) return fail(CILQR_ERR_ARG, "M=%d outside [0,%d]", M, h->max_obstacles);
  return CILQR_OK;
}

template <typename T>
hipError_t dmalloc(T** p, size_t n) {
  *p = nullptr;
  if (n == 0) return hipSuccess;
  return hipMalloc((void**)p, n * sizeof(T));
}

}  // namespace

// Kernel family by batch shape: lanes per solve, 64 = the one-wavefront-per-solve family (cilqr_solve.hip), less =
// the grouped family (cilqr_solve_groups.hip).  Drawn from tools/family_shapes.py (profiles/r03_family_shapes.txt: both families
// at N = 30 … 160, M = 0 … 16, B = 2048 … 16384, first calls, i.e. WITHOUT the schedule hint — on a planner's tick sequence
// the hint changes nothing, profiles/r03_schedule_hint_ticks.txt) and tools/group_lanes_sweep.py.  Up to one solve per SIMD the
// wavefront family always (its backward pass on the matrix cores and scalar-path forward pass give it the shorter serial chain);
// beyond, it keeps batches of a few solves per SIMD while a solve is short — the shorter the horizon and the fewer the
// obstacles, the longer — and the grouped family, whose phase L shares the lanes of finished solves since round 3, takes the
// rest.  Very long horizons (N > 110: the records no longer fit the grouped family's LDS chunks well) stay on the wavefront
// family at every size measured.
static int pick_group_lanes(const cilqr_handle* h, int B, int N, int M) {
  const int f = h->force_g;
  if (f == 1 || f == 2 || f == 4 || f == 8 || f == 16 || f == 32 || f == 64) return f;
  // (with hundreds of obstacle entries per step the solve is a stream over its obstacle table: the wavefront-per-solve
  // family reads it as whole 400-640 B rows per instruction and measures ≈2x faster there — BASELINE config 3)
  if (M > 32 || N > 110) return 64;
  // largest batch that stays on the wavefront family, in half solves per SIMD
  int cap2;
  // (redrawn at the end of round 3, when the family had got its shared-phase-L kernel up to two solves per SIMD and N = 127:
  // profiles/r03_family_shapes.txt)
  if (N <= 32) cap2 = 16;
  else if (N <= 56) cap2 = 8;
  else if (N <= 92) cap2 = 4;
  else cap2 = 8;
  if (2L * B <= (long)cap2 * h->simds) return 64;
  int G = 32;
  while (G > 1 && (long)G * B > 64L * h->simds) G >>= 1;
  // not below 2 lanes per solve (4 for horizons beyond one round of lanes): with the lanes of finished solves helping in phase L
  // twice as many, smaller wavefronts — started as the first ones end — beat one wavefront per SIMD with 64 solves and 3-step
  // hand-over chunks each (profiles/r03_group_lanes_sweep.txt: B = 65536, N = 50: G = 2 5.7 ms against 7.1 at G = 1; N = 80:
  // G = 4 16.4-19.6 ms against 20.5-24.6)
  const int g_min = N > 64 ? 4 : 2;
  if (G < g_min) G = g_min;
  return G;
}

// Sampled obstacles: wavefronts per solve that share phase L (0: the one-wavefront kernel); the launcher falls back to one where
// the split kernel does not apply (N > 64, fewer obstacles than wavefronts, a map set, the reference-loop mode).
static int pick_split_wavefronts(const cilqr_handle* h, int B) {
  if (h->split_off) return 0;
  if (h->split_w) return h->split_w;
  return B <= h->simds ? 4 : 2;
}

// Static obstacles on the one-wavefront family: further wavefronts per solve for phase L (cilqr_solve_share_kernel) up to about two solves
// per SIMD — tools/share_ab.py, profiles/r03_share_kernel.txt: config-2 scenes 0.372 against 0.404 ms at B = 256, 0.390 / 0.415 at 1024,
// 0.406 / 0.433 at 2048, level at 3072, slower at 4096 (0.547 / 0.476: the second wavefronts cost residency there).
// Up to THREE QUARTERS of a solve per SIMD three: the obstacle terms on two of them (even / odd entries: obstacle_loop's own two chains),
// Jacobians and control barrier on the last — the solves that decide such a launch are the ones with every obstacle close (B = 256: 0.356
// against 0.362 ms).  Not at one solve per SIMD: three wavefronts of 153 registers fill a SIMD, so a CU holds exactly its four workgroups
// and every unevenness of the dispatch makes one wait for a whole solve (rocprofv3, 61 launches at B = 1024: 395 µs average, 538 µs
// maximum with three; 391 / 418 with two).  0: one wavefront.
static int pick_share(const cilqr_handle* h, int B, int N, int M) {
  // how far beyond one solve per SIMD the further wavefronts pay depends on how many workgroups a CU still holds, i.e. on the horizon
  // (tools/share_ab.py with CILQR_SHARE_MAX_B open, profiles/r03_share_kernel.txt, last section: N = 30 still 6 % ahead at four solves
  // per SIMD, N = 40 8 % at three, N = 50 8 % at two and level at three, N = 56 / 60 5 / 3 % at 1.5 and behind at two, N = 64 ahead at
  // 1.25 and behind at 1.5, N = 80 2 % at one) — in quarters of a solve per SIMD:
  const int q = N <= 32 ? 16 : N <= 44 ? 12 : N <= 52 ? 8 : N <= 60 ? 6 : N <= 64 ? 5 : 4;
  const long cap = h->share_max >= 0 ? (long)h->share_max : (long)q * h->simds / 4;
  if (h->share_off || B > cap) return 0;
  const int w = h->share_w ? h->share_w : (4 * B <= 3 * h->simds ? 3 : 2);
  if (h->unc.layer)  // a map set: its term on the last aux wavefront; two wavefronts per SIMD, so three per solve up to half a solve per SIMD
    return N > 64 ? 2 : h->share_w ? h->share_w : (2 * B <= h->simds ? 3 : (B <= h->simds ? 2 : 0));
  return w == 3 && (M < 2 || N > 64) ? 2 : w;  // (horizons 65 … 127: two steps per lane, built for two wavefronts)
}

// LDS a solve of the one-wavefront family may take with its obstacle table inside.  32 KiB keeps five solves per CU resident — what a
// batch beyond one solve per SIMD needs; a batch of at most k ≤ 4 solves per CU cannot use that residency, and each of its solves may
// as well have 1/k of the CU's 160 KiB (beyond 64 KiB the launcher raises the kernels' limit): the table of up to ≈ 20 obstacles at two
// solves per CU, ≈ 45 at one, then lies in LDS instead of being streamed from the workspace by every pass, and the shape can take the
// shared-phase-L kernel (tools/share_ab.py, N = 50: M = 12 at B = 256 0.485 → 0.370 ms, M = 8 at B = 1024 0.503 → 0.435 ms).
static int lds_table_budget(const cilqr_handle* h, int B) {
  if (h->tab_budget_kb > 0) return h->tab_budget_kb * 1024;  // (environment CILQR_LDS_TABLE_KB at create: A/B hook)
  const int cus = h->simds / 4, k = (B + cus - 1) / cus;
  if (k < 1 || k > 4) return 32 * 1024;
  const int share = (160 * 1024) / k - 2048;
  return share < 32 * 1024 ? 32 * 1024 : share;
}

// The one-wavefront-per-solve family with a schedule hint.  A batch of more solves than SIMDs is dispatched in workgroup order,
// and its launch ends when the last workgroup does: a 20-pass solve that starts among the last costs its full length on top of
// everything else (config-2 scenes at B = 4096: 0.91 ms as given, 0.53 ms with the longest solves first; config 3: 3.8 → 2.4 ms,
// tools/schedule_order.py).  Pass counts are not known in advance — but a planner solves nearly the same scenes tick after tick,
// so each call records its solves' pass counts and a small kernel sorts them into the NEXT call's dispatch order (same batch
// size, same stream; otherwise, and in a first call, the order is the identity).  Any order gives the same results: a solve
// depends on nothing but its own inputs.  CILQR_NO_SCHEDULE_HINT in the environment at create switches it off.
static int launch_wave_scheduled(cilqr_handle* h, cilqr::SolveArgs& a, void* stream) {
  const bool hinted = !h->hint_off && a.B > h->simds;
  a.order = hinted && h->hint_B == a.B && h->hint_stream == stream ? h->d_order : nullptr;
  a.hint_passes = hinted ? h->d_hint_passes : nullptr;
  // CILQR_PAIR_KERNEL: up to one solve per SIMD every solve gets a second wavefront on another SIMD of its CU that linearises the
  // new trajectory behind the forward pass (cilqr_solve_pair_kernel).  Measured slower at every batch size (the
  // second wavefront's work is paid for by the main wavefronts that share its SIMD): an experiment, not the default.
  a.pair = h->pair_on && a.B <= h->simds ? 1 : 0;
  // Default up to two solves per SIMD (share_max solves): further wavefronts per solve take the obstacle, control-barrier and Jacobian
  // terms of phase L while the first searches the closest samples (cilqr_solve_share_kernel; bit-identical results; the launcher
  // falls back where it does not apply: table not in LDS, N > 127, a map set, the reference-loop mode).
  if (!a.pair) a.pair = pick_share(h, a.B, a.N, a.M);
  a.tab_budget = lds_table_budget(h, a.B);
  HIP_TRY(cilqr::launch_solve_wave(a, (hipStream_t)stream));
  if (hinted) {
    HIP_TRY(cilqr::launch_schedule_order(h->d_hint_passes, a.B, h->d_order, (hipStream_t)stream));
    h->hint_B = a.B; h->hint_stream = stream;
  }
  return CILQR_OK;
}

extern "C" {

int cilqr_abi_version(void) { return CILQR_ABI_VERSION; }

int cilqr_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return 0;
  int usable = 0;
  for (int d = 0; d < count; ++d) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++usable;
  }
  return usable;
}

const char* cilqr_last_error(void) { return cilqr::g_last_error.c_str(); }

void cilqr_params_default(cilqr_params* p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  // planning / iLQR (I/Parameters.cpp:6-16)
  p->num_of_local_wpts = 20; p->poly_order = 5; p->desired_speed = 5.0;
  p->timestep = 0.1; p->horizon = 40; p->tolerance = 1e-4; p->max_iterations = 20;
  p->num_states = 4; p->num_ctrls = 2;
  // weights (:19-26)
  p->w_acc = 1.0; p->w_yawrate = 4.0; p->w_pos = 0.65; p->w_vel = 3.0; p->w_obstacle = 1.0; p->w_uncertainty = 1.0;
  // barrier constants (:29-42)
  p->q1_acc = 1.0; p->q2_acc = 1.0; p->q1_yawrate = 1.0; p->q2_yawrate = 1.0;
  p->q1_front = 2.75; p->q2_front = 2.75; p->q1_rear = 2.5; p->q2_rear = 2.5;
  p->q1_uncertainty = 2.5; p->q2_uncertainty = 2.5;
  // limits and vehicle (:45-60)
  p->acc_max = 2.0; p->acc_min = -5.5; p->steer_angle_min = -0.75; p->steer_angle_max = 0.75;
  p->wheelbase = 2.94; p->speed_max = 30.0;
  p->steer_control_max = 1.0; p->steer_control_min = -1.0;
  p->throttle_control_max = 1.0; p->throttle_control_min = -1.0;
  // obstacle model (:63-74)
  p->t_safe = 0.1; p->s_safe_a = 0; p->s_safe_b = 0; p->ego_rad = 1.35;
  p->ego_front = 1.47 + 0.925; p->ego_rear = 1.47 + 0.925;
  p->length = 4.79; p->width = 2.16; p->safe_length = 0.0; p->safe_width = 0.0;
  // I/iLQR.cpp:17-18
  p->lamb_factor = 10; p->lamb_max = 10000;
}

int cilqr_default_control_seq(int N, double* U) {
  if (N < 1 || !U) return fail(CILQR_ERR_ARG, "cilqr_default_control_seq: bad argument");
  const int num_zeros = N / 2;  // I/iLQR.cpp:12
  for (int i = 0; i < N; ++i) {
    U[2 * i] = 0.5;
    U[2 * i + 1] = i < num_zeros ? 0.0 : 0.1;
  }
  return CILQR_OK;
}

int cilqr_create(const cilqr_params* p, int max_batch, int max_horizon, int max_obstacles, int device,
                 cilqr_handle** out) {
  if (!p || !out) return fail(CILQR_ERR_ARG, "cilqr_create: null argument");
  *out = nullptr;
  if (max_batch < 1 || max_horizon < 1 || max_horizon > CILQR_MAX_HORIZON || max_obstacles < 0)
    return fail(CILQR_ERR_ARG, "cilqr_create: sizes out of range (max_horizon ≤ %d)", CILQR_MAX_HORIZON);
  if (p->num_states != CILQR_NX || p->num_ctrls != CILQR_NU || p->poly_order + 1 != CILQR_POLY_COEFFS)
    return fail(CILQR_ERR_UNSUPPORTED, "only num_states=4, num_ctrls=2, poly_order=5 are implemented (reference model, I/Model.cpp)");
  if (p->num_of_local_wpts < 1 || p->num_of_local_wpts > 100 || p->max_iterations < 1)
    return fail(CILQR_ERR_ARG, "cilqr_create: num_of_local_wpts / max_iterations out of range");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(CILQR_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  if (device < 0 || device >= count) return fail(CILQR_ERR_ARG, "device %d not in [0,%d)", device, count);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(CILQR_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(CILQR_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
  if (hipSetDevice(device) != hipSuccess) return fail(CILQR_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);

  cilqr_handle* h = new (std::nothrow) cilqr_handle();
  if (!h) return fail(CILQR_ERR_ARG, "out of host memory");
  memset(h, 0, sizeof(*h));
  h->params = *p;
  derive(*p, h->kp);
  h->device = device;
  h->simds = prop.multiProcessorCount > 0 ? prop.multiProcessorCount * 4 : 1024;
  h->max_batch = max_batch; h->max_horizon = max_horizon; h->max_obstacles = max_obstacles;
  if (const char* fg = getenv("CILQR_FORCE_G")) h->force_g = atoi(fg);
  const size_t B = max_batch, N = max_horizon, M = max_obstacles;
  hipError_t err = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  {  // arena of the host-buffer entry points: the larger of the two layouts that can be asked for (cilqr_host_io.cpp)
    const cilqr::IoLayout plain = cilqr::io_layout(B, N, M, true, 0), sampled = cilqr::io_layout(B, N, M, false, 1);
    h->arena_cap = plain.end > sampled.end ? plain.end : sampled.end;
    if (err == hipSuccess) err = hipMalloc((void**)&h->d_arena, h->arena_cap);
    h->stage_cap = h->arena_cap < ((size_t)1 << 20) ? h->arena_cap : ((size_t)1 << 20);  // pinned: calls up to 1 MiB travel packed
    if (err == hipSuccess) err = hipHostMalloc((void**)&h->stage, h->stage_cap, hipHostMallocDefault);
  }
  const size_t Bpad = (B + 63) / 64 * 64;  // the grouped kernels pad the batch to whole wavefronts
  if (err == hipSuccess) err = dmalloc(&h->d_obs_tab, Bpad * M * N * 6);
  if (err == hipSuccess) err = dmalloc(&h->d_ws, cilqr::solve_groups_ws_doubles(max_batch, max_horizon));
  if (err == hipSuccess) err = dmalloc(&h->d_redo, B);
  if (err == hipSuccess) err = dmalloc(&h->d_hint_passes, B);
  if (err == hipSuccess) err = dmalloc(&h->d_order, B);
  h->hint_B = 0; h->hint_stream = nullptr;
  h->hint_off = getenv("CILQR_NO_SCHEDULE_HINT") != nullptr;
  h->pair_on = getenv("CILQR_PAIR_KERNEL") != nullptr;
  h->steal_off = getenv("CILQR_NO_LANE_SHARING") != nullptr;
  h->split_off = getenv("CILQR_NO_SPLIT_KERNEL") != nullptr;
  h->share_off = getenv("CILQR_NO_SHARE_KERNEL") != nullptr;
  if (const char* kb = getenv("CILQR_LDS_TABLE_KB")) h->tab_budget_kb = atoi(kb);
  h->share_max = -1;  // by horizon (pick_share)
  if (const char* sw = getenv("CILQR_SHARE_W")) h->share_w = atoi(sw) == 3 ? 3 : 2;  // (A/B hook: two or three wavefronts wherever the kernel applies)
  if (const char* sm = getenv("CILQR_SHARE_MAX_B")) h->share_max = atoi(sm);  // (A/B hook: largest batch on the shared-phase-L kernel)
  if (const char* sw = getenv("CILQR_SPLIT_W")) h->split_w = atoi(sw);  // (test hook: 2 or 4 wavefronts per solve)
  if (err == hipSuccess) err = dmalloc(&h->d_pair, (size_t)2);
  if (err == hipSuccess) err = dmalloc(&h->d_triple, (size_t)3);
  if (err == hipSuccess) err = dmalloc(&h->d_gather, (size_t)3);
  h->comm_ranks = 1;
  if (err == hipSuccess) err = dmalloc(&h->d_oob, (size_t)1);
  if (err == hipSuccess) err = dmalloc(&h->d_poses, (size_t)8 * 1024 * 4);
  if (err == hipSuccess) err = dmalloc(&h->d_occ_steps, (size_t)8 * 128);
  if (err == hipSuccess) {  // scratch of cilqr_local_plan_batch for max_batch candidates and a 1024-waypoint path
    void* unused = nullptr;
    const size_t W = (size_t)p->num_of_local_wpts;
    if (cilqr::scratch_bytes(h, cilqr::SCR_PLAN_IO, (B * (4 + CILQR_POLY_COEFFS + 2 + 2 * W)) * sizeof(double) + B * sizeof(int32_t), &unused) != CILQR_OK ||
        cilqr::scratch_bytes(h, cilqr::SCR_PLAN_PATH, 2 * 1024 * sizeof(double), &unused) != CILQR_OK)
      err = hipErrorOutOfMemory;
  }
  if (err != hipSuccess) {
    int rc = fail(CILQR_ERR_HIP, "cilqr_create: device allocation failed: %s", hipGetErrorString(err));
    cilqr_destroy(h);
    return rc;
  }
  *out = h;
  return CILQR_OK;
}

int cilqr_destroy(cilqr_handle* h) {
  if (!h) return CILQR_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  (void)cilqr_comm_destroy(h);
  if (h->stage) (void)hipHostFree(h->stage);
  for (void* p : h->scratch)
    if (p) (void)hipFree(p);
  void* ptrs[] = {h->d_poses, h->d_unc_layer, h->d_triple, h->d_gather, h->d_arena, h->d_obs_tab, h->d_ws, h->d_redo, h->d_hint_passes, h->d_order, h->d_pair, h->d_src, h->d_dst, h->d_bbox, h->d_oob, h->d_occ_steps};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return CILQR_OK;
}

int cilqr_set_diag_buffer(cilqr_handle* h, uint64_t* dev_buf) {
  if (!h) return fail(CILQR_ERR_ARG, "null handle");
  h->diag = (unsigned long long*)dev_buf;
  return CILQR_OK;
}

namespace {
// include/cilqr.h, cilqr_set_uncertainty_map: geometry and footprint constants of the map cost, formed once on the host
int fill_unc(const cilqr_handle* h, const cilqr_uncertainty_map* m, cilqr::UncArgs& u) {
  if (!m || !m->layer) return fail(CILQR_ERR_ARG, "cilqr_set_uncertainty_map: null map or layer");
  const cilqr_map_geom& g = m->geom;
  if (g.rows < 2 || g.cols < 2 || !(g.res > 0.0) || !(g.len_x > 0.0) || !(g.len_y > 0.0))
    return fail(CILQR_ERR_ARG, "cilqr_set_uncertainty_map: bad geometry (needs at least 2x2 cells)");
  if (m->probes_l < 1 || m->probes_w < 1 || m->probes_l > 64 || m->probes_w > 64)
    return fail(CILQR_ERR_ARG, "cilqr_set_uncertainty_map: probes_l / probes_w must be in [1, 64]");
  if (m->layer_stride < 0 || (m->layer_stride > 0 && m->layer_stride < (int64_t)g.rows * g.cols))
    return fail(CILQR_ERR_ARG, "cilqr_set_uncertainty_map: layer_stride smaller than one layer");
  const cilqr_params& p = h->params;
  u.layer = m->layer;
  u.poses = m->poses;
  u.stride = m->layer_stride;
  u.rows = g.rows; u.cols = g.cols; u.nl = m->probes_l; u.nw = m->probes_w;
  u.x_first = g.pos_x + (0.5 * g.len_x - 0.5 * g.res);  // cell (0,0) centre, GridMapMath.cpp:114-127
  u.y_first = g.pos_y + (0.5 * g.len_y - 0.5 * g.res);
  u.inv_res = 1.0 / g.res;
  u.px = m->pose_x; u.py = m->pose_y;
  u.cp = cos(m->pose_theta);  // host libm, like the warp's pose (M/src/local_costmap.cpp:201-202)
  u.sp = sin(m->pose_theta);
  u.la0 = u.nl > 1 ? -0.5 * p.safe_length : 0.0;
  u.la_step = u.nl > 1 ? p.safe_length / (double)(u.nl - 1) : 0.0;
  u.wb0 = u.nw > 1 ? -0.5 * p.safe_width : 0.0;
  u.wb_step = u.nw > 1 ? p.safe_width / (double)(u.nw - 1) : 0.0;
  u.q1 = p.q1_uncertainty; u.q2 = p.q2_uncertainty;
  u.scale = p.w_uncertainty / (double)(u.nl * u.nw);
  return CILQR_OK;
}
}  // namespace

int cilqr_set_uncertainty_map_device(cilqr_handle* h, const cilqr_uncertainty_map* map) {
  if (!h) return fail(CILQR_ERR_ARG, "null handle");
  cilqr::UncArgs u;
  int rc = fill_unc(h, map, u);
  if (rc) return rc;
  h->unc = u;
  return CILQR_OK;
}

int cilqr_set_uncertainty_map(cilqr_handle* h, const cilqr_uncertainty_map* map) {
  if (!h) return fail(CILQR_ERR_ARG, "null handle");
  if (map && (map->layer_stride != 0 || map->poses))
    return fail(CILQR_ERR_ARG, "cilqr_set_uncertainty_map: the host form takes one shared layer (per-solve layers / poses: use the _device form)");
  cilqr::UncArgs u;
  int rc = fill_unc(h, map, u);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(h->device));
  const size_t n = (size_t)map->geom.rows * map->geom.cols;
  if (n > h->unc_layer_cap) {  // grows on the first tick of a larger map only
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->d_unc_layer) HIP_TRY(hipFree(h->d_unc_layer));
    h->d_unc_layer = nullptr; h->unc_layer_cap = 0;
    HIP_TRY(dmalloc(&h->d_unc_layer, n));
    h->unc_layer_cap = n;
  }
  // on the handle's stream: ordered before the solves of the host-buffer entry points, which use the same stream
  HIP_TRY(hipMemcpyAsync(h->d_unc_layer, map->layer, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));  // the caller's buffer is free again, and device-pointer solves on other streams see the layer
  u.layer = h->d_unc_layer;
  h->unc = u;
  return CILQR_OK;
}

int cilqr_clear_uncertainty_map(cilqr_handle* h) {
  if (!h) return fail(CILQR_ERR_ARG, "null handle");
  memset(&h->unc, 0, sizeof(h->unc));
  return CILQR_OK;
}

int cilqr_debug_uncertainty_cost(cilqr_handle* h, int n, const double* states, double* cost, double* vx, double* mx) {
  if (!h || n < 1 || !states || !cost || !vx || !mx) return fail(CILQR_ERR_ARG, "cilqr_debug_uncertainty_cost: bad argument");
  if (!h->unc.layer) return fail(CILQR_ERR_ARG, "cilqr_debug_uncertainty_cost: no uncertainty map is set");
  HIP_TRY(hipSetDevice(h->device));
  void* v = nullptr;
  int rc = cilqr::scratch_bytes(h, cilqr::SCR_DEBUG, sizeof(double) * 10 * (size_t)n, &v);
  if (rc) return rc;
  double* d = (double*)v;
  HIP_TRY(hipMemcpyAsync(d, states, sizeof(double) * 4 * n, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(cilqr::launch_unc_cost(h->unc, n, d, d + 4 * (size_t)n, d + 5 * (size_t)n, d + 7 * (size_t)n, h->stream));
  HIP_TRY(hipMemcpyAsync(cost, d + 4 * (size_t)n, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(vx, d + 5 * (size_t)n, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(mx, d + 7 * (size_t)n, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CILQR_OK;
}

int cilqr_set_pass_count_buffer(cilqr_handle* h, int32_t* dev_buf) {
  if (!h) return fail(CILQR_ERR_ARG, "null handle");
  h->passes = dev_buf;
  return CILQR_OK;
}

int cilqr_debug_quu_inverse(cilqr_handle* h, int n, const double* Quu, const double* lamb, double* Qinv, int general) {
  if (!h || n < 1 || !Quu || !lamb || !Qinv) return fail(CILQR_ERR_ARG, "cilqr_debug_quu_inverse: bad argument");
  HIP_TRY(hipSetDevice(h->device));
  void* v = nullptr;
  int rc = cilqr::scratch_bytes(h, cilqr::SCR_DEBUG, sizeof(double) * 9 * (size_t)n, &v);
  if (rc) return rc;
  double *dq = (double*)v, *dl = dq + 4 * (size_t)n, *dout = dl + n;
  HIP_TRY(hipMemcpyAsync(dq, Quu, sizeof(double) * 4 * n, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(dl, lamb, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(cilqr::launch_quu_inverse(n, dq, dl, dout, general, h->stream));
  HIP_TRY(hipMemcpyAsync(Qinv, dout, sizeof(double) * 4 * n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CILQR_OK;
}

int cilqr_debug_closest_sample(cilqr_handle* h, int n, const double* queries, int32_t* out) {
  if (!h || n < 1 || !queries || !out) return fail(CILQR_ERR_ARG, "cilqr_debug_closest_sample: bad argument");
  HIP_TRY(hipSetDevice(h->device));
  void* v = nullptr;
  int rc = cilqr::scratch_bytes(h, cilqr::SCR_DEBUG, (sizeof(double) * 10 + sizeof(int32_t) * 4) * (size_t)n, &v);
  if (rc) return rc;
  double* din = (double*)v;
  int32_t* dout = reinterpret_cast<int32_t*>(din + 10 * (size_t)n);
  HIP_TRY(hipMemcpyAsync(din, queries, sizeof(double) * 10 * n, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(cilqr::launch_closest_sample(n, h->kp.n_samples, din, dout, h->stream));
  HIP_TRY(hipMemcpyAsync(out, dout, sizeof(int32_t) * 3 * n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CILQR_OK;
}

int cilqr_debug_blur_ellipse(cilqr_handle* h, int n, const double* abc, double* out) {
  if (!h || n < 1 || !abc || !out) return fail(CILQR_ERR_ARG, "cilqr_debug_blur_ellipse: bad argument");
  HIP_TRY(hipSetDevice(h->device));
  void* v = nullptr;
  int rc = cilqr::scratch_bytes(h, cilqr::SCR_DEBUG, sizeof(double) * 6 * (size_t)n, &v);
  if (rc) return rc;
  double *d_in = (double*)v, *d_out = d_in + 3 * (size_t)n;
  HIP_TRY(hipMemcpyAsync(d_in, abc, sizeof(double) * 3 * n, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(cilqr::launch_blur_ellipse(n, d_in, d_out, h->stream));
  HIP_TRY(hipMemcpyAsync(out, d_out, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CILQR_OK;
}

int cilqr_wait(cilqr_handle* h) {
  if (!h) return fail(CILQR_ERR_ARG, "null handle");
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CILQR_OK;
}

int cilqr_solve_family(const cilqr_handle* h, int B, int N, int M) {
  if (!h || B < 0 || N < 1 || M < 0) return fail(CILQR_ERR_ARG, "cilqr_solve_family: bad argument");
  return pick_group_lanes(h, B, N, M);
}

int cilqr_solve_wavefronts(const cilqr_handle* h, int B, int N, int M) {
  if (!h || B < 0 || N < 1 || M < 0) return fail(CILQR_ERR_ARG, "cilqr_solve_wavefronts: bad argument");
  if (pick_group_lanes(h, B, N, M) != 64) return 1;
  const int w = pick_share(h, B, N, M);
  return w && !h->pair_on && cilqr::solve_share_applies(N, M, h->kp.n_samples, lds_table_budget(h, B)) ? w : 1;
}

int cilqr_solve_sampled_wavefronts(const cilqr_handle* h, int B, int N, int n_obs) {
  if (!h || B < 0 || N < 1 || n_obs < 1) return fail(CILQR_ERR_ARG, "cilqr_solve_sampled_wavefronts: bad argument");
  const int w = pick_split_wavefronts(h, B);
  if (w < 2 || N > 64 || n_obs < w) return 1;
  return w >= 4 ? 4 : 2;
}

int cilqr_solve_batch_device(cilqr_handle* h, void* stream, int B, int N, int M, const double* x0, double* U,
                             const double* poly, const double* xplan_fl, const double* obs_pose, const double* obs_dim,
                             const double* obs_weight, double* X_out, double* J_out, int32_t* iters_out,
                             int32_t* status_out, uint32_t flags) {
  int rc = check_sizes(h, B, N, M);
  if (rc) return rc;
  if (B == 0) return CILQR_OK;
  if (!x0 || !U || !poly || !xplan_fl || !X_out) return fail(CILQR_ERR_ARG, "cilqr_solve_batch: null required pointer");
  if (M > 0 && (!obs_pose || !obs_dim)) return fail(CILQR_ERR_ARG, "cilqr_solve_batch: M > 0 but obstacle tables are null");
  cilqr::SolveArgs a;
  a.x0 = x0; a.U = U; a.poly = poly; a.xplan_fl = xplan_fl;
  a.obs_pose = obs_pose; a.obs_dim = obs_dim; a.obs_weight = M > 0 ? obs_weight : nullptr;
  a.X_out = X_out; a.J_out = J_out; a.iters_out = iters_out; a.status_out = status_out;
  a.samp_off = nullptr; a.n_samples = 0; a.samp_w = 0.0;
  a.obs_tab = h->d_obs_tab;
  a.fwd = h->d_ws;  // (the grouped family's workspace: 42·N + 12 doubles per solve ≥ the 16·(N + 1) needed here; never both at once)
  a.order = nullptr; a.hint_passes = nullptr; a.pair = 0; a.tab_budget = 0; a.steal = h->steal_off ? 0 : 1; a.split = 0;
  a.redo = h->d_redo;
  a.diag = h->diag;
  a.passes = h->passes;
  a.unc = h->unc;
  a.B = B; a.N = N; a.M = M; a.flags = flags;
  a.kp = h->kp;
  HIP_TRY(hipSetDevice(h->device));
  const int G = pick_group_lanes(h, B, N, M);
  if (G == 64 && cilqr::solve_lds_bytes(N, h->kp.n_samples) > cilqr::SOLVE_LDS_MAX)
    return fail(CILQR_ERR_UNSUPPORTED, "cilqr_solve_batch: horizon %d needs %zu bytes of LDS per solve (limit %zu)", N,
                cilqr::solve_lds_bytes(N, h->kp.n_samples), cilqr::SOLVE_LDS_MAX);
  if (G == 64) return launch_wave_scheduled(h, a, stream);
  HIP_TRY(cilqr::launch_solve_groups(a, G, h->d_ws, (hipStream_t)stream));
  return CILQR_OK;
}

}  // extern "C"

extern "C" {

int cilqr_solve_batch(cilqr_handle* h, int B, int N, int M, const double* x0, double* U, const double* poly,
                      const double* xplan_fl, const double* obs_pose, const double* obs_dim, const double* obs_weight,
                      double* X_out, double* J_out, int32_t* iters_out, int32_t* status_out, uint32_t flags) {
  int rc = check_sizes(h, B, N, M);
  if (rc) return rc;
  if (B == 0) return CILQR_OK;
  if (!x0 || !U || !poly || !xplan_fl || !X_out) return fail(CILQR_ERR_ARG, "cilqr_solve_batch: null required pointer");
  if (M > 0 && (!obs_pose || !obs_dim)) return fail(CILQR_ERR_ARG, "cilqr_solve_batch: M > 0 but obstacle tables are null");
  cilqr::HostBatch q{B, N, M, 0, x0, U, poly, xplan_fl, obs_pose, obs_dim, obs_weight, nullptr, 0.0, X_out, J_out, iters_out, status_out, flags};
  rc = cilqr::host_solve_enqueue(h, q);  // (on failure: stream drained, handle free again)
  if (rc) return rc;
  return cilqr::host_solve_finish(h);
}

int cilqr_solve_batch_sampled_device(cilqr_handle* h, void* stream, int B, int N, int n_obs, int n_samples, const double* x0,
                                     double* U, const double* poly, const double* xplan_fl, const double* nom_pose,
                                     const double* nom_dim, const double* sample_offset, double sample_weight, double* X_out,
                                     double* J_out, int32_t* iters_out, int32_t* status_out, uint32_t flags) {
  if (n_obs < 1 || n_samples < 2) return fail(CILQR_ERR_ARG, "cilqr_solve_batch_sampled: needs n_obs >= 1 and n_samples >= 2");
  if ((long)n_obs * n_samples > 1 << 20) return fail(CILQR_ERR_ARG, "cilqr_solve_batch_sampled: n_obs * n_samples too large");
  int rc = check_sizes(h, B, N, n_obs * n_samples);  // the equivalent materialised obstacle count
  if (rc) return rc;
  if (B == 0) return CILQR_OK;
  if (!x0 || !U || !poly || !xplan_fl || !X_out || !nom_pose || !nom_dim || !sample_offset)
    return fail(CILQR_ERR_ARG, "cilqr_solve_batch_sampled: null required pointer");
  if (cilqr::solve_lds_bytes(N, h->kp.n_samples) + cilqr::solve_sampled_lds_bytes(n_obs, n_samples) > cilqr::SOLVE_LDS_MAX)
    return fail(CILQR_ERR_UNSUPPORTED, "cilqr_solve_batch_sampled: n_obs * n_samples offset records do not fit LDS beside the solve");
  if (cilqr::solve_sampled_tab_doubles(n_obs, N) > (size_t)h->max_obstacles * 6 * (size_t)h->max_horizon)
    return fail(CILQR_ERR_UNSUPPORTED, "cilqr_solve_batch_sampled: nominal records exceed the obstacle workspace reserved at create");
  cilqr::SolveArgs a;
  a.x0 = x0; a.U = U; a.poly = poly; a.xplan_fl = xplan_fl;
  a.obs_pose = nom_pose; a.obs_dim = nom_dim; a.obs_weight = nullptr;
  a.X_out = X_out; a.J_out = J_out; a.iters_out = iters_out; a.status_out = status_out;
  a.samp_off = sample_offset; a.n_samples = n_samples; a.samp_w = sample_weight;
  a.obs_tab = h->d_obs_tab;  // n_obs·N·8 doubles per solve ≤ the n_obs·n_samples·N·6 reserved for the materialised form
  a.fwd = h->d_ws;
  a.pair = 0; a.tab_budget = 0; a.steal = 0;
  // wavefronts per solve sharing phase L (cilqr_solve_split_kernel): four up to one solve per SIMD, where a shorter pass is all that
  // counts, two beyond (tools/split_ab.py, profiles/r03_split_kernel.txt: B = 256 0.88 / 1.35 / 2.08 ms with 4 / 2 / 1 wavefronts,
  // B = 1024 1.47 / 1.55 / 2.11, B = 4096 3.86 / 3.27 / 3.97, B = 8192 6.81 / 5.48 / 5.68)
  a.split = pick_split_wavefronts(h, B);
  a.redo = h->d_redo;
  a.diag = h->diag;
  a.passes = h->passes;
  a.unc = h->unc;
  a.B = B; a.N = N; a.M = n_obs; a.flags = flags;
  a.kp = h->kp;
  HIP_TRY(hipSetDevice(h->device));
  return launch_wave_scheduled(h, a, stream);  // the LDS-resident family at every batch size
}

int cilqr_solve_batch_sampled(cilqr_handle* h, int B, int N, int n_obs, int n_samples, const double* x0, double* U,
                              const double* poly, const double* xplan_fl, const double* nom_pose, const double* nom_dim,
                              const double* sample_offset, double sample_weight, double* X_out, double* J_out,
                              int32_t* iters_out, int32_t* status_out, uint32_t flags) {
  if (n_obs < 1 || n_samples < 2) return fail(CILQR_ERR_ARG, "cilqr_solve_batch_sampled: needs n_obs >= 1 and n_samples >= 2");
  if ((long)n_obs * n_samples > 1 << 20) return fail(CILQR_ERR_ARG, "cilqr_solve_batch_sampled: n_obs * n_samples too large");
  int rc = check_sizes(h, B, N, n_obs * n_samples);
  if (rc) return rc;
  if (B == 0) return CILQR_OK;
  if (!x0 || !U || !poly || !xplan_fl || !X_out || !nom_pose || !nom_dim || !sample_offset)
    return fail(CILQR_ERR_ARG, "cilqr_solve_batch_sampled: null required pointer");
  cilqr::HostBatch q{B, N, n_obs, n_samples, x0, U, poly, xplan_fl, nom_pose, nom_dim, nullptr, sample_offset, sample_weight, X_out, J_out,
                     iters_out, status_out, flags};
  rc = cilqr::host_solve_enqueue(h, q);  // (on failure: stream drained, handle free again)
  if (rc) return rc;
  return cilqr::host_solve_finish(h);
}

int cilqr_argmin_device(cilqr_handle* h, void* stream, int B, const double* J, double* out_pair) {
  if (!h || !J || !out_pair || B < 1) return fail(CILQR_ERR_ARG, "cilqr_argmin_device: bad argument");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(cilqr::launch_argmin(J, B, out_pair, nullptr, 0.0, (hipStream_t)stream));
  return CILQR_OK;
}

int cilqr_blur_costmap_device(cilqr_handle* h, void* stream, const float* src, const cilqr_map_geom* g, int index, double vtheta,
                              double sigma_x, double sigma_y, double sigma_theta, float* out, int32_t* count_out) {
  if (!h || !src || !g || !out) return fail(CILQR_ERR_ARG, "cilqr_blur_costmap: null argument");
  if (g->rows < 1 || g->cols < 1 || !(g->res > 0.0) || index < 0) return fail(CILQR_ERR_ARG, "cilqr_blur_costmap: bad geometry");
  HIP_TRY(hipSetDevice(h->device));
  cilqr::BlurArgs a;
  a.src = src; a.out = out; a.count_out = count_out;
  a.occ_out = nullptr; a.occ_min = 0.0f; a.occ_den = 100.0f;
  a.g = *g; a.index = index;
  a.sin_t = sin(vtheta);  // host libm, as the caller of the reference does (M/src/local_costmap.cpp:201-202)
  a.cos_t = cos(vtheta);
  a.sigma_x = sigma_x; a.sigma_y = sigma_y; a.sigma_theta = sigma_theta;
  HIP_TRY(cilqr::launch_blur(a, (hipStream_t)stream));
  return CILQR_OK;
}

int cilqr_blur_costmap(cilqr_handle* h, const float* src, const cilqr_map_geom* g, int index, double vtheta, double sigma_x,
                       double sigma_y, double sigma_theta, float* out, int32_t* count_out) {
  if (!h || !src || !g || !out) return fail(CILQR_ERR_ARG, "cilqr_blur_costmap: null argument");
  if (g->rows < 1 || g->cols < 1) return fail(CILQR_ERR_ARG, "cilqr_blur_costmap: bad geometry");
  HIP_TRY(hipSetDevice(h->device));
  const size_t n = (size_t)g->rows * g->cols;
  if (n > h->src_cap) {
    if (h->d_src) HIP_TRY(hipFree(h->d_src));
    h->d_src = nullptr; h->src_cap = 0;
    HIP_TRY(dmalloc(&h->d_src, n));
    h->src_cap = n;
  }
  if (n > h->dst_cap) {
    if (h->d_dst) HIP_TRY(hipFree(h->d_dst));
    h->d_dst = nullptr; h->dst_cap = 0;
    HIP_TRY(dmalloc(&h->d_dst, n));
    h->dst_cap = n;
  }
  void* v_cnt = nullptr;
  if (count_out) {
    int rcs = cilqr::scratch_bytes(h, cilqr::SCR_COUNT, n * sizeof(int32_t), &v_cnt);
    if (rcs) return rcs;
  }
  int32_t* d_cnt = (int32_t*)v_cnt;
  hipStream_t s = h->stream;
  HIP_TRY(hipMemcpyAsync(h->d_src, src, n * sizeof(float), hipMemcpyHostToDevice, s));
  int rc = cilqr_blur_costmap_device(h, s, h->d_src, g, index, vtheta, sigma_x, sigma_y, sigma_theta, h->d_dst, d_cnt);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(out, h->d_dst, n * sizeof(float), hipMemcpyDeviceToHost, s));
  if (count_out) HIP_TRY(hipMemcpyAsync(count_out, d_cnt, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return CILQR_OK;
}

int cilqr_local_plan_batch_device(cilqr_handle* h, void* stream, int B, int P, const double* path, int64_t path_stride,
                                  const double* ego, double* poly, double* xplan_fl, double* ref_traj, int32_t* n_out) {
  if (!h || !path || !ego || !poly || !xplan_fl) return fail(CILQR_ERR_ARG, "cilqr_local_plan_batch: null argument");
  if (B < 1 || P < 1 || path_stride < 0) return fail(CILQR_ERR_ARG, "cilqr_local_plan_batch: bad size");
  const cilqr_params& p = h->params;
  if (p.num_of_local_wpts < 1 || p.poly_order < 0 || p.poly_order + 1 > CILQR_POLY_COEFFS)
    return fail(CILQR_ERR_UNSUPPORTED, "cilqr_local_plan_batch: poly_order must be in [0, 5]");
  if (cilqr::local_plan_lds_bytes(p.num_of_local_wpts, p.poly_order + 1) > 64 * 1024)
    return fail(CILQR_ERR_UNSUPPORTED, "cilqr_local_plan_batch: num_of_local_wpts too large for the per-candidate LDS slot");
  HIP_TRY(hipSetDevice(h->device));
  cilqr::LocalPlanArgs a;
  a.path = path; a.path_stride = path_stride; a.ego = ego;
  a.poly = poly; a.xplan_fl = xplan_fl; a.ref_traj = ref_traj; a.n_out = n_out;
  a.B = B; a.P = P; a.n_wpts = p.num_of_local_wpts; a.cols = p.poly_order + 1;
  HIP_TRY(cilqr::launch_local_plan(a, (hipStream_t)stream));
  return CILQR_OK;
}

int cilqr_local_plan_batch(cilqr_handle* h, int B, int P, const double* path, int64_t path_stride, const double* ego,
                           double* poly, double* xplan_fl, double* ref_traj, int32_t* n_out) {
  if (!h || !path || !ego || !poly || !xplan_fl) return fail(CILQR_ERR_ARG, "cilqr_local_plan_batch: null argument");
  if (B < 1 || P < 1 || path_stride < 0) return fail(CILQR_ERR_ARG, "cilqr_local_plan_batch: bad size");
  HIP_TRY(hipSetDevice(h->device));
  const size_t W = h->params.num_of_local_wpts, b = B;
  const size_t n_path = path_stride ? (size_t)(B - 1) * path_stride + 2 * (size_t)P : 2 * (size_t)P;
  // device scratch owned by the handle (reserved at create for max_batch candidates; the path block grows with the first long path)
  void *vp = nullptr, *vio = nullptr;
  int rc = cilqr::scratch_bytes(h, cilqr::SCR_PLAN_PATH, n_path * sizeof(double), &vp);
  if (rc) return rc;
  const size_t o_ego = 0, o_poly = o_ego + b * 4, o_fl = o_poly + b * CILQR_POLY_COEFFS, o_ref = o_fl + b * 2, o_n = o_ref + b * 2 * W;
  rc = cilqr::scratch_bytes(h, cilqr::SCR_PLAN_IO, o_n * sizeof(double) + b * sizeof(int32_t), &vio);
  if (rc) return rc;
  double* d_path = (double*)vp;
  double* io = (double*)vio;
  int32_t* d_n = (int32_t*)(io + o_n);
  hipStream_t s = h->stream;
  if (ref_traj) HIP_TRY(hipMemsetAsync(io + o_ref, 0, b * 2 * W * sizeof(double), s));
  HIP_TRY(hipMemcpyAsync(d_path, path, n_path * sizeof(double), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(io + o_ego, ego, b * 4 * sizeof(double), hipMemcpyHostToDevice, s));
  rc = cilqr_local_plan_batch_device(h, s, B, P, d_path, path_stride, io + o_ego, io + o_poly, io + o_fl, ref_traj ? io + o_ref : nullptr,
                                     n_out ? d_n : nullptr);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(poly, io + o_poly, b * CILQR_POLY_COEFFS * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(xplan_fl, io + o_fl, b * 2 * sizeof(double), hipMemcpyDeviceToHost, s));
  if (ref_traj) HIP_TRY(hipMemcpyAsync(ref_traj, io + o_ref, b * 2 * W * sizeof(double), hipMemcpyDeviceToHost, s));
  if (n_out) HIP_TRY(hipMemcpyAsync(n_out, d_n, b * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return CILQR_OK;
}

int cilqr_occupancy_to_layer_device(cilqr_handle* h, void* stream, const int8_t* occ, int64_t n_cells, float* layer) {
  if (!h || n_cells < 0 || (n_cells > 0 && (!occ || !layer))) return fail(CILQR_ERR_ARG, "cilqr_occupancy_to_layer: bad argument");
  if (n_cells == 0) return CILQR_OK;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(cilqr::launch_occ_to_layer(occ, layer, (long)n_cells, (hipStream_t)stream));
  return CILQR_OK;
}

int cilqr_layer_to_occupancy_device(cilqr_handle* h, void* stream, const float* layer, int64_t n_cells, float data_min,
                                    float data_max, int8_t* occ) {
  if (!h || n_cells < 0 || (n_cells > 0 && (!occ || !layer))) return fail(CILQR_ERR_ARG, "cilqr_layer_to_occupancy: bad argument");
  if (n_cells == 0) return CILQR_OK;
  HIP_TRY(hipSetDevice(h->device));
  // the step table is rebuilt on the call's stream; eight slots in rotation keep calls in flight on other streams apart
  float* steps = h->d_occ_steps + 128 * (h->occ_slot++ & 7);
  HIP_TRY(cilqr::launch_layer_to_occ(layer, occ, (long)n_cells, data_min, data_max, steps, (hipStream_t)stream));
  return CILQR_OK;
}

namespace {
// host-buffer conversions: staged through the handle's scratch slots
int convert_host(cilqr_handle* h, const void* in, size_t in_bytes, void* out, size_t out_bytes, bool to_layer, int64_t n, float lo, float hi) {
  HIP_TRY(hipSetDevice(h->device));
  if (n == 0) return CILQR_OK;
  void *d_in = nullptr, *d_out = nullptr;
  int rc = cilqr::scratch_bytes(h, cilqr::SCR_CONV_IN, in_bytes, &d_in);
  if (rc == CILQR_OK) rc = cilqr::scratch_bytes(h, cilqr::SCR_CONV_OUT, out_bytes, &d_out);
  if (rc) return rc;
  hipStream_t s = h->stream;
  HIP_TRY(hipMemcpyAsync(d_in, in, in_bytes, hipMemcpyHostToDevice, s));
  rc = to_layer ? cilqr_occupancy_to_layer_device(h, s, (const int8_t*)d_in, n, (float*)d_out)
                : cilqr_layer_to_occupancy_device(h, s, (const float*)d_in, n, lo, hi, (int8_t*)d_out);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return CILQR_OK;
}
}  // namespace

int cilqr_occupancy_to_layer(cilqr_handle* h, const int8_t* occ, int64_t n_cells, float* layer) {
  if (!h || n_cells < 0 || (n_cells > 0 && (!occ || !layer))) return fail(CILQR_ERR_ARG, "cilqr_occupancy_to_layer: bad argument");
  if (n_cells == 0) return CILQR_OK;
  return convert_host(h, occ, (size_t)n_cells, layer, (size_t)n_cells * sizeof(float), true, n_cells, 0.f, 0.f);
}

int cilqr_layer_to_occupancy(cilqr_handle* h, const float* layer, int64_t n_cells, float data_min, float data_max, int8_t* occ) {
  if (!h || n_cells < 0 || (n_cells > 0 && (!occ || !layer))) return fail(CILQR_ERR_ARG, "cilqr_layer_to_occupancy: bad argument");
  if (n_cells == 0) return CILQR_OK;
  return convert_host(h, layer, (size_t)n_cells * sizeof(float), occ, (size_t)n_cells, false, n_cells, data_min, data_max);
}

int cilqr_costmap_frame_device(cilqr_handle* h, void* stream, const float* global_layer, const cilqr_map_geom* global_geom,
                               const cilqr_map_geom* vehicle_geom, double vx, double vy, double vtheta, const float* bbox,
                               double sigma_x, double sigma_y, double sigma_theta, float* vehicle_layer,
                               float* uncertainty_layer, int8_t* occupancy_out, int64_t* n_out_of_range_dev) {
  if (!h || !global_layer || !global_geom || !vehicle_geom || !vehicle_layer || !uncertainty_layer)
    return fail(CILQR_ERR_ARG, "cilqr_costmap_frame: null argument");
  int rc = cilqr_warp_costmap_device(h, stream, global_layer, global_geom, vehicle_layer, vehicle_geom, vx, vy, vtheta, bbox,
                                     n_out_of_range_dev);
  if (rc) return rc;
  if (vehicle_geom->rows < 1 || vehicle_geom->cols < 1 || !(vehicle_geom->res > 0.0)) return fail(CILQR_ERR_ARG, "cilqr_costmap_frame: bad geometry");
  cilqr::BlurArgs a;
  a.src = vehicle_layer; a.out = uncertainty_layer; a.count_out = nullptr;
  a.occ_out = occupancy_out; a.occ_min = 0.0f; a.occ_den = 100.0f - 0.0f;  // toOccupancyGrid(..., 0, 100, ...) (M/src/local_costmap.cpp:298)
  a.g = *vehicle_geom; a.index = 0;
  a.sin_t = sin(vtheta); a.cos_t = cos(vtheta);
  a.sigma_x = sigma_x; a.sigma_y = sigma_y; a.sigma_theta = sigma_theta;
  HIP_TRY(cilqr::launch_blur(a, (hipStream_t)stream));
  return CILQR_OK;
}

int cilqr_map_geom_set(cilqr_map_geom* g, double len_x, double len_y, double res, double pos_x, double pos_y) {
  if (!g || !(len_x > 0.0) || !(len_y > 0.0) || !(res > 0.0)) return fail(CILQR_ERR_ARG, "cilqr_map_geom_set: bad argument");
  // GridMap::setGeometry (G/grid_map_core/src/GridMap.cpp:45-62)
  g->rows = (int)round(len_x / res);
  g->cols = (int)round(len_y / res);
  g->res = res;
  g->len_x = (double)g->rows * res;
  g->len_y = (double)g->cols * res;
  g->pos_x = pos_x;
  g->pos_y = pos_y;
  return CILQR_OK;
}

int cilqr_warp_costmap_device(cilqr_handle* h, void* stream, const float* src, const cilqr_map_geom* sg, float* dst,
                              const cilqr_map_geom* dg, double vx, double vy, double vtheta, const float* bbox,
                              int64_t* n_oob_dev) {
  if (!h || !src || !sg || !dst || !dg) return fail(CILQR_ERR_ARG, "cilqr_warp_costmap: null argument");
  if (sg->rows < 1 || sg->cols < 1 || dg->rows < 1 || dg->cols < 1 || !(sg->res > 0.0) || !(dg->res > 0.0))
    return fail(CILQR_ERR_ARG, "cilqr_warp_costmap: bad geometry");
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(h->device));
  cilqr::WarpArgs a;
  a.src = src; a.dst = dst; a.bbox = bbox;
  a.n_oob = (unsigned long long*)n_oob_dev;
  a.sg = *sg; a.dg = *dg;
  a.vx = vx; a.vy = vy;
  a.sin_t = sin(vtheta);  // host libm, as the reference (M/src/local_costmap.cpp:201-202)
  a.cos_t = cos(vtheta);
  if (n_oob_dev) HIP_TRY(hipMemsetAsync(n_oob_dev, 0, sizeof(int64_t), s));
  HIP_TRY(cilqr::launch_warp(a, s));
  return CILQR_OK;
}

int cilqr_warp_costmap_batch_device(cilqr_handle* h, void* stream, const float* src, const cilqr_map_geom* sg, float* dst,
                                    const cilqr_map_geom* dg, int K, const double* poses, const float* bbox, int64_t* n_oob_dev) {
  if (!h || !src || !sg || !dst || !dg || !poses) return fail(CILQR_ERR_ARG, "cilqr_warp_costmap_batch: null argument");
  if (K < 1 || K > 1024) return fail(CILQR_ERR_ARG, "cilqr_warp_costmap_batch: K=%d outside [1,1024]", K);
  if (sg->rows < 1 || sg->cols < 1 || dg->rows < 1 || dg->cols < 1 || !(sg->res > 0.0) || !(dg->res > 0.0))
    return fail(CILQR_ERR_ARG, "cilqr_warp_costmap_batch: bad geometry");
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(h->device));
  if (dg->rows % 4 != 0) {  // the 16-byte-store kernel needs rows in fours: frame by frame with the single-frame kernel
    const size_t cells = (size_t)dg->rows * dg->cols;
    for (int k = 0; k < K; ++k) {
      int rc = cilqr_warp_costmap_device(h, stream, src, sg, dst + (size_t)k * cells, dg, poses[3 * k], poses[3 * k + 1], poses[3 * k + 2], bbox,
                                         n_oob_dev ? n_oob_dev + k : nullptr);
      if (rc) return rc;
    }
    return CILQR_OK;
  }
  double table[1024 * 4];
  for (int k = 0; k < K; ++k) {
    table[4 * k] = poses[3 * k];
    table[4 * k + 1] = poses[3 * k + 1];
    table[4 * k + 2] = sin(poses[3 * k + 2]);  // host libm, as the reference (M/src/local_costmap.cpp:201-202)
    table[4 * k + 3] = cos(poses[3 * k + 2]);
  }
  // eight device slots in rotation keep the tables of calls still in flight (on other streams) apart; the copy itself is
  // staged by the runtime before this call returns (pageable source)
  double* d_table = h->d_poses + (size_t)1024 * 4 * (h->pose_slot++ & 7);
  HIP_TRY(hipMemcpyAsync(d_table, table, sizeof(double) * 4 * K, hipMemcpyHostToDevice, s));
  if (n_oob_dev) HIP_TRY(hipMemsetAsync(n_oob_dev, 0, sizeof(int64_t) * K, s));
  cilqr::WarpBatchArgs a;
  a.src = src; a.dst = dst; a.bbox = bbox;
  a.n_oob = (unsigned long long*)n_oob_dev;
  a.poses = d_table;
  a.sg = *sg; a.dg = *dg;
  HIP_TRY(cilqr::launch_warp_batch(a, K, s));
  return CILQR_OK;
}

int cilqr_warp_costmap(cilqr_handle* h, const float* src, const cilqr_map_geom* sg, float* dst, const cilqr_map_geom* dg,
                       double vx, double vy, double vtheta, const float* bbox, int64_t* n_out_of_range) {
  if (!h || !src || !sg || !dst || !dg) return fail(CILQR_ERR_ARG, "cilqr_warp_costmap: null argument");
  if (sg->rows < 1 || sg->cols < 1 || dg->rows < 1 || dg->cols < 1) return fail(CILQR_ERR_ARG, "cilqr_warp_costmap: bad geometry");
  HIP_TRY(hipSetDevice(h->device));
  const size_t ns = (size_t)sg->rows * sg->cols, nd = (size_t)dg->rows * dg->cols;
  if (ns > h->src_cap) {
    if (h->d_src) HIP_TRY(hipFree(h->d_src));
    h->d_src = nullptr; h->src_cap = 0;
    HIP_TRY(dmalloc(&h->d_src, ns));
    h->src_cap = ns;
  }
  if (nd > h->dst_cap) {
    if (h->d_dst) HIP_TRY(hipFree(h->d_dst));
    h->d_dst = nullptr; h->dst_cap = 0;
    HIP_TRY(dmalloc(&h->d_dst, nd));
    h->dst_cap = nd;
  }
  if (bbox && nd > h->bbox_cap) {
    if (h->d_bbox) HIP_TRY(hipFree(h->d_bbox));
    h->d_bbox = nullptr; h->bbox_cap = 0;
    HIP_TRY(dmalloc(&h->d_bbox, nd));
    h->bbox_cap = nd;
  }
  hipStream_t s = h->stream;
  HIP_TRY(hipMemcpyAsync(h->d_src, src, ns * sizeof(float), hipMemcpyHostToDevice, s));
  if (bbox) HIP_TRY(hipMemcpyAsync(h->d_bbox, bbox, nd * sizeof(float), hipMemcpyHostToDevice, s));
  int rc = cilqr_warp_costmap_device(h, s, h->d_src, sg, h->d_dst, dg, vx, vy, vtheta, bbox ? h->d_bbox : nullptr,
                                     (int64_t*)h->d_oob);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(dst, h->d_dst, nd * sizeof(float), hipMemcpyDeviceToHost, s));
  unsigned long long oob = 0;
  HIP_TRY(hipMemcpyAsync(&oob, h->d_oob, sizeof(oob), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (n_out_of_range) *n_out_of_range = (int64_t)oob;
  return CILQR_OK;
}

}  // extern "C"
