// cilqr_solve.hip — batched constrained-iLQR solve for gfx950 (MI355X), ONE WAVEFRONT PER SOLVE, LDS-resident.
// This is the kernel family for batches up to about one solve per SIMD (B ≤ 1024 on the 1024 SIMDs of an MI355X, e.g.
// BASELINE config 2); larger batches go to the G-lanes-per-solve family in cilqr_solve_groups.hip (launch_solve picks).
//
// Hot path of the reference planner: iLQR::get_optimal_control_seq (I/iLQR.cpp:201-245) with everything it
// calls — nominal rollout (:51-62), Constraints::get_state_cost / get_control_cost / get_J
// (I/Constraints.cpp:145-227, 86-137, 534-561), Obstacle::get_obstalce_cost (I/Obstacle.cpp:39-112),
// Model::get_A_matrix / get_B_matrix (I/Model.cpp:100-155), the backward Riccati recursion (I/iLQR.cpp:133-191)
// and the forward pass (:68-86).  I/ = CILQR/src/ilqr/include/ilqr/ of the reference.
//
// Mapping: workgroup = one 64-lane wavefront = one solve, whole ≤20-iteration loop inside one
// launch.  Everything a solve touches between its first load and its last store lives in LDS:
//   samp  [S]           ordinates of the S = 200 path samples (they depend only on poly / x_local_plan, I/Constraints.cpp:28-42;
//                       the abscissae are equispaced and re-formed with one fma where needed)
//   X a/b [(N+1)][6]    state records {x, y, v, theta, cos theta, sin theta}, double-buffered (X / X_new)
//   U a/b [N][2]        controls, double-buffered (U / U_new)
//   rec   [N][14]       per-step linearisation {l_x(3), l_xx(3), l_u(2), l_uu(2), A/B entries(4)} (16 with p, q in the GENERAL kernel)
//   kK                  feed-forward k and feedback K of the backward pass: stored over the record of their step
//   cst   [22]          constant entries of the matrix operands of phase R
//   tab   [M][N][6]     obstacle table (when it fits; else the same layout in a global workspace)
// Phases per iteration:
//   L  lanes = timesteps: closest path sample, tracking + obstacle + control barrier derivatives, A/B entries,
//      the stage cost of get_J, wavefront-shuffle reduction of J;
//   R  backward Riccati recursion, sequential in t: five v_mfma_f64_4x4x4_4b_f64 per step, the value function held as 4×4 blocks
//      across the lanes, operands fetched from the records by per-lane LDS addresses (riccati_mfma); the GENERAL kernel
//      evaluates the same recursion entry by entry on the vector ALU (riccati);
//   F  forward pass, sequential in t, every lane computing the same values.
#include "cilqr_device.hpp"

namespace cilqr {

using namespace dev;

namespace {

constexpr int RECF = REC - 2;  // record width of the production kernel: p and q are not stored (they are (dt/2)·al, (dt/2)·be)
// Forward-pass record of the production kernel, in GLOBAL memory (SolveArgs::fwd, [N + 1][FREC] per solve, the last one a dump):
// {k(2), K(2x4)} written by phase R, {x, y, v, theta, u0, u1} of the old trajectory written by phase L; phase F reads a step's
// record with two s_load_dwordx16 (forward_smem).
constexpr int FREC = 16;

__device__ __forceinline__ double readfirstlane_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readfirstlane(lo);
  hi = __builtin_amdgcn_readfirstlane(hi);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}

// Sum over the wavefront, the same value in every lane's result (and wave-uniform).  Inside a row of 16 lanes by four DPP
// butterflies (lane ^ 1, lane ^ 2, mirrored within 8, mirrored within 16: after each, all lanes of the growing group hold the
// group's sum), the four row sums by v_readlane: ≈ 24 instructions, against six dependent ds_bpermute round trips of ≈ 120
// ticks each for the shuffle form above (phase L's cost reduction: ≈ 700 → ≈ 130 ticks per call).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_any_f64(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_uniform(double v) {
  v += dpp_f64<0xB1>(v);   // quad_perm [1, 0, 3, 2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2, 3, 0, 1]
  v += dpp_f64<0x141>(v);  // row_half_mirror
  v += dpp_f64<0x140>(v);  // row_mirror
  return (readlane_any_f64(v, 0) + readlane_any_f64(v, 16)) + (readlane_any_f64(v, 32) + readlane_any_f64(v, 48));
}
__device__ __forceinline__ double wave_max_uniform(double v) {  // the same scheme for the maximum (NaN-free inputs)
  v = fmax(v, dpp_f64<0xB1>(v));
  v = fmax(v, dpp_f64<0x4E>(v));
  v = fmax(v, dpp_f64<0x141>(v));
  v = fmax(v, dpp_f64<0x140>(v));
  return fmax(fmax(readlane_any_f64(v, 0), readlane_any_f64(v, 16)), fmax(readlane_any_f64(v, 32), readlane_any_f64(v, 48)));
}

// closest_sample's Newton search (cilqr_device.hpp) needs the polynomial's coefficients — into LDS, slots CST_PC … + 5 of `cst`, which
// riccati_mfma's constants leave free — and an upper bound of its second derivative over the samples (returned; wave-uniform).
constexpr int CST_PC = 6;
__device__ __forceinline__ double path_curvature_wave(const SampleGrid& grid, const double* pc, int S, int lane, double* cst) {
  double A = 0.0, B = 0.0;
  for (int q = lane; q < S; q += WAVE) {
    double a2, a3;
    path_curvature_terms(pc, fma(grid.dxs, (double)q, grid.xf), a2, a3);
    A = fmax(A, a2 == a2 ? a2 : __builtin_huge_val());
    B = fmax(B, a3 == a3 ? a3 : __builtin_huge_val());
  }
  A = wave_max_uniform(A);
  B = wave_max_uniform(B);
  if (lane < CILQR_POLY_COEFFS) cst[CST_PC + lane] = pc[lane];
  return path_curvature_bound(grid, pc, S, A, B);
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, WAVE));
  return v;
}

__device__ __forceinline__ void store_state(double* X, int t, const State& s) {
  double* r = X + t * XR;
  r[0] = s.x; r[1] = s.y; r[2] = s.v; r[3] = s.th; r[4] = s.c; r[5] = s.s;
}

struct LdsSamples {  // sample accessor over the LDS copy: ordinate stored, abscissa re-formed exactly as sample_xy forms it
  const double* samp;
  double xf, dxs;
  __device__ __forceinline__ void operator()(int s, double& x, double& y) const { x = fma(dxs, (double)s, xf); y = samp[s]; }
};

struct TabObstacles {  // obstacle accessor over the [m][t][field] table (LDS or global)
  const double* tab;   // already offset by t: entry of obstacle 0 at this lane's step
  const double* wts;
  int N;
  double w_default;
  // An entry is 48 contiguous bytes, read as three 16-byte accesses.  In LDS the lane stride of 12 dwords makes each
  // quarter-wave's b128 reads cover all 64 banks exactly once (12·i mod 64, i < 16, are 16 distinct multiples of 4).
  __device__ __forceinline__ bool operator()(int m, ObsEntry& e, double& w) const {
    const double2* p = reinterpret_cast<const double2*>(tab + (size_t)m * TABF * N);
    const double2 a = p[0], b = p[1], c = p[2];
    e.ox = a.x; e.oy = a.y; e.co = b.x; e.so = b.y; e.ia2 = c.x; e.ib2 = c.y;
    w = wts ? wts[m] : w_default;
    return true;
  }
};

struct TabObstaclesStrided {  // the same over the entries first, first + stride, … (a wavefront's share: cilqr_solve_share_kernel)
  const double* tab;
  const double* wts;
  int N, first, stride;
  double w_default;
  __device__ __forceinline__ bool operator()(int j, ObsEntry& e, double& w) const {
    const int m = first + j * stride;
    const double2* p = reinterpret_cast<const double2*>(tab + (size_t)m * TABF * N);
    const double2 a = p[0], b = p[1], c = p[2];
    e.ox = a.x; e.oy = a.y; e.co = b.x; e.so = b.y; e.ia2 = c.x; e.ib2 = c.y;
    w = wts ? wts[m] : w_default;
    return true;
  }
};

template <bool STREAMED>
struct TabSource {  // every obstacle has its own table row; STREAMED: the table lies in global memory (lin_step's PAIRED mode)
  static constexpr bool kPaired = STREAMED;
  const double* tab;
  const double* wts;
  int N;
  double w_default;
  __device__ __forceinline__ TabObstacles at(int t) const { return TabObstacles{tab + (size_t)t * TABF, wts, N, w_default}; }
};

// Sampled obstacles (BASELINE config 3: n_obs moving obstacles × S pose samples, sample = nominal trajectory + a constant
// (dx, dy, dtheta)): entry (o, s) at step t is derived from the nominal record of (o, t) and the offset record of (o, s)
// instead of being read from a materialised n_obs·S·N table — that table is 614 KB per solve at config 3 and was
// re-streamed from HBM by every iteration (measured ≈ 20 GB per launch of 4096 solves).  The nominal records (8 doubles per
// (o, t)) stay in global memory and are read once per obstacle and iteration; the offset records live in LDS and are read
// at a wave-uniform address.  cos/sin of the sample heading come from the angle-addition formulas.
constexpr int NOMF = 8;  // x, y, cos, sin, v·t_safe, half-length + margins, half-width + margins, pad
constexpr int OFFF = 6;  // dx, dy, then (cos, sin of the SAMPLE's heading, 1/a², 1/b²) for an obstacle of constant shape, else (cos dtheta, sin dtheta, -, -)
struct SampledObstacles {
  const double* nom;  // this lane's step: record of obstacle 0; obstacle stride N·NOMF
  const double* off;  // LDS [o][s][OFFF]
  int N, S, n_obs;
  double w;
  int o, s;
  int o_stride;                            // 1; 2 where two wavefronts share a solve's obstacles (cilqr_solve_split_kernel)
  bool far;                                // the obstacle in use is negligible at every step of this wavefront (see below)
  const double* rmax;                      // LDS [o]: largest |(dx, dy)| over the obstacle's samples; then [n_obs + o]: constant-shape flag
  bool cshape;                             // the obstacle in use keeps heading, speed and dimensions over the horizon (see the prologue)
  double fx, fy, rx, ry, kf, kr;           // this step's two ego circle centres; sqrt(1 + 64/q2) for either circle
  double x, y, c0, s0, vt, ha, hb;         // nominal record in use
  double nx, ny, nc0, ns0, nvt, nha, nhb;  // the next obstacle's, requested one obstacle (S entries) ahead
  __device__ __forceinline__ void request(int oo) {
    const double2* p = reinterpret_cast<const double2*>(nom + (size_t)oo * NOMF * N);
    const double2 a = p[0], b = p[1], c = p[2], d = p[3];
    nx = a.x; ny = a.y; nc0 = b.x; ns0 = b.y; nvt = c.x; nha = c.y; nhb = d.x;
  }
  // called with m = 0, 1, 2, … in order (lin_step does); false: the entry is negligible, nothing was written
  __device__ __forceinline__ bool operator()(int, ObsEntry& e, double& wout) {
    if (s == 0) {
      x = nx; y = ny; c0 = nc0; s0 = ns0; vt = nvt; ha = nha; hb = nhb;
      if (o + o_stride < n_obs) request(o + o_stride);
      // Whole-obstacle test before any per-sample work.  Every sample's centre is within R = rmax[o] of the nominal one
      // and both its semi-axes are at most A = max(ha, hb) + |vt| whatever its heading, so d'Pd ≥ ((D - R)/A)² for an ego
      // circle at distance D from the nominal centre; q2·(1 - d'Pd) ≤ -64 follows from D ≥ R + A·sqrt(1 + 64/q2).  If that
      // holds for both circles at every step of the wavefront, the obstacle's samples are all below lin_step's own
      // threshold: skipped here for the price of this test instead of 46 instructions per sample.
      const double A = fmax(ha, hb) + fabs(vt), R = rmax[o];
      const double df = (fx - x) * (fx - x) + (fy - y) * (fy - y), dr = (rx - x) * (rx - x) + (ry - y) * (ry - y);
      const double tf = (R + A * kf) * (1.0 + 1.0e-9), tr = (R + A * kr) * (1.0 + 1.0e-9);
      const bool is_far = df >= tf * tf && dr >= tr * tr;  // false for NaN
      far = __builtin_amdgcn_ballot_w64(!is_far) == 0;
      cshape = rmax[n_obs + o] != 0.0;  // wave-uniform
    }
    if (far) {
      if (++s == S) { s = 0; o += o_stride; }
      return false;
    }
    const double2* q = reinterpret_cast<const double2*>(off + ((size_t)o * S + s) * OFFF);
    const double2 d = q[0], r = q[1];
    if (cshape) {  // heading, semi-axes of this sample do not depend on the step: formed once per solve in the prologue
      const double2 u = q[2];
      e.co = r.x; e.so = r.y; e.ia2 = u.x; e.ib2 = u.y;
    } else {
      e.co = c0 * r.x - s0 * r.y;
      e.so = s0 * r.x + c0 * r.y;
      const double ra = rcp_newton(ha + fabs(vt * e.co)), rb = rcp_newton(hb + fabs(vt * e.so));  // I/Obstacle.cpp:42-43
      e.ia2 = ra * ra;
      e.ib2 = rb * rb;
    }
    e.ox = x + d.x;
    e.oy = y + d.y;
    wout = w;
    if (++s == S) { s = 0; o += o_stride; }
    return true;
  }
};
struct SampledSource {
  static constexpr bool kPaired = false;
  const double* nom;
  const double* off;
  const double* rmax;
  const double* X;  // LDS trajectory (XR doubles per step)
  int N, S, n_obs;
  double w, ego_front, ego_rear, kf, kr;
  int o_first = 0, o_stride = 1;  // this source's share of the obstacles: o_first, o_first + o_stride, …
  __device__ __forceinline__ SampledObstacles at(int t) const {
    SampledObstacles a;
    a.nom = nom + (size_t)t * NOMF; a.off = off; a.N = N; a.S = S; a.n_obs = n_obs; a.w = w;
    a.o = o_first; a.o_stride = o_stride; a.s = 0; a.far = false; a.cshape = false; a.rmax = rmax;
    const double* xr = X + t * XR;
    a.fx = xr[0] + xr[4] * ego_front; a.fy = xr[1] + xr[5] * ego_front;
    a.rx = xr[0] - xr[4] * ego_rear; a.ry = xr[1] - xr[5] * ego_rear;
    a.kf = kf; a.kr = kr;
    if (o_first < n_obs) a.request(o_first);
    return a;
  }
};

// Prologue of the sampled-obstacle mode (TAB = 2), by NT threads of the workgroup (whole wavefronts): nominal records, the
// constant-shape flags, offset records, largest sample displacement per obstacle.
template <int NT>
__device__ __forceinline__ void sampled_prologue(const SolveArgs& a, const KParams& kp, int b, int N, int M, double* tab, double* off, double* rmax) {
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
  for (int m = wave; m < M; m += NT / WAVE) {  // nominal records: what I/Obstacle.cpp:41-62 needs of (obstacle, step) before the sample offset
    for (int t = lane; t < N; t += WAVE) {
      const double* pose = a.obs_pose + (((size_t)b * M + m) * N + t) * 4;
      const double* dim = a.obs_dim + (((size_t)b * M + m) * N + t) * 2;
      double sn, cs;
      sincos(pose[3], &sn, &cs);
      double* o = tab + ((size_t)m * N + t) * NOMF;
      o[0] = pose[0]; o[1] = pose[1]; o[2] = cs; o[3] = sn; o[4] = pose[2] * kp.t_safe;
      o[5] = dim[0] / 2.0 + kp.s_safe_a + kp.ego_rad;
      o[6] = dim[1] / 2.0 + kp.s_safe_b + kp.ego_rad + 1;
      o[7] = 0.0;
    }
  }
  // An obstacle that keeps its heading, speed and dimensions over the horizon (a vehicle driving straight: every obstacle of
  // the benchmark) gives each of its samples a heading and semi-axes that do not depend on the step: they are derived here,
  // once per solve (same arithmetic as the per-entry derivation in SampledObstacles: bit-identical results), instead of 22
  // instructions per entry, step and iteration.  Flag per obstacle behind rmax.
  for (int m = wave; m < M; m += NT / WAVE) {
    bool same = true;
    const double* pose0 = a.obs_pose + ((size_t)b * M + m) * N * 4;
    const double* dim0 = a.obs_dim + ((size_t)b * M + m) * N * 2;
    for (int t = lane; t < N; t += WAVE) {
      const double* pose = pose0 + (size_t)t * 4;
      const double* dim = dim0 + (size_t)t * 2;
      same = same && pose[2] == pose0[2] && pose[3] == pose0[3] && dim[0] == dim0[0] && dim[1] == dim0[1];
    }
    const bool all_same = __builtin_amdgcn_ballot_w64(!same) == 0;
    if (lane == 0) rmax[M + m] = all_same ? 1.0 : 0.0;
  }
  __syncthreads();
  const int n_off = M * a.n_samples;
  for (int i = tid; i < n_off; i += NT) {
    const double* q = a.samp_off + ((size_t)b * n_off + i) * 3;
    double sn, cs;
    sincos(q[2], &sn, &cs);
    double* o = off + (size_t)i * OFFF;
    o[0] = q[0]; o[1] = q[1];
    const int m = i / a.n_samples;
    if (rmax[M + m] != 0.0) {
      const double* pose0 = a.obs_pose + ((size_t)b * M + m) * N * 4;
      const double* dim0 = a.obs_dim + ((size_t)b * M + m) * N * 2;
      double s0, c0;
      sincos(pose0[3], &s0, &c0);  // the nominal record's own cos / sin
      const double vt = pose0[2] * kp.t_safe;
      const double ha = dim0[0] / 2.0 + kp.s_safe_a + kp.ego_rad, hb = dim0[1] / 2.0 + kp.s_safe_b + kp.ego_rad + 1;
      const double co = c0 * cs - s0 * sn, so = s0 * cs + c0 * sn;
      const double ra = rcp_newton(ha + fabs(vt * co)), rb = rcp_newton(hb + fabs(vt * so));  // I/Obstacle.cpp:42-43
      o[2] = co; o[3] = so; o[4] = ra * ra; o[5] = rb * rb;
    } else {
      o[2] = cs; o[3] = sn; o[4] = 0.0; o[5] = 0.0;
    }
  }
  for (int o = tid; o < M; o += NT) {  // largest sample displacement per obstacle (SampledObstacles' whole-obstacle test)
    double r2 = 0.0;
    for (int q = 0; q < a.n_samples; ++q) {
      const double* f = a.samp_off + ((size_t)b * n_off + (size_t)o * a.n_samples + q) * 3;
      const double d2 = f[0] * f[0] + f[1] * f[1];
      r2 = fmax(r2, d2 == d2 ? d2 : __builtin_huge_val());
    }
    rmax[o] = sqrt(r2);
  }
}

// Shader-clock stamp of the diagnostic instantiation, ordered by the compiler behind the value it names (the hardware issues in
// order: a stamp is taken when everything before it has ISSUED, and an instruction that needs an unfinished result stalls there).
__device__ __forceinline__ unsigned long long stamp_after(double v) {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(v) : "memory");
  return t;
}
// Stamps inside a step of the serial phases: requested without a wait (a wait per stamp costs the step ≈ 70 ticks each and the
// prefetched operand reads their cover), collected behind ONE s_waitcnt at the end of the step.
__device__ __forceinline__ void stamp_request(unsigned long long& t, double v) {
  asm volatile("s_memtime %0" : "=s"(t) : "v"(v) : "memory");
}
__device__ __forceinline__ void stamps_collect(unsigned long long& a, unsigned long long& b, unsigned long long& c, unsigned long long& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c), "+s"(d) : : "memory");
}
#define CILQR_SUB(slot, val)                          \
  if (DIAG) {                                         \
    const unsigned long long now_ = stamp_after(val); \
    sub[slot] += now_ - sub_t;                        \
    sub_t = now_;                                     \
  }

// Phase L.  Returns this lane's partial of J over its timesteps.  M = number of obstacle entries per step.
// RECW: doubles per stored record.  DIAG: sub[0..3] += ticks of {cos/sin columns, closest sample, cost derivatives, stores}.
template <int RECW, bool FSMEM, bool DIAG, typename Source>
__device__ __forceinline__ double linearize(const KParams& kp, int N, int M, int lane, const double* samp, int S,
                                            const SampleGrid& grid, double* X, const double* U, double* rec,
                                            const Source& src, const UncArgs* unc, UncPose upose, int ub, double* fwd,
                                            unsigned long long* sub) {
  double Jpart = 0.0;
  unsigned long long sub_t = 0;
  if (DIAG) sub_t = stamp_after(0.0);
  if (FSMEM) {
    // Scalar-path forward pass (forward_smem): it stores {x, y, v, theta} only — its cos/sin store was the last LDS instruction of a step
    // and the next step's wait paid its latency (34 ticks per step, 50 steps per pass) — so the cos/sin columns are filled here,
    // by lanes, from theta (35 instructions per call; the headings are within sincos_loop's range: rollout_fast, MAX_TURN).
    for (int t = lane; t <= N; t += WAVE) {
      double sn, cs;
      sincos_loop(X[t * XR + 3], sn, cs);
      X[t * XR + 4] = cs; X[t * XR + 5] = sn;
    }
    __syncthreads();
  }
  CILQR_SUB(0, Jpart)
  for (int t = lane; t < N; t += WAVE) {
    const double* xr = X + t * XR;
    const double* xn = X + (t + 1) * XR;
    const double px = xr[0], py = xr[1];
    if (FSMEM) {  // the forward pass reads the old state and control of step t through the scalar path
      double2* q = reinterpret_cast<double2*>(fwd + t * FREC + 10);
      q[0] = make_double2(px, py);
      q[1] = make_double2(xr[2], xr[3]);
      q[2] = make_double2(U[2 * t], U[2 * t + 1]);
    }
    const int cs = closest_sample<false>(S, grid, px, py, LdsSamples{samp, grid.xf, grid.dxs});
    CILQR_SUB(1, (double)cs)
    Rec c;
    Jpart += lin_step<true, Source::kPaired, FSMEM>(kp, px, py, xr[2], xr[4], xr[5], U[2 * t], U[2 * t + 1], xn[2], xn[4], xn[5], fma(grid.dxs, (double)cs, grid.xf),
                      samp[cs], M, src.at(t), c);
    CILQR_SUB(2, c.lx0 + c.lu0 + c.ga + Jpart)
    double* r = rec + t * RECW;
    r[0] = c.lx0; r[1] = c.lx1; r[2] = c.lx2; r[3] = c.l00; r[4] = c.l01; r[5] = c.l11;
    r[7] = c.lu1; r[9] = c.luu1;
    r[10] = c.al; r[11] = c.be; r[12] = c.ga; r[13] = c.de;
    if (RECW == REC) {
      r[6] = c.lu0; r[8] = c.luu0;
      r[14] = c.p; r[15] = c.q;
    } else {  // production records: acceleration in units of (dt/2)·u0 (riccati_mfma)
      const double ih = 2.0 / kp.dt;
      r[6] = c.lu0 * ih; r[8] = c.luu0 * (ih * ih);
    }
  }
  if (unc) {  // uniform: a map is set — its term joins l_x, l_xx after the obstacles' (I/Constraints.cpp:188-201)
    for (int t = lane; t < N; t += WAVE) {
      const double* xr = X + t * XR;
      double* r = rec + t * RECW;
      double lx0 = r[0], lx1 = r[1], h00 = r[3], h01 = r[4], h11 = r[5];
      unc_cost_add(*unc, upose, ub, xr[0], xr[1], xr[4], xr[5], lx0, lx1, h00, h01, h11);
      r[0] = lx0; r[1] = lx1; r[3] = h00; r[4] = h01; r[5] = h11;
    }
  }
  CILQR_SUB(3, Jpart)
  return Jpart;
}

// get_J only (used once after an accepted last iteration).
__device__ __forceinline__ double cost_only(const KParams& kp, int N, int lane, const double* samp, int S,
                                            const SampleGrid& grid, const double* X, const double* U) {
  double Jpart = 0.0;
  for (int t = lane; t < N; t += WAVE) {
    const double* xr = X + t * XR;
    const int cs = closest_sample<false>(S, grid, xr[0], xr[1], LdsSamples{samp, grid.xf, grid.dxs});
    Jpart += stage_cost(kp, xr[0] - fma(grid.dxs, (double)cs, grid.xf), xr[1] - samp[cs], xr[2] - kp.desired_speed, U[2 * t], U[2 * t + 1]);
  }
  return Jpart;
}

template <int RECW>
__device__ __forceinline__ void load_rec(Rec& o, const double* rec, int j) {
  const double* r = rec + j * RECW;
  o.lx0 = r[0]; o.lx1 = r[1]; o.lx2 = r[2]; o.l00 = r[3]; o.l01 = r[4]; o.l11 = r[5];
  o.lu0 = r[6]; o.lu1 = r[7]; o.luu0 = r[8]; o.luu1 = r[9];
  o.al = r[10]; o.be = r[11]; o.ga = r[12]; o.de = r[13];
  if (RECW == REC) { o.p = r[14]; o.q = r[15]; } else { o.p = 0.0; o.q = 0.0; }
}

// The gains of step j are stored over the first 10 doubles of step j's linearisation record — it has just been read for the
// last time (the recursion holds it in registers) — so `kK` is the record array itself and KS its record width: no gain array
// of its own in LDS (4 KB at N = 50).
template <int KS>
__device__ __forceinline__ void store_gains(double* kK, int j, const Gains& g) {
  if (threadIdx.x == 0) {  // every lane holds the same values; one lane stores
    double* o = kK + j * KS;
#pragma unroll
    for (int i = 0; i < KR; ++i) o[i] = g.g[i];
  }
}

// Phase R: iLQR::backward_pass recursion (I/iLQR.cpp:108-191).  All lanes compute the same values; operands are
// broadcast LDS reads issued one step ahead into the register set the next step uses (two steps per trip, no copies).
// GENERAL = false: branch-free fast pass; false ⇒ some step was suspect (hand the solve to the GENERAL kernel).
// GENERAL = true : branching pass; false ⇒ non-finite Q_uu (the reference's backward_pass returns false).
template <bool GENERAL>
__device__ __forceinline__ bool riccati(const KParams& kp, int N, const double* rec, double* kK, double lamb_in) {
  constexpr int RW = GENERAL ? REC : RECF;
  double dt = kp.dt, two_wvel = kp.w_vel * 2, lamb = lamb_in;
  CILQR_PIN(dt); CILQR_PIN(two_wvel); CILQR_PIN(lamb);
  Rec ra, rb;
  load_rec<RW>(ra, rec, N - 1);
  Value V;
  value_terminal(V, ra, two_wvel);
  unsigned long long suspect = 0;
  Gains g;
  bool ok;
  int j = N - 1;
  for (; j >= 1; j -= 2) {
    load_rec<RW>(rb, rec, j - 1);
    riccati_step<!GENERAL>(ra, V, dt, two_wvel, lamb, g, ok);
    if (GENERAL) { if (!ok) return false; } else suspect |= __builtin_amdgcn_ballot_w64(!ok);
    store_gains<RW>(kK, j, g);
    load_rec<RW>(ra, rec, j >= 2 ? j - 2 : 0);
    riccati_step<!GENERAL>(rb, V, dt, two_wvel, lamb, g, ok);
    if (GENERAL) { if (!ok) return false; } else suspect |= __builtin_amdgcn_ballot_w64(!ok);
    store_gains<RW>(kK, j - 1, g);
  }
  if (j == 0) {
    riccati_step<!GENERAL>(ra, V, dt, two_wvel, lamb, g, ok);
    if (GENERAL) { if (!ok) return false; } else suspect |= __builtin_amdgcn_ballot_w64(!ok);
    store_gains<RW>(kK, 0, g);
  }
  return suspect == 0;
}

// ---- Phase R on the matrix cores (production kernel) --------------------------------------------------------------------------
// The recursion is a chain of 4×4 fp64 matrix products, which `riccati` above evaluates entry by entry, every lane computing
// every entry (≈ 131 fp64 instructions per step on the serial chain).  v_mfma_f64_4x4x4_4b_f64 forms four 4×4×4 products in one
// instruction.  Measured on gfx950 (tools/ubench_mfma_f64.hip): a lone wavefront issues one every 16.2 ticks (20 / 28 when the
// result feeds the next one's C / A or B operand), each entry a chain of fused multiply-adds, and vector instructions do NOT
// issue beside it — a matrix instruction costs what three fp64 vector instructions cost, so the count of both is what matters.
// Its lane map, likewise measured: with a matrix X kept as "entry (r, c) in lane c + 4·blk + 16·r" (blk = 0..3: the
// instruction's four independent blocks), the instruction computes, per block,   D = Xaᵀ · Xb + Xc   with all four of Xa, Xb, Xc,
// D in that one layout — a result is the next product's right or (transposed) left operand with no lane movement.
//
// Two blocks are used (blocks 2, 3 repeat 0, 1); "[X | Y]" = X in block 0, Y in block 1; B~ = [B 0 0] padded to 4×4:
//   P  = [V | V]' [A | B~] + [0 | vc]                         = [V'A | V'B~ + vc]          vc = column 2 holds V_x, else 0
//   Da = [A | A]' P + [l_xx | column 2: l_x]                   = [Q_xx | (Q_xu, Q_x, 0)]                                (:149-151)
//   Db = [B~ | B~]' P + [0 | (l_uu, l_u, 0)]                   = [Q_ux in rows 0-1 | (Q_uu, Q_u, 0) in rows 0-1]        (:150-153)
//   a, b, d of Q_uu by v_readlane; PSD test, determinant, reciprocal on all lanes as in riccati_step           (:155-175)
//   Dk = -(1/det)·[adj | adj]' Db                              = [K in rows 0-1 | (*, k, 0)]      adj = adjugate of Q_uu + lamb·I, laid out by lane masks (:177-178)
//   H  = Db + lamb·Dk, block 0 copied to block 1               = [Q_ux + lamb·K | same]   (= -G' by the identity of riccati_step<FAST>)
//   Dv = H' Dk + Da                                            = [V_xx | (*, V_x, 0)]                                   (:180-181)
//   next V = block 0 of Dv in both blocks; next vc = Dv masked to column 2 of block 1.
// Five matrix instructions, four DPP moves and ≈ 40 others per step instead of ≈ 162 vector instructions.  V enters the first
// product transposed: V_xx is symmetric, its two triangles agree to rounding (the reference computes both as well).  The
// per-step operands come from the 14-double records, each lane fetching the entry of its (block, r, c) — or a constant
// from a small table — through a per-lane LDS address: five ds_read_b64 per step.  Gains leave through the lanes that hold them,
// into the layout phase F reads.
// Units: the first column of B is (p, q, dt, 0) = (dt/2)·(al, be, 2, 0) (I/Model.cpp:139-155 against :100-127), so the pass
// carries the acceleration in units of (dt/2)·u0: B~'s entries are read straight from al, be and two constants, phase L stores
// l_u(0)·(2/dt) and l_uu(0)·(2/dt)² in these records (linearize), lamb·I becomes diag(lamb·(2/dt)², lamb), H'Dk is invariant,
// and phase F multiplies the acceleration gains by 2/dt in the instruction that adds the old control (forward_step<true>).
// Sums are formed in another order than in riccati_step (four fused
// multiply-adds over k per entry, structural zeros included), so results agree with it to rounding, like the other identities
// of the production kernel; a non-finite or non-PSD Q_uu hands the solve to the GENERAL kernel exactly as before.
#define CILQR_MFMA(xa, xb, xc) __builtin_amdgcn_mfma_f64_4x4x4f64(xa, xb, xc, 0, 0, 0)
// Constant table behind the records: {0, 1, dt, 2·w_vel, 2} twice, RECF doubles apart — a lane that reads a constant keeps its
// address while the others step through the records, and the two steps of one loop trip are read at immediate offsets 0 and RECF.
// (Records of 16 doubles would hold p and q, but a lane stride of 128 bytes puts phase L's record stores on two banks only:
// measured +17 % on phase L; 112 bytes spread a quarter-wave's 16-byte stores over all 64 banks.)
constexpr int RCST = REC + 6;

__device__ __forceinline__ double readlane_f64(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}
// block 1 (and 3) := block 0 (and 2), rows kept: DPP row_shr:4 under bank mask 0b1010
__device__ __forceinline__ double odd_blocks_from_even(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x114, 0xF, 0xA, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x114, 0xF, 0xA, false);
  return __hiloint2double(hi, lo);
}

struct MfmaOperands { double AB, AA, BB, Ca, Cb; };
struct MfmaCursor {  // this lane's read position in each operand (at the LOWER step of a pair) and its stride per pair (0: constant)
  const double *AB, *AA, *BB, *Ca, *Cb;
  int sAB, sAA, sBB, sCa, sCb;
  __device__ __forceinline__ void upper(MfmaOperands& o) const { o.AB = AB[RECF]; o.AA = AA[RECF]; o.BB = BB[RECF]; o.Ca = Ca[RECF]; o.Cb = Cb[RECF]; }
  __device__ __forceinline__ void lower(MfmaOperands& o) const { o.AB = AB[0]; o.AA = AA[0]; o.BB = BB[0]; o.Ca = Ca[0]; o.Cb = Cb[0]; }
  __device__ __forceinline__ void back() { AB -= sAB; AA -= sAA; BB -= sBB; Ca -= sCa; Cb -= sCb; }
};

// slot: record slot (≥ 0) or constant -1 - k of this lane's entry; `low`: the record below the topmost one
__device__ __forceinline__ void mfma_place(const double*& ptr, int& stride, int slot, const double* low, const double* cst) {
  ptr = slot >= 0 ? low + slot : cst + (-1 - slot);
  stride = slot >= 0 ? 2 * RECF : 0;
}

// DIAG: sub[5..7] += ticks of {the three products up to a, b, d in scalar registers; determinant and reciprocal; the rest of the step}.
template <bool FSMEM, bool DIAG>
__device__ __forceinline__ bool riccati_mfma(int N, const double* rec, double* kK, double* fwd, const double* cst, double inv_half_dt, double lamb_in,
                                             unsigned long long* sub) {
  constexpr int C0 = -1, C1 = -2, CDT = -3, CW = -4, C2 = -5;
  const int lane = threadIdx.x;
  const int e = ((lane >> 4) << 2) | (lane & 3);  // 4·r + c of this lane's entry
  const bool odd = (lane & 4) != 0;               // block 1 (or 3)
  // record slots: 0-2 l_x, 3-5 l_xx (00, 01, 11), 6-7 l_u, 8-9 l_uu, 10-13 al, be, ga, de (linearize)
  const int slotA = e == 0 || e == 5 || e == 10 || e == 15 ? C1 : e == 2 ? 10 : e == 6 ? 11 : e == 3 ? 12 : e == 7 ? 13 : C0;
  const int slotB = e == 0 ? 10 : e == 4 ? 11 : e == 8 ? C2 : e == 13 ? CDT : C0;  // first column (p, q, dt, 0) / (dt/2) = (al, be, 2, 0)
  const int slotXX = e == 0 ? 3 : e == 1 || e == 4 ? 4 : e == 5 ? 5 : e == 10 ? CW : C0;
  const int slotX = e == 2 ? 0 : e == 6 ? 1 : e == 10 ? 2 : C0;
  const int slotUU = e == 0 ? 8 : e == 5 ? 9 : e == 2 ? 6 : e == 6 ? 7 : C0;
  const double* low = rec + (N - 2) * RECF;  // (N = 1: the record "below" is never read)
  MfmaCursor cur;
  mfma_place(cur.AB, cur.sAB, odd ? slotB : slotA, low, cst);
  mfma_place(cur.AA, cur.sAA, slotA, low, cst);
  mfma_place(cur.BB, cur.sBB, slotB, low, cst);
  mfma_place(cur.Ca, cur.sCa, odd ? slotX : slotXX, low, cst);
  mfma_place(cur.Cb, cur.sCb, odd ? slotUU : C0, low, cst);
  double m00 = e == 0 ? 1.0 : 0.0, m01 = e == 1 || e == 4 ? 1.0 : 0.0, m11 = e == 5 ? 1.0 : 0.0;
  double mvc = odd && (lane & 3) == 2 ? 1.0 : 0.0;
  // lamb·I in the scaled units: diag(lamb·(2/dt)², lamb); per lane the factor of its row of K
  double lamb = lamb_in, lamb0 = lamb_in * (inv_half_dt * inv_half_dt);
  double lamb_row = (lane >> 4) == 0 ? lamb0 : lamb;
  CILQR_PIN(m00); CILQR_PIN(m01); CILQR_PIN(m11); CILQR_PIN(mvc); CILQR_PIN(lamb); CILQR_PIN(lamb0); CILQR_PIN(lamb_row);
  // gains in Dk: K(0, c) in lanes 0-3, K(1, c) in lanes 16-19 (block 0); k(0), k(1) in lanes 6, 22 (column 2 of block 1)
  // FSMEM: they go to the forward-pass records in global memory, the lanes that hold no gain write into the dump record behind
  // the last.  Otherwise: over the first 10 doubles of the step's record in LDS, the other lanes into its slots 10-13, which
  // nobody reads any more.  Every lane stores either way: no EXEC change.
  const bool stores = lane == 6 || lane == 22 || lane < 4 || (lane >= 16 && lane < 20);
  const int gslot = lane == 6 ? 0 : lane == 22 ? 1 : lane < 4 ? 2 + lane : 6 + (lane & 3);
  unsigned goff = stores ? (unsigned)(((N - 1) * FREC + gslot) * 8) : (unsigned)((N * FREC + (lane & 15)) * 8);
  const unsigned gstride = stores ? FREC * 8 : 0;
  char* const gbase = reinterpret_cast<char*>(fwd);
  double* gp = kK + (N - 1) * RECF + (stores ? gslot : 10 + (lane & 3));

  MfmaOperands oa, ob;
  cur.upper(oa);
  // :108-113: terminal value = stage N-1: V = l_xx in both blocks, vc = l_x in column 2 of block 1
  double V = odd_blocks_from_even(oa.Ca), vc = oa.Ca * mvc;
  // PSD test of riccati_step<FAST> (det0 ≥ 0 and a + d ≥ 0, i.e. det0, a and d all ≥ 0), kept as sign bits: the high words of
  // det0, a and d are OR-ed into one word over the pass, and the last det0 is looked at for NaN at the end (a NaN anywhere
  // in the recursion stays in V_xx down to step 0).  -0.0 counts as negative: such a solve is merely handed over.
  int signs_v = 0, signs_s = 0;  // (one word in a vector register for det0, one in a scalar register for a and d: no transfers)
  double det0 = 0.0;
  auto step = [&](const MfmaOperands& o) {
    unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (DIAG) stamp_request(s0, V);
    const double P = CILQR_MFMA(V, o.AB, vc);
    const double Db = CILQR_MFMA(o.BB, P, o.Cb);
    const double Da = CILQR_MFMA(o.AA, P, o.Ca);
    const double a = readlane_f64(Db, 4), b = readlane_f64(Db, 5), d = readlane_f64(Db, 21);
    if (DIAG) stamp_request(s1, a + b + d);
    const double bb = b * b;
    det0 = fma(a, d, -bb);
    signs_v |= __double2hiint(det0);
    signs_s |= __double2hiint(a) | __double2hiint(d);
    const double ar = a + lamb0, dr = d + lamb;
    const double nr = -rcp_newton(fma(ar, dr, -bb));
    if (DIAG) stamp_request(s2, nr);
    const double adj = fma(ar, m11, fma(-b, m01, dr * m00));
    const double Dk = CILQR_MFMA(adj, Db, 0.0) * nr;
    const double H0 = fma(lamb_row, Dk, Db);
    const double H = odd_blocks_from_even(H0);
    double Dv = CILQR_MFMA(H, Dk, Da);
    vc = Dv * mvc;
    CILQR_PIN(vc);  // (the product before the copy below, so that the copy can be made in place)
    CILQR_PIN(Dv);
    V = odd_blocks_from_even(Dv);
    if (FSMEM) {
      *reinterpret_cast<double*>(gbase + goff) = Dk;
      goff -= gstride;
    } else {
      *gp = Dk;
      gp -= RECF;
    }
    if (DIAG) {
      stamp_request(s3, V);
      stamps_collect(s0, s1, s2, s3);
      sub[5] += s1 - s0; sub[6] += s2 - s1; sub[7] += s3 - s2;
    }
  };
  // two steps per trip: the lower step's operands are read while the upper one computes, the next trip's upper operands while
  // the lower one computes; the last one or two steps are peeled so that no read reaches below the first record
  int j = N - 1;
  for (; j >= 2; j -= 2) {
    cur.lower(ob);
    step(oa);
    cur.back();
    cur.upper(oa);
    step(ob);
  }
  if (j == 1) {
    cur.lower(ob);
    step(oa);
    step(ob);
  } else {
    step(oa);
  }
  return __builtin_amdgcn_ballot_w64((signs_v | signs_s) < 0 || !(det0 == det0)) == 0;
}

template <int KS>
__device__ __forceinline__ void load_fwd(FwdIn& o, const double* X, const double* U, const double* kK, int i) {
  const double* xo = X + i * XR;
  o.x = xo[0]; o.y = xo[1]; o.v = xo[2]; o.th = xo[3];
  o.u0 = U[2 * i]; o.u1 = U[2 * i + 1];
  const double* g = kK + i * KS;
#pragma unroll
  for (int k = 0; k < KR; ++k) o.g[k] = g[k];
}

__device__ __forceinline__ void fwd_step_store(const FwdConst& k, const FwdIn& c, State& s, double& max_turn, double inv_half_dt,
                                               double* Un_i, double* Xn_next) {
  double u0, u1;
  forward_step<true>(k, c, s, max_turn, u0, u1, inv_half_dt);
  // every lane holds the same values and stores them to the same addresses: no EXEC juggling around the stores
  Un_i[0] = u0; Un_i[1] = u1;
  Xn_next[0] = s.x; Xn_next[1] = s.y; Xn_next[2] = s.v; Xn_next[3] = s.th; Xn_next[4] = s.c; Xn_next[5] = s.s;
}

// ---- Phase F with its operands through the scalar path (production kernel) --------------------------------------------------------
// The forward pass needs 16 doubles per step (old state and control, gains), the same for all lanes.  As broadcast LDS reads
// that was 8 ds_read2_b64 per step beside 4 stores — and the LDS pipeline of a CU serves its four SIMDs: measured with
// tools/phase_cycles.py at 1 / 2 / 4 wavefronts per CU (B = 256 / 512 / 1024) phase F took 437 / 461 / 529 ticks per step (365
// with the reads removed), phase R with its 6 LDS instructions 378 / 385 / 387.  Neither narrowing EXEC around the LDS
// instructions (by branches: 583; from one inline-assembly block per step: 536) nor 128-bit reads (553) changed that: what
// counts is the number of LDS instructions in flight on the CU.  So the operands take the scalar path instead: phases L and R
// write them to a 128-byte record per step in global memory (FREC), phase F reads a step with two s_load_dwordx16 into scalar
// registers, one step ahead, and uses them as the scalar operand of its vector instructions (one per instruction: every
// product of the pass has one loaded and one computed factor).  Hand-over: s_waitcnt vmcnt(0) — the vector stores have reached
// L2 — and s_dcache_inv before the first scalar load (checked on its own by tools/ubench_smem.hip).  Every record is a first touch
// for the scalar cache; the ≈ 250 ticks between a request and its wait cover that (one-dword loads two to five steps ahead, to
// warm the cache, made the step slower: 430 ticks against 400).  hipcc does not see these loads: every use lies behind f_swait,
// which carries the registers as in/out operands of the s_waitcnt that makes them valid.
typedef double d8_t __attribute__((ext_vector_type(8)));
struct FwdS { d8_t lo, hi; };  // lo = {k0, k1, K00..K03, K10, K11}, hi = {K12, K13, x, y, v, theta, u0, u1}
__device__ __forceinline__ void f_sload_first(FwdS& r, const double* p) {
  asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(r.lo), "=&s"(r.hi) : "s"(p) : "memory");
}
// The next record into the SAME registers: as in/out operands, so that every use of the old values lies before this point and
// one set of 32 scalar registers serves the whole pass (two sets alive at once do not fit beside the kernel's other scalars).
__device__ __forceinline__ void f_sload_next(FwdS& r, const double* p) {
  asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "+s"(r.lo), "+s"(r.hi) : "s"(p) : "memory");
}
// The wait is the compiler's own s_waitcnt (the builtin), not text inside the assembly statement: hipcc's wait-count pass then
// KNOWS that no LDS store of an earlier step is pending behind this point.  With the wait hidden in assembly text it protected the
// data registers of those stores itself where it saw fit — in the two-wavefront kernel with an s_waitcnt lgkmcnt(1) in the middle
// of the step, which (the scalar loads, invisible to it, being in flight there) stalled every step on the record it had just
// requested: F 372 → 504 ticks per step.  The empty assembly statement behind it carries the registers, as before.
__device__ __forceinline__ void f_swait(FwdS& r) {
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
  asm volatile("" : "+s"(r.lo), "+s"(r.hi) : : "memory");
}
__device__ __forceinline__ void f_sunpack(FwdIn& o, const FwdS& r) {
  o.g[0] = r.lo[0]; o.g[1] = r.lo[1]; o.g[2] = r.lo[2]; o.g[3] = r.lo[3]; o.g[4] = r.lo[4]; o.g[5] = r.lo[5];
  o.g[6] = r.lo[6]; o.g[7] = r.lo[7]; o.g[8] = r.hi[0]; o.g[9] = r.hi[1];
  o.x = r.hi[2]; o.y = r.hi[3]; o.v = r.hi[4]; o.th = r.hi[5]; o.u0 = r.hi[6]; o.u1 = r.hi[7];
}

// Phase F: iLQR::forward_pass (I/iLQR.cpp:68-86), production kernel.  A step: wait for its record; the controls from the
// record and the running state (forward_controls); request the next record; the dynamics (dyn_step_loop), ≈ 250 ticks in which
// the request completes; results to LDS (every lane stores the same values to the same addresses), where the next phase L reads
// them by lanes.  Returns false if a step turned the heading by more than MAX_TURN (rotate_heading): the results are then not to
// be used and the solve is handed to the GENERAL kernel.
// PUBLISH (two-wavefront kernel): the second wavefront of the workgroup watches theta of the states for their arrival
// (state_arrived below), so a state's theta is stored behind everything else of its step.
template <bool PUBLISH = false>
__device__ __forceinline__ bool forward_smem(const KParams& kp, int N, const double* X, const double* fwd, double* Xn, double* Un) {
  asm volatile("s_waitcnt vmcnt(0)\n\ts_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");  // L's and R's record stores → scalar loads
  FwdConst k;
  make_fwd_const(k, kp);
  double ihd = 2.0 / kp.dt;  // the acceleration gains are in units of (dt/2)·u0 (forward_controls<true>)
  CILQR_PIN(ihd);
  State s;
  s.x = X[0]; s.y = X[1]; s.v = X[2]; s.th = X[3]; s.c = X[4]; s.s = X[5];
  store_state(Xn, 0, s);
  double max_turn = 0.0;  // (the first heading was checked by the rollout)
  auto recp = [&](int j) { return fwd + (size_t)(j < N ? j : N) * FREC; };  // (past the end: the dump record)
  FwdS r;
  FwdIn c;
  f_sload_first(r, recp(0));
  auto step = [&](int i) {
    f_swait(r);
    f_sunpack(c, r);
    double u0, u1;
    forward_controls_scalar(c, s, u0, u1, ihd);
    f_sload_next(r, recp(i + 1));
    Un[2 * i] = u0; Un[2 * i + 1] = u1;
    const double delta = dyn_pose_loop(k, s, u0, u1, max_turn);
    double* xo = Xn + (i + 1) * XR;  // (cos/sin columns: filled by the next phase L)
    xo[0] = s.x; xo[1] = s.y;
    if (PUBLISH) asm volatile("" ::: "memory");  // (compiler order only: the LDS executes one wavefront's stores in order)
    xo[2] = s.v; xo[3] = s.th;
    // the rotation last: its ≈ 100 ticks cover the latency of the stores above, which the next step's s_waitcnt lgkmcnt(0) —
    // the only safe wait with scalar loads in flight — would otherwise pay
    rotate_heading(k, delta, s.s, s.c);
    CILQR_PIN2(s.s, s.c);
  };
  // two steps per trip: the rotated (cos, sin) of one step are the inputs of the next in other registers, no copies at the back edge
  int i = 0;
  for (; i + 1 < N; i += 2) {
    step(i);
    step(i + 1);
  }
  if (i < N) step(i);
  f_swait(r);  // (the request of the last step: the dump record)
  return max_turn <= MAX_TURN;
}

// Phase F: iLQR::forward_pass (I/iLQR.cpp:68-86), operands from LDS (the production instantiations that stream their obstacles
// from global memory; forward_smem otherwise).  All lanes compute and store the same values; the operands of
// step i+1 (old state, old control, gains) are read while step i computes (two steps per trip, no register copies).
// Returns false if a step turned the heading by more than MAX_TURN (rotate_heading, cilqr_device.hpp): the results are then
// not to be used and the solve is handed to the GENERAL kernel.
template <int KS>
__device__ __forceinline__ bool forward_fast(const KParams& kp, int N, const double* X, const double* U, const double* kK,
                                             double* Xn, double* Un) {
  FwdConst k;
  make_fwd_const(k, kp);
  double ihd = 2.0 / kp.dt;  // the acceleration gains are in units of (dt/2)·u0 (forward_step<true>)
  CILQR_PIN(ihd);
  State s;
  s.x = X[0]; s.y = X[1]; s.v = X[2]; s.th = X[3]; s.c = X[4]; s.s = X[5];
  if (threadIdx.x == 0) store_state(Xn, 0, s);
  double max_turn = 0.0;  // (the first heading was checked by the rollout)
  FwdIn fa, fb;
  load_fwd<KS>(fa, X, U, kK, 0);
  int i = 0;
  for (; i + 1 < N; i += 2) {
    load_fwd<KS>(fb, X, U, kK, i + 1);
    fwd_step_store(k, fa, s, max_turn, ihd, Un + 2 * i, Xn + (i + 1) * XR);
    load_fwd<KS>(fa, X, U, kK, i + 2 < N ? i + 2 : i + 1);
    fwd_step_store(k, fb, s, max_turn, ihd, Un + 2 * (i + 1), Xn + (i + 2) * XR);
  }
  if (i < N) fwd_step_store(k, fa, s, max_turn, ihd, Un + 2 * i, Xn + (i + 1) * XR);
  return max_turn <= MAX_TURN;
}

// The same pass with the range-guarded sincos (library path for huge arguments); GENERAL kernel only.
__device__ __forceinline__ void forward_general(const KParams& kp, int N, const double* X, const double* U, const double* kK,
                                                double* Xn, double* Un) {
  State s;
  s.x = X[0]; s.y = X[1]; s.v = X[2]; s.th = X[3]; s.c = X[4]; s.s = X[5];
  if (threadIdx.x == 0) store_state(Xn, 0, s);
  for (int i = 0; i < N; ++i) {
    const double* xo = X + i * XR;
    const double* g = kK + i * REC;  // the GENERAL kernel's records are 16 doubles wide
    const double d0 = s.x - xo[0], d1 = s.y - xo[1], d2 = s.v - xo[2], d3 = s.th - xo[3];
    const double u0 = fma(g[5], d3, fma(g[4], d2, fma(g[3], d1, fma(g[2], d0, U[2 * i] + g[0]))));
    const double u1 = fma(g[9], d3, fma(g[8], d2, fma(g[7], d1, fma(g[6], d0, U[2 * i + 1] + g[1]))));
    s = dyn_step(kp, s, u0, u1);
    if (threadIdx.x == 0) {
      Un[2 * i] = u0;
      Un[2 * i + 1] = u1;
      store_state(Xn, i + 1, s);
    }
  }
}

// Nominal rollout (I/iLQR.cpp:51-62) on the in-loop sincos; false ⇒ hand over to the GENERAL kernel.
template <bool PUBLISH = false>
__device__ __forceinline__ bool rollout_fast(const KParams& kp, int N, const double* x0, const double* U, double* X) {
  FwdConst k;
  make_fwd_const(k, kp);
  State s;
  s.x = x0[0]; s.y = x0[1]; s.v = x0[2]; s.th = x0[3];
  const bool th0_ok = fabs(s.th) < MAX_HEADING0;
  double max_turn = 0.0;
  sincos_loop(s.th, s.s, s.c);
  store_state(X, 0, s);  // (every lane holds the same values and stores them to the same addresses: no EXEC change)
  for (int i = 0; i < N; ++i) {
    dyn_step_loop(k, s, U[2 * i], U[2 * i + 1], max_turn);
    if (PUBLISH) {  // theta last (forward_smem)
      double* r = X + (i + 1) * XR;
      r[0] = s.x; r[1] = s.y; r[4] = s.c; r[5] = s.s;
      asm volatile("" ::: "memory");
      r[2] = s.v; r[3] = s.th;
    } else {
      store_state(X, i + 1, s);
    }
  }
  return th0_ok && max_turn <= MAX_TURN;
}

// DIAG: per-solve shader-clock totals by phase, written to a.diag[b][8] = {prologue, L, R, F, epilogue, L count, R count,
// total}.  A separate instantiation so that the production kernel carries no stamps.
// TAB: where a step's obstacle entries come from — 0: the table in the global workspace; 1: the table in LDS (chosen by the
// launcher when it fits beside the rest at the wanted residency); 2: sampled obstacles, derived on the fly (SampledObstacles).
// UNC: an uncertainty map is set (cilqr_set_uncertainty_map*).  A separate instantiation because the mere presence of the map
// term's code — even behind a branch never taken — cost config 2 1.2 % through register allocation (A/B on one box: 0.7463 vs
// 0.7556 ms): without a map the kernel that runs contains none of it.
// GENERAL: false = the production kernel: branch-free fast passes.  A solve that meets anything the fast passes do not
// cover (Q_uu not positive semi-definite or not finite; a heading beyond the in-loop sincos range) stops WITHOUT touching
// its outputs and sets a.redo[b]; the GENERAL = true kernel, launched right behind on the same stream, redoes exactly
// those solves from their untouched inputs with the branching passes and returns at once for all others.
// (second launch bound: without a map the kernel must leave room for TWO wavefronts per SIMD — ≤ 256 vector registers; batches
// beyond one solve per SIMD depend on it, and the sampled instantiation had crept to 257 + 1 after a refactoring, unnoticed)
template <bool DIAG, int TAB, bool GENERAL, bool UNC>
__global__ __launch_bounds__(WAVE, UNC ? 1 : 2) void cilqr_solve_kernel(SolveArgs a) {
  unsigned long long tk0 = 0, tk = 0, c_pro = 0, c_L = 0, c_R = 0, c_F = 0, n_L = 0, n_R = 0;
  unsigned long long sub[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // DIAG: ticks inside phases L (0-4) and R (5-7)
  if (DIAG) tk0 = tk = __builtin_readcyclecounter();
#define CILQR_STAMP(acc)                                 \
  if (DIAG) {                                            \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    acc += now_ - tk;                                    \
    tk = now_;                                           \
  }
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if ((int)blockIdx.x >= a.B) return;
  // dispatch order: cilqr_api.cpp, schedule hint (wave-uniform; said so to the compiler, which loads it through a vector register)
  const int b = __builtin_amdgcn_readfirstlane(a.order ? a.order[blockIdx.x] : (int)blockIdx.x);
  const int lane = threadIdx.x;
  const KParams kp = a.kp;
  const int N = a.N, M = a.M, S = kp.n_samples;
  const int n_entries = TAB == 2 ? M * a.n_samples : M;  // obstacle entries per step
  const bool has_unc = UNC;
  if (b >= a.B) return;
  if (GENERAL && a.redo[b] == 0) return;

  // The production kernel runs the forward pass IN PLACE unless CILQR_FLAG_FAITHFUL_ITERS is set: with the early exit a forward
  // pass is only ever run for an accepted iteration, so its result always replaces the trajectory it started from, and the pass
  // reads a step's old state, control and gains one step before it overwrites them (forward_fast).  No candidate buffers: 3.2 KB
  // less LDS per solve at N = 50.
  constexpr int RECW = GENERAL ? REC : RECF;
  // Forward-pass operands through the scalar path (forward_smem) where the obstacle table lies in LDS; where a solve streams its
  // obstacles from global memory (TAB 0, 2) the records would compete with that stream for L2 and the scalar loads miss: config 3
  // measured 0.95 M solves/s with them against 1.05 M without.
  constexpr bool FSMEM = !GENERAL && TAB == 1;
  const bool twin = GENERAL || (a.flags & CILQR_FLAG_FAITHFUL_ITERS) != 0;
  double* samp = lds;
  double* Xa = samp + ((S + 1) & ~1);  // (even count: the records behind stay 16-byte aligned)
  double* Xb = twin ? Xa + (N + 1) * XR : Xa;
  double* Ua = Xb + (N + 1) * XR;
  double* Ub = twin ? Ua + 2 * N : Ua;
  double* rec = Ub + 2 * N;
  double* kK = rec;  // the gains overlay the records (store_gains)
  double* cst = rec + N * RECW;  // {0, 1, dt, 2·w_vel}: riccati_mfma's constant entries
  double* fwd = a.fwd + (size_t)b * (N + 1) * FREC;  // forward-pass records of this solve (production kernel)
  double* tab = TAB == 1 ? cst + RCST : a.obs_tab + (size_t)b * M * (TAB == 2 ? NOMF : TABF) * N;
  double* off = cst + RCST;  // TAB == 2: offset records [o][s][OFFF], then rmax[o]
  double* rmax = off + (size_t)M * a.n_samples * OFFF;

  // ---- prologue -------------------------------------------------------------------------------------------
  SampleGrid grid;
  make_sample_grid(grid, a.xplan_fl[2 * b], a.xplan_fl[2 * b + 1], S);
  {
    const double* pc = a.poly + (size_t)b * CILQR_POLY_COEFFS;
    for (int s = lane; s < S; s += WAVE) {
      double xs;
      sample_xy(grid, pc, s, xs, samp[s]);
    }
  }
  double* Ug = a.U + (size_t)b * 2 * N;
  for (int i = lane; i < 2 * N; i += WAVE) Ua[i] = Ug[i];
  if (lane < 16 && (lane & 7) < 5) cst[(lane & 7) + (lane >> 3) * RECF] = (lane & 7) == 0 ? 0.0 : (lane & 7) == 1 ? 1.0 : (lane & 7) == 2 ? kp.dt : (lane & 7) == 3 ? kp.w_vel * 2 : 2.0;

  const double* wts = (TAB != 2 && a.obs_weight) ? a.obs_weight + (size_t)b * M : nullptr;
  if (TAB == 2) {
    sampled_prologue<WAVE>(a, kp, b, N, M, tab, off, rmax);
  } else {
    for (int m = 0; m < M; ++m) {  // obstacle table, I/Obstacle.cpp:41-62
      for (int t = lane; t < N; t += WAVE) {
        const ObsEntry e = make_obs_entry(kp, a.obs_pose + (((size_t)b * M + m) * N + t) * 4, a.obs_dim + (((size_t)b * M + m) * N + t) * 2);
        double* o = tab + ((size_t)m * N + t) * TABF;
        o[0] = e.ox; o[1] = e.oy; o[2] = e.co; o[3] = e.so; o[4] = e.ia2; o[5] = e.ib2;
      }
    }
  }
  __syncthreads();
  {  // largest step between adjacent path samples in y: closest_sample's second window (cilqr_device.hpp)
    double m = 0.0;
    for (int q = lane; q + 1 < S; q += WAVE) {
      const double d = fabs(samp[q + 1] - samp[q]);
      m = fmax(m, d == d ? d : __builtin_huge_val());
    }
    grid.dmax = wave_max_uniform(m);
  }

  UncPose upose{0, 0, 1, 0};
  if (has_unc) upose = unc_pose(a.unc, b);
  bool handover = false;  // fast kernel only: this solve needs the GENERAL kernel
  if (GENERAL) {          // nominal rollout, I/iLQR.cpp:51-62
    const double* x0 = a.x0 + (size_t)b * 4;
    State s;
    s.x = x0[0]; s.y = x0[1]; s.v = x0[2]; s.th = x0[3];
    sincos_fast(s.th, &s.s, &s.c);
    if (lane == 0) store_state(Xa, 0, s);
    for (int i = 0; i < N; ++i) {
      s = dyn_step(kp, s, Ua[2 * i], Ua[2 * i + 1]);
      if (lane == 0) store_state(Xa, i + 1, s);
    }
  } else {
    handover = !rollout_fast(kp, N, a.x0 + (size_t)b * 4, Ua, Xa) || (a.flags & CILQR_FLAG_GENERAL_ONLY) != 0;
  }
  __syncthreads();

  CILQR_STAMP(c_pro)
  // ---- iteration loop, I/iLQR.cpp:204-239 --------------------------------------------------------------------
  double* Xc = Xa;
  double* Uc = Ua;
  double* Xn = Xb;
  double* Un = Ub;
  double J_old = DBL_MAX, lamb = 1.0, J_new = 0.0;
  int iters = 0, status = CILQR_EXIT_MAX_ITER, n_pass = 0;
  bool j_valid = false;  // J_new is get_J of the current (Xc, Uc)
  const bool faithful = (a.flags & CILQR_FLAG_FAITHFUL_ITERS) != 0;
  const int max_it = kp.max_iterations;
  for (int it = 0; it < max_it && !handover; ++it) {
    ++iters;
    // The reference evaluates backward_pass, forward_pass, then J_new = get_J(X, U) on the CURRENT X, U (:213-217).
    // The linearisation and J share their closest-point searches, so they are computed together, first.
    {
      double part;
      const KParams kpl = phase_params();  // phase-local read of the parameter block (cilqr_device.hpp)
      const UncArgs* unc = has_unc ? &phase_args().unc : nullptr;  // uniform: a map is set (cilqr_set_uncertainty_map*)
      if (TAB == 2) part = linearize<RECW, FSMEM, DIAG>(kpl, N, n_entries, lane, samp, S, grid, Xc, Uc, rec, SampledSource{tab, off, rmax, Xc, N, a.n_samples, M, a.samp_w, kpl.ego_front, kpl.ego_rear,
                                                     sqrt(1.0 + 64.0 / kpl.q2_front), sqrt(1.0 + 64.0 / kpl.q2_rear)}, unc, upose, b, fwd, sub);
      else part = linearize<RECW, FSMEM, DIAG>(kpl, N, n_entries, lane, samp, S, grid, Xc, Uc, rec, TabSource<TAB == 0>{tab, wts, N, kpl.w_obstacle}, unc, upose, b, fwd, sub);
      unsigned long long sub_t = 0;
      if (DIAG) sub_t = stamp_after(part);
      J_new = wave_sum_uniform(part);
      CILQR_SUB(4, J_new)
    }
    j_valid = true;
    __syncthreads();
    CILQR_STAMP(c_L)
    if (DIAG) ++n_L;
    const bool accept = J_new < J_old;
    if (!accept && !faithful) {
      // The reference runs backward_pass BEFORE this test (:213-215) and stops there if it fails.  With a NaN in the
      // trajectory (the only way J_new is NaN) every Q_uu from that step down is NaN, i.e. that is the failing case.
      if (J_new != J_new) { status = CILQR_EXIT_NUMERIC; break; }
      // Otherwise a rejection leaves X, U untouched, so every later iteration recomputes the same J_new == J_old and
      // rejects again until lamb > lamb_max or the iteration cap: only lamb and the counter change.
      for (;;) {
        lamb = lamb * kp.lamb_factor;
        if (lamb > kp.lamb_max) { status = CILQR_EXIT_LAMBDA_MAX; break; }
        if (++it >= max_it) { status = CILQR_EXIT_MAX_ITER; break; }
        ++iters;
      }
      break;
    }
    if (!(GENERAL ? riccati<true>(kp, N, rec, kK, lamb) : riccati_mfma<FSMEM, DIAG>(N, rec, kK, fwd, cst, 2.0 / kp.dt, lamb, sub))) {
      if (GENERAL) { status = CILQR_EXIT_NUMERIC; break; }
      handover = true;
      break;
    }
    __syncthreads();
    CILQR_STAMP(c_R)
    ++n_pass;
    if (DIAG) ++n_R;
    if (GENERAL) {
      forward_general(kp, N, Xc, Uc, kK, Xn, Un);
    } else if (!(FSMEM ? forward_smem(KParams(phase_params()), N, Xc, fwd, Xn, Un) : forward_fast<RECW>(KParams(phase_params()), N, Xc, Uc, kK, Xn, Un))) {
      handover = true;
      break;
    }
    __syncthreads();
    CILQR_STAMP(c_F)
    if (accept) {
      double* t0 = Xc; Xc = Xn; Xn = t0;
      double* t1 = Uc; Uc = Un; Un = t1;
      j_valid = false;
      lamb = lamb / kp.lamb_factor;
      if (fabs(J_new - J_old) < kp.tolerance) { status = CILQR_EXIT_TOLERANCE; break; }
    } else {
      lamb = lamb * kp.lamb_factor;
      if (lamb > kp.lamb_max) { status = CILQR_EXIT_LAMBDA_MAX; break; }
    }
    J_old = J_new;
  }

  if (!GENERAL) {
    int32_t* const hint = phase_args().hint_passes;
    if (lane == 0) {
      phase_args().redo[b] = handover ? 1 : 0;
      if (handover && hint) hint[b] = 63;  // redone by the GENERAL kernel: to the front of the next call's order
    }
    if (handover) return;  // outputs (and the in/out U) untouched: the GENERAL kernel starts from the same inputs
  }

  // ---- epilogue: X_result / U_result (:243-244) ----------------------------------------------------------------
  __syncthreads();
  const SolveArgs ae = phase_args();  // output pointers read here, not carried through the loop (cilqr_device.hpp)
  double* Uo = ae.U + (size_t)b * 2 * N;
  for (int i = lane; i < 2 * N; i += WAVE) Uo[i] = Uc[i];
  double* Xg = ae.X_out + (size_t)b * 4 * (N + 1);
  for (int i = lane; i < 4 * (N + 1); i += WAVE) Xg[i] = Xc[(i >> 2) * XR + (i & 3)];
  if (ae.J_out) {
    if (!j_valid) J_new = wave_sum_uniform(cost_only(ae.kp, N, lane, samp, S, grid, Xc, Uc));
    if (lane == 0) ae.J_out[b] = J_new;
  }
  if (lane == 0) {
    if (ae.iters_out) ae.iters_out[b] = iters;
    if (ae.status_out) ae.status_out[b] = status;
    if (ae.passes) ae.passes[b] = n_pass;
    if (ae.hint_passes && !GENERAL) ae.hint_passes[b] = n_pass;
  }
  if (DIAG && lane == 0 && a.diag) {
    const unsigned long long now_ = __builtin_readcyclecounter();
    unsigned long long* o = a.diag + (size_t)b * DIAG_SLOTS;
    o[0] = c_pro; o[1] = c_L; o[2] = c_R; o[3] = c_F; o[4] = now_ - tk; o[5] = n_L; o[6] = n_R; o[7] = now_ - tk0;
    for (int q = 0; q < 8; ++q) o[8 + q] = sub[q];
  }
#undef CILQR_STAMP
}

// ==== Two wavefronts per solve: an experiment, OFF by default (CILQR_PAIR_KERNEL at cilqr_create switches it on) ====================
// The serial phases R and F of the kernel above run at the issue rate of a lone wavefront, and phase L — lanes = timesteps — can
// only start when F has produced the whole new trajectory: a pass is R + F + L (config 2: 19 k + 19 k + 8 k ticks).  Here a solve
// is a workgroup of TWO wavefronts (the dispatcher places them on two SIMDs of one CU: tools/ubench_wave_place.hip):
//   wavefront 0 ("main") runs the rollout, R and F exactly as above;
//   wavefront 1 ("aux")  linearises the NEW trajectory while F is still producing it: step t can be linearised as soon as
//                        states t and t + 1 are in LDS, so the aux wavefront takes the horizon in chunks of up to 16 steps, four
//                        lanes per step (linearize_quads), each chunk as soon as its last state has arrived, and only the last
//                        chunk is still to do when F ends.
// Arrival of a state is seen in the data itself: before a rollout or forward pass starts, theta of every state it is going to
// write is set to a signalling-NaN bit pattern, which no arithmetic result can have (a computed NaN is quiet); F stores theta
// last (forward_smem<PUBLISH>), one wavefront's LDS stores execute in order, so "theta of state k is not the pattern" means
// states 0..k and controls 0..k-1 are there.  No flag stores on F's chain.  The two wavefronts meet at two barriers per pass: B1
// behind R (main: "records consumed, F starts"; carries main's decision to go on or stop) and B2 behind F (aux: "records, J and
// forward-pass rows of the new trajectory are complete").  Main never waits inside a phase; aux only waits for stores that main
// is certain to make (every rollout / forward pass runs to its end), and a bounded poll count turns a broken promise into a
// hand-over to the GENERAL kernel instead of a hang.
// MEASURED (round 3, profiles/r03_pair_kernel_ab.txt; tools/pair_ab.py, tools/phase_cycles.py with PAIR=1): it does not pay.
// A chunk of 16 steps costs the aux wavefront ≈ 4.3 k ticks — the linearisation of a step is one dependent chain (sincos → search
// window → scan → barrier exponentials → sums) that four lanes shorten only from 6.9 k to 4.3 k — so what main still waits for
// behind F is ≈ 6.4 k ticks instead of the 8.2 k of phase L; F itself goes from 372 to 392 ticks per step with the aux on
// another SIMD of its CU and to 424 once other solves' aux wavefronts share its SIMD (B = 1024: every SIMD holds one main and one
// aux, and a SIMD that issues for one does not issue for the other); launches: B = 64 0.406 against 0.390 ms, 256 0.410 / 0.407,
// 512 0.439 / 0.412, 1024 0.474 / 0.418.  Kept, with its tests, as the record of that A/B.
// Results: the statements of lin_step piece by piece (cilqr_device.hpp), a step's obstacle entries summed in entry order, J
// summed in the order of wave_sum_uniform — agreement with the one-wavefront kernel to rounding (observed ≤ 4e-13 on config 2;
// the compiler contracts a few sums differently in the two mappings), accept / reject paths equal on every solve tested.
constexpr unsigned long long THETA_PENDING = 0x7FF00000DEADBEEFull;
constexpr int PAIR_CTL = 4;       // doubles: {J of the trajectory in LDS, command | abort (two int32), -, -}
constexpr int PAIR_SPIN = 1 << 16;  // polls (≈ 150 ticks each) before the aux wavefront gives up on a state: ≫ any phase
constexpr int CMD_GO = 1, CMD_EXIT = 2;
constexpr int PAIR_LAST = 10;  // steps of the last chunk of a linearisation (linearize_quads)

template <int K>
__device__ __forceinline__ double quad_bcast(double v) {  // lane K of every quad to its four lanes (DPP quad_perm [K, K, K, K])
  return dpp_f64<K * 0x55>(v);
}
template <int K>
__device__ __forceinline__ int quad_bcast_i(int v) { return __builtin_amdgcn_update_dpp(v, v, K * 0x55, 0xF, 0xF, false); }

// Waits until state k of the trajectory in X has arrived (see above).  false: gave up.
// LONG_NAP: sleep ≈ 1000 ticks between polls instead of ≈ 64 — every poll is an LDS instruction in the queue that the main
// wavefront's stores (and the s_waitcnt lgkmcnt(0) of its every step) go through; only the last chunk's arrival is urgent.
template <bool LONG_NAP>
__device__ __forceinline__ bool state_arrived(const double* X, int k, int& budget) {
  const volatile unsigned long long* p = reinterpret_cast<const volatile unsigned long long*>(X + k * XR + 3);
  while (*p == THETA_PENDING) {
    if (--budget < 0) return false;
    if (LONG_NAP) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");  // the reads of the state lie behind the poll
  return true;
}
__device__ __forceinline__ void mark_pending(double* X, int N, int first_lane, int stride) {
  for (int t = 1 + first_lane; t <= N; t += stride) reinterpret_cast<unsigned long long*>(X + t * XR + 3)[0] = THETA_PENDING;
}

// Phase L of the two-wavefront kernel: one wavefront, lanes = 16 steps × 4; chunk c as soon as state min(16c + 16, N) is there.
// Writes the records, the forward-pass rows {x, y, v, theta, u0, u1} of the old trajectory, J per step and the sum J → ctl[0].
// false: a state did not arrive within the poll budget (ctl's abort word is set; the results are not to be used).
// DIAG: dg[0] += ticks spent waiting for states, dg[1] += ticks of the chunks' work, dg[2] += ticks from the arrival of the last
// state to the end (what the main wavefront waits for at the barrier), dg[3] += 1.
template <bool DIAG>
__device__ __forceinline__ bool linearize_quads(const KParams& kp, int N, int M, int lane, const double* samp, int S,
                                                const SampleGrid& grid, const double* X, const double* U, double* rec,
                                                const double* tab, const double* wts, double* fwd, double* Jt, double* ctl,
                                                unsigned long long* dg) {
  const int sub = lane & 3;
  int budget = PAIR_SPIN;
  bool ok = true;
  const LdsSamples at{samp, grid.xf, grid.dxs};
  unsigned long long tq = 0;
  if (DIAG) tq = __builtin_readcyclecounter();
  // Experiment switch of the stamped instantiation (flags bits 8-9; same results): 1 = poll for the states as usual but do the
  // chunks' work only when the main wavefront's phase is over (an extra barrier there); 2 = no polling either.  Tells the cost
  // of the polls and of the chunk work to the main wavefront's phase apart.
  const int dbg = DIAG ? (int)((phase_args().flags >> 8) & 3) : 0;
  if (dbg == 1 && ok) ok = state_arrived<true>(X, N, budget);
  if (dbg) __syncthreads();
  // Chunks of at most 16 steps, laid out from the END of the horizon: the last chunk holds the last PAIR_LAST steps — it can only
  // start when the forward pass is over, and the chunk before it must be done by then, i.e. must have started one chunk's work
  // (≈ 4 k ticks ≈ 10 forward steps) earlier; the first chunk takes what is left.
  for (int t0 = 0, t1; t0 < N; t0 = t1) {
    {
      const int tail = N - t0;  // steps still to do
      t1 = tail <= PAIR_LAST ? N : t0 + (tail - PAIR_LAST - 1) % 16 + 1;
    }
    if (ok && !dbg) ok = t1 < N ? state_arrived<true>(X, t1, budget) : state_arrived<false>(X, N, budget);
    if (DIAG) {
      const unsigned long long now_ = __builtin_readcyclecounter();
      dg[0] += now_ - tq;
      tq = now_;
    }
    const int t = t0 + (lane >> 2);
    const bool act = t < t1;
    const int tc = act ? t : t1 - 1;  // (lanes past the chunk repeat its last step and store nothing)
    const double* xr = X + tc * XR;
    const double* xn = X + (tc + 1) * XR;
    const double px = xr[0], py = xr[1], v = xr[2], th = xr[3], vn = xn[2], thn = xn[3];
    const double u0 = U[2 * tc], u1 = U[2 * tc + 1];
    // cos / sin of both headings: lanes 0, 1 of a quad evaluate theta_t, lanes 2, 3 theta_{t+1} (one sincos per lane)
    double sA, cA;
    sincos_loop(sub < 2 ? th : thn, sA, cA);
    const double ct = quad_bcast<0>(cA), st = quad_bcast<0>(sA), cn = quad_bcast<2>(cA), sn = quad_bcast<2>(sA);
    if (act && sub == 0) {  // the next forward pass reads the old state and control of step t through the scalar path
      double2* q = reinterpret_cast<double2*>(fwd + t * FREC + 10);
      q[0] = make_double2(px, py);
      q[1] = make_double2(v, th);
      q[2] = make_double2(u0, u1);
    }
    // closest path sample: the window of closest_sample, its candidates dealt round-robin to the quad, then the lexicographic
    // minimum of (distance, index) over the quad — the strict-< first minimum of the ascending scan (I/Constraints.cpp:43-56)
    int lo, hi;
    closest_window<false>(S, grid, px, py, at, lo, hi);
    double md = __builtin_huge_val();
    int best = 0x7fffffff;
    int s = lo + sub;
    if (sub == 0) { md = sample_dist(at, lo, px, py); best = lo; s += 4; }
    for (; s <= hi; s += 4) {
      const double d = sample_dist(at, s, px, py);
      if (d < md) { md = d; best = s; }
    }
    if (sub == 0 && md != md) md = -__builtin_huge_val();  // (a NaN first distance keeps the first sample, as in the scan)
    {
      double od = dpp_f64<0xB1>(md);
      int ob = __builtin_amdgcn_update_dpp(best, best, 0xB1, 0xF, 0xF, false);
      bool take = od < md || (od == md && ob < best);
      md = take ? od : md; best = take ? ob : best;
      od = dpp_f64<0x4E>(md);
      ob = __builtin_amdgcn_update_dpp(best, best, 0x4E, 0xF, 0xF, false);
      take = od < md || (od == md && ob < best);
      best = take ? ob : best;
    }
    const double cx = fma(grid.dxs, (double)best, grid.xf), cy = samp[best];
    // tracking cost, as lin_step forms it
    const double dx = px - cx, dy = py - cy, dv = v - kp.desired_speed;
    StepSums a{0.0, 0.0, 0.0, 0.0, 0.0}, ao{0.0, 0.0, 0.0, 0.0, 0.0};  // even / odd entries (obstacle_loop's SPLIT order)
    Rec c;
    c.lx2 = (2 * kp.w_vel) * dv;
    const double J = stage_cost(kp, dx, dy, dv, u0, u1);
    // obstacles: lane `sub` of the quad evaluates entries sub, sub + 4, …; their terms join the sums in entry order
    const ObsConsts oc = make_obs_consts(kp, px, py, ct, st);
    for (int m0 = 0; m0 < M; m0 += 4) {
      const int m = min(m0 + sub, M - 1);
      const double2* pe = reinterpret_cast<const double2*>(tab + ((size_t)m * N + tc) * TABF);
      const double2 ea = pe[0], eb = pe[1], ec = pe[2];
      const ObsEntry e{ea.x, ea.y, eb.x, eb.y, ec.x, ec.y};
      const double w = wts ? wts[m] : kp.w_obstacle;
      ObsPrep p;
      obs_prep(oc, e, p);
      const bool need = m0 + sub < M && obs_needed(p);
      ObsTerms g{0.0, 0.0, 0.0, 0.0, 0.0};
      if (__builtin_amdgcn_ballot_w64(need) != 0) g = obs_terms(oc, e, p);
      const double we = need ? w : 0.0;  // (lin_step<…, LANE_EXACT>: a step that does not need the entry adds exactly nothing)
      obs_accumulate(a, ObsTerms{quad_bcast<0>(g.gx), quad_bcast<0>(g.gy), quad_bcast<0>(g.gxx), quad_bcast<0>(g.gxy), quad_bcast<0>(g.gyy)}, quad_bcast<0>(we));
      if (m0 + 1 < M) obs_accumulate(ao, ObsTerms{quad_bcast<1>(g.gx), quad_bcast<1>(g.gy), quad_bcast<1>(g.gxx), quad_bcast<1>(g.gxy), quad_bcast<1>(g.gyy)}, quad_bcast<1>(we));
      if (m0 + 2 < M) obs_accumulate(a, ObsTerms{quad_bcast<2>(g.gx), quad_bcast<2>(g.gy), quad_bcast<2>(g.gxx), quad_bcast<2>(g.gxy), quad_bcast<2>(g.gyy)}, quad_bcast<2>(we));
      if (m0 + 3 < M) obs_accumulate(ao, ObsTerms{quad_bcast<3>(g.gx), quad_bcast<3>(g.gy), quad_bcast<3>(g.gxx), quad_bcast<3>(g.gxy), quad_bcast<3>(g.gyy)}, quad_bcast<3>(we));
    }
    // control cost: one of its four exponentials per lane of the quad
    double a1, a2, a3, a4;
    ctrl_args(kp, u0, u1, v, a1, a2, a3, a4);
    const double ee = exp_fast(sub == 0 ? a1 : sub == 1 ? a2 : sub == 2 ? a3 : a4);
    a.lx0 += ao.lx0; a.lx1 += ao.lx1; a.h00 += ao.h00; a.h01 += ao.h01; a.h11 += ao.h11;
    state_terms(kp, dx, dy, a, c.lx0, c.lx1, c.l00, c.l01, c.l11);
    ctrl_terms(kp, u0, u1, quad_bcast<0>(ee), quad_bcast<1>(ee), quad_bcast<2>(ee), quad_bcast<3>(ee), c);
    ab_terms(kp, u0, vn, cn, sn, c);
    if (act && sub == 0) {
      double* r = rec + t * RECF;
      const double ih = 2.0 / kp.dt;  // production records: acceleration in units of (dt/2)·u0 (riccati_mfma)
      r[0] = c.lx0; r[1] = c.lx1; r[2] = c.lx2; r[3] = c.l00; r[4] = c.l01; r[5] = c.l11;
      r[6] = c.lu0 * ih; r[7] = c.lu1; r[8] = c.luu0 * (ih * ih); r[9] = c.luu1;
      r[10] = c.al; r[11] = c.be; r[12] = c.ga; r[13] = c.de;
      Jt[t] = J;
    }
    if (DIAG) {
      const unsigned long long now_ = stamp_after(J);
      dg[1] += now_ - tq;
      if (t1 >= N) dg[2] -= tq;
      tq = now_;
    }
  }
  // J = sum over the steps, in the order of the one-wavefront kernel: lanes = steps (t, t + 64, …), then wave_sum_uniform
  double part = 0.0;
  for (int t = lane; t < N; t += WAVE) part += Jt[t];
  const double J = wave_sum_uniform(part);
  if (lane == 0) {
    ctl[0] = J;
    if (!ok) reinterpret_cast<int*>(ctl + 1)[1] = 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the forward-pass rows have reached L2 before the barrier that follows
  if (DIAG) {
    dg[2] += __builtin_readcyclecounter();
    dg[3] += 1;
  }
  return ok;
}

// LDS: the compact layout of cilqr_solve_kernel (samp, X, U, rec, cst, tab), then J per step and the control block.
template <bool DIAG>
__global__ __launch_bounds__(2 * WAVE) void cilqr_solve_pair_kernel(SolveArgs a) {
  unsigned long long tk0 = 0, tk = 0, c_pro = 0, c_L = 0, c_R = 0, c_F = 0, n_L = 0, n_R = 0;
  unsigned long long sub[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (DIAG) tk0 = tk = __builtin_readcyclecounter();
#define CILQR_STAMP(acc)                                 \
  if (DIAG) {                                            \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    acc += now_ - tk;                                    \
    tk = now_;                                           \
  }
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if ((int)blockIdx.x >= a.B) return;
  const int b = __builtin_amdgcn_readfirstlane(a.order ? a.order[blockIdx.x] : (int)blockIdx.x);
  if (b >= a.B) return;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const KParams kp = a.kp;
  const int N = a.N, M = a.M, S = kp.n_samples;
  double* samp = lds;
  double* Xa = samp + ((S + 1) & ~1);
  double* Ua = Xa + (N + 1) * XR;
  double* rec = Ua + 2 * N;
  double* cst = rec + N * RECF;
  double* tab = cst + RCST;
  double* Jt = tab + (size_t)M * TABF * N;
  double* ctl = Jt + ((N + 1) & ~1);
  double* fwd = a.fwd + (size_t)b * (N + 1) * FREC;
  int* const cmd = reinterpret_cast<int*>(ctl + 1);  // [0] command, [1] abort

  // ---- prologue, both wavefronts -----------------------------------------------------------------------------------
  SampleGrid grid;
  make_sample_grid(grid, a.xplan_fl[2 * b], a.xplan_fl[2 * b + 1], S);
  {
    const double* pc = a.poly + (size_t)b * CILQR_POLY_COEFFS;
    for (int s = tid; s < S; s += 2 * WAVE) {
      double xs;
      sample_xy(grid, pc, s, xs, samp[s]);
    }
  }
  const double* Ug = a.U + (size_t)b * 2 * N;
  for (int i = tid; i < 2 * N; i += 2 * WAVE) Ua[i] = Ug[i];
  if (tid < 16 && (tid & 7) < 5) cst[(tid & 7) + (tid >> 3) * RECF] = (tid & 7) == 0 ? 0.0 : (tid & 7) == 1 ? 1.0 : (tid & 7) == 2 ? kp.dt : (tid & 7) == 3 ? kp.w_vel * 2 : 2.0;
  const double* wts = a.obs_weight ? a.obs_weight + (size_t)b * M : nullptr;
  for (int m = wave; m < M; m += 2) {  // obstacle table, I/Obstacle.cpp:41-62
    for (int t = lane; t < N; t += WAVE) {
      const ObsEntry e = make_obs_entry(kp, a.obs_pose + (((size_t)b * M + m) * N + t) * 4, a.obs_dim + (((size_t)b * M + m) * N + t) * 2);
      double* o = tab + ((size_t)m * N + t) * TABF;
      o[0] = e.ox; o[1] = e.oy; o[2] = e.co; o[3] = e.so; o[4] = e.ia2; o[5] = e.ib2;
    }
  }
  mark_pending(Xa, N, tid, 2 * WAVE);
  if (tid == 0) { cmd[0] = 0; cmd[1] = 0; }
  __syncthreads();

  if (wave != 0) {
    // ---- the aux wavefront -------------------------------------------------------------------------------------------
    {  // largest step between adjacent path samples in y: closest_sample's second window (cilqr_device.hpp)
      double m = 0.0;
      for (int q = lane; q + 1 < S; q += WAVE) {
        const double d = fabs(samp[q + 1] - samp[q]);
        m = fmax(m, d == d ? d : __builtin_huge_val());
      }
      grid.dmax = wave_max_uniform(m);
    }
    unsigned long long dg[4] = {0, 0, 0, 0};
    for (;;) {
      linearize_quads<DIAG>(KParams(phase_params()), N, M, lane, samp, S, grid, Xa, Ua, rec, tab, wts, fwd, Jt, ctl, dg);
      __syncthreads();  // B2: the new trajectory is linearised
      __syncthreads();  // B1: main has decided (and, going on, has run R)
      if (*reinterpret_cast<volatile int*>(cmd) != CMD_GO) break;
    }
    if (DIAG && lane == 0 && a.diag) {  // slots 8..11: the aux wavefront's own account (cilqr_set_diag_buffer)
      unsigned long long* o = a.diag + (size_t)b * DIAG_SLOTS + 8;
      o[0] = dg[0]; o[1] = dg[1]; o[2] = dg[2]; o[3] = dg[3];
    }
    return;
  }

  // ---- the main wavefront ------------------------------------------------------------------------------------------------
  bool handover = !rollout_fast<true>(kp, N, a.x0 + (size_t)b * 4, Ua, Xa) || (a.flags & CILQR_FLAG_GENERAL_ONLY) != 0;  // nominal rollout, I/iLQR.cpp:51-62
  const int dbg = DIAG ? (int)((a.flags >> 8) & 3) : 0;  // (linearize_quads)
  if (dbg) __syncthreads();
  __syncthreads();  // B2
  CILQR_STAMP(c_pro)
  double J_old = DBL_MAX, lamb = 1.0, J_new = 0.0;
  int iters = 0, status = CILQR_EXIT_MAX_ITER, n_pass = 0;
  const int max_it = kp.max_iterations;
  // iteration loop, I/iLQR.cpp:204-239, as in cilqr_solve_kernel (early exit at the first rejection; forward pass in place)
  for (int it = 0; it < max_it && !handover; ++it) {
    ++iters;
    if (reinterpret_cast<volatile int*>(cmd)[1] != 0) { handover = true; break; }  // the aux wavefront gave up on a state
    J_new = *reinterpret_cast<volatile double*>(ctl);
    if (DIAG) ++n_L;
    const bool accept = J_new < J_old;
    if (!accept) {
      if (J_new != J_new) { status = CILQR_EXIT_NUMERIC; break; }
      for (;;) {
        lamb = lamb * kp.lamb_factor;
        if (lamb > kp.lamb_max) { status = CILQR_EXIT_LAMBDA_MAX; break; }
        if (++it >= max_it) { status = CILQR_EXIT_MAX_ITER; break; }
        ++iters;
      }
      break;
    }
    mark_pending(Xa, N, lane, WAVE);  // the forward pass below rewrites states 1..N; the aux wavefront sleeps at B1 meanwhile
    CILQR_STAMP(c_L)
    if (!riccati_mfma<true, DIAG>(N, rec, rec, fwd, cst, 2.0 / kp.dt, lamb, sub)) { handover = true; break; }
    if (lane == 0) *reinterpret_cast<volatile int*>(cmd) = CMD_GO;
    __syncthreads();  // B1
    CILQR_STAMP(c_R)
    ++n_pass;
    if (DIAG) ++n_R;
    const bool f_ok = forward_smem<true>(KParams(phase_params()), N, Xa, fwd, Xa, Ua);
    CILQR_STAMP(c_F)
    if (dbg) __syncthreads();
    __syncthreads();  // B2
    CILQR_STAMP(c_L)
    if (!f_ok) { handover = true; break; }
    lamb = lamb / kp.lamb_factor;
    if (fabs(J_new - J_old) < kp.tolerance) {
      status = CILQR_EXIT_TOLERANCE;
      J_new = *reinterpret_cast<volatile double*>(ctl);  // get_J of the final trajectory
      break;
    }
    J_old = J_new;
    if (it + 1 >= max_it) J_new = *reinterpret_cast<volatile double*>(ctl);
  }
  if (lane == 0) *reinterpret_cast<volatile int*>(cmd) = CMD_EXIT;
  __syncthreads();  // B1: releases the aux wavefront

  int32_t* const hint = phase_args().hint_passes;
  if (lane == 0) {
    phase_args().redo[b] = handover ? 1 : 0;
    if (handover && hint) hint[b] = 63;
  }
  if (handover) return;  // outputs (and the in/out U) untouched: the GENERAL kernel starts from the same inputs

  // ---- epilogue: X_result / U_result (:243-244) ----------------------------------------------------------------
  const SolveArgs ae = phase_args();
  double* Uo = ae.U + (size_t)b * 2 * N;
  for (int i = lane; i < 2 * N; i += WAVE) Uo[i] = Ua[i];
  double* Xg = ae.X_out + (size_t)b * 4 * (N + 1);
  for (int i = lane; i < 4 * (N + 1); i += WAVE) Xg[i] = Xa[(i >> 2) * XR + (i & 3)];
  if (lane == 0) {
    if (ae.J_out) ae.J_out[b] = J_new;
    if (ae.iters_out) ae.iters_out[b] = iters;
    if (ae.status_out) ae.status_out[b] = status;
    if (ae.passes) ae.passes[b] = n_pass;
    if (ae.hint_passes) ae.hint_passes[b] = n_pass;
  }
  if (DIAG && lane == 0 && a.diag) {
    const unsigned long long now_ = __builtin_readcyclecounter();
    unsigned long long* o = a.diag + (size_t)b * DIAG_SLOTS;
    o[0] = c_pro; o[1] = c_L; o[2] = c_R; o[3] = c_F; o[4] = now_ - tk; o[5] = n_L; o[6] = n_R; o[7] = now_ - tk0;
    for (int q = 4; q < 8; ++q) o[8 + q] = sub[q];  // (slots 8..11: written by the aux wavefront)
  }
#undef CILQR_STAMP
}

// ==== Sampled obstacles, two wavefronts per solve (BASELINE config 3) =============================================================
// With hundreds of obstacle entries per step phase L is 80 % of a pass (config 3: L 152 k ticks against R 19.6 k + F 21.9 k), and a
// batch of more solves than SIMDs ends when its LONGEST solve does: 20 passes × 193 k ticks, however well the rest of the batch
// keeps the chip busy — on a planner's tick sequence the launch lasts about twice what its work alone would take
// (profiles/r03_schedule_hint_ticks.txt; pass counts of the previous tick predict too little for the dispatch order to help).
// Here a solve is a workgroup of two wavefronts that share phase L: both take lanes = timesteps, wavefront 0 the obstacles
// 0, 2, 4, … of every step (and everything else of the step: closest sample, tracking, control barrier, Jacobians), wavefront 1
// the obstacles 1, 3, 5, …; wavefront 1 leaves its five sums per step in LDS, wavefront 0 adds them to its own behind one
// barrier and goes on alone through R and F.  A pass is then ≈ 76 k + 41 k ticks instead of 193 k, the work the same: four
// workgroups of two wavefronts per CU instead of seven of one (the kernel's ≈ 240 vector registers allow eight wavefronts).
// Sums: the entries of a step are added in two interleaved halves instead of one run — results differ from the one-wavefront
// kernel in the last bits (tolerance against the oracle unchanged), deterministic, independent of the batch around the solve.
// Horizons up to 64 (one step per lane: wavefront 0 holds its step's record in registers across the barrier).
// W = wavefronts per solve (2 or 4): wavefront w takes the obstacles w, w + W, …; wavefront 0 adds the others' sums in the order 1, 2, 3.
// DIAG: wavefront 0 stamps its phases into a.diag[b] = {prologue, L (its own share + the wait for the others + the combine), R, F,
// epilogue, #L, #R, total}, as cilqr_solve_kernel does.
// UNC: an uncertainty map is set — its term on the last wavefront, behind that wavefront's obstacle share; main adds the five scaled sums
// per step behind all obstacle sums (the order of the one-wavefront kernel, where unc_cost_add joins the finished record).
template <int W, bool DIAG, bool UNC = false>
__global__ __launch_bounds__(W * WAVE, 2) void cilqr_solve_split_kernel(SolveArgs a) {  // (two wavefronts per SIMD: ≤ 256 vector registers)
  unsigned long long tk0 = 0, tk = 0, c_pro = 0, c_L = 0, c_R = 0, c_F = 0, n_L = 0, n_R = 0;
  if (DIAG) tk0 = tk = __builtin_readcyclecounter();
#define CILQR_STAMP(acc)                                 \
  if (DIAG) {                                            \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    acc += now_ - tk;                                    \
    tk = now_;                                           \
  }
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if ((int)blockIdx.x >= a.B) return;
  const int b = __builtin_amdgcn_readfirstlane(a.order ? a.order[blockIdx.x] : (int)blockIdx.x);
  if (b >= a.B) return;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const KParams kp = a.kp;
  const int N = a.N, M = a.M, S = kp.n_samples, NSMP = a.n_samples;
  double* samp = lds;
  double* Xa = samp + ((S + 1) & ~1);
  double* Ua = Xa + (N + 1) * XR;
  double* rec = Ua + 2 * N;
  double* cst = rec + N * RECF;
  double* off = cst + RCST;
  double* rmax = off + (size_t)M * NSMP * OFFF;
  double* part = rmax + 2 * M;                        // [W - 1][N][5]: the other wavefronts' sums of a step's obstacle terms
  double* umap = part + (size_t)(W - 1) * 5 * N;      // UNC: [N][5], the map term's scaled sums of a step
  double* ctl = part + ((((W - 1) + (UNC ? 1 : 0)) * 5 * N + 1) & ~1);  // command word
  int* const cmd = reinterpret_cast<int*>(ctl);
  double* tab = a.obs_tab + (size_t)b * M * NOMF * N;

  // ---- prologue, both wavefronts -----------------------------------------------------------------------------------------------
  SampleGrid grid;
  make_sample_grid(grid, a.xplan_fl[2 * b], a.xplan_fl[2 * b + 1], S);
  {
    const double* pc = a.poly + (size_t)b * CILQR_POLY_COEFFS;
    for (int s = tid; s < S; s += W * WAVE) {
      double xs;
      sample_xy(grid, pc, s, xs, samp[s]);
    }
  }
  const double* Ug = a.U + (size_t)b * 2 * N;
  for (int i = tid; i < 2 * N; i += W * WAVE) Ua[i] = Ug[i];
  if (tid < 16 && (tid & 7) < 5) cst[(tid & 7) + (tid >> 3) * RECF] = (tid & 7) == 0 ? 0.0 : (tid & 7) == 1 ? 1.0 : (tid & 7) == 2 ? kp.dt : (tid & 7) == 3 ? kp.w_vel * 2 : 2.0;
  if (tid == 0) cmd[0] = 0;
  sampled_prologue<W * WAVE>(a, kp, b, N, M, tab, off, rmax);
  UncPose upose{0, 0, 1, 0};
  if (UNC) upose = unc_pose(a.unc, b);
  __syncthreads();
  bool handover = false;
  if (wave == 0) {
    {  // largest step between adjacent path samples in y: closest_sample's second window (cilqr_device.hpp)
      double m = 0.0;
      for (int q = lane; q + 1 < S; q += WAVE) {
        const double d = fabs(samp[q + 1] - samp[q]);
        m = fmax(m, d == d ? d : __builtin_huge_val());
      }
      grid.dmax = wave_max_uniform(m);
    }
    handover = !rollout_fast(kp, N, a.x0 + (size_t)b * 4, Ua, Xa) || (a.flags & CILQR_FLAG_GENERAL_ONLY) != 0;  // nominal rollout, I/iLQR.cpp:51-62
    if (handover && lane == 0) *reinterpret_cast<volatile int*>(cmd) = CMD_EXIT;
  }
  __syncthreads();  // the trajectory is in LDS

  const int t = lane;
  const bool act = t < N;
  const int n_mine = (M - wave + W - 1) / W;  // obstacles wave, wave + W, … < M
  if (wave != 0) {
    // ---- wavefronts 1 … W-1: their share of every step's obstacle terms, pass after pass ------------------------------------------------
    for (;;) {
      if (*reinterpret_cast<volatile int*>(cmd) == CMD_EXIT) break;
      if (act) {
        const KParams kpl = phase_params();
        const double* xr = Xa + t * XR;
        SampledSource src{tab, off, rmax, Xa, N, NSMP, M, a.samp_w, kpl.ego_front, kpl.ego_rear,
                          sqrt(1.0 + 64.0 / kpl.q2_front), sqrt(1.0 + 64.0 / kpl.q2_rear)};
        src.o_first = wave; src.o_stride = W;
        StepSums s5{0.0, 0.0, 0.0, 0.0, 0.0};
        obstacle_loop<true, false, false, false>(make_obs_consts(kpl, xr[0], xr[1], xr[4], xr[5]), n_mine * NSMP, src.at(t), s5);
        double* q = part + ((size_t)(wave - 1) * N + t) * 5;
        q[0] = s5.lx0; q[1] = s5.lx1; q[2] = s5.h00; q[3] = s5.h01; q[4] = s5.h11;
        if (UNC && wave == W - 1) {  // the map term, from zero: w·vx, w·mx as unc_cost_add would add them to a finished record
          double g0 = 0.0, g1 = 0.0, h00 = 0.0, h01 = 0.0, h11 = 0.0;
          unc_cost_add(phase_args().unc, upose, b, xr[0], xr[1], xr[4], xr[5], g0, g1, h00, h01, h11);
          double* qu = umap + (size_t)t * 5;
          qu[0] = g0; qu[1] = g1; qu[2] = h00; qu[3] = h01; qu[4] = h11;
        }
      }
      __syncthreads();  // A: the sums are in LDS
      __syncthreads();  // B: wavefront 0 has decided, and, going on, has written the next trajectory
    }
    return;
  }

  // ---- wavefront 0 ---------------------------------------------------------------------------------------------------------------
  CILQR_STAMP(c_pro)
  double J_old = DBL_MAX, lamb = 1.0, J_new = 0.0;
  int iters = 0, status = CILQR_EXIT_MAX_ITER, n_pass = 0;
  bool j_valid = false;
  const int max_it = kp.max_iterations;
  for (int it = 0; it < max_it && !handover; ++it) {
    ++iters;
    {  // phase L: this wavefront's share, then wavefront 1's sums on top
      const KParams kpl = phase_params();
      Rec c{};
      double Jt = 0.0;
      if (act) {
        const double* xr = Xa + t * XR;
        const double* xn = Xa + (t + 1) * XR;
        const int cs = closest_sample<false>(S, grid, xr[0], xr[1], LdsSamples{samp, grid.xf, grid.dxs});
        SampledSource src{tab, off, rmax, Xa, N, NSMP, M, a.samp_w, kpl.ego_front, kpl.ego_rear,
                          sqrt(1.0 + 64.0 / kpl.q2_front), sqrt(1.0 + 64.0 / kpl.q2_rear)};
        src.o_first = 0; src.o_stride = W;
        Jt = lin_step<true, false, false, false>(kpl, xr[0], xr[1], xr[2], xr[4], xr[5], Ua[2 * t], Ua[2 * t + 1], xn[2], xn[4], xn[5],
                                           fma(grid.dxs, (double)cs, grid.xf), samp[cs], n_mine * NSMP, src.at(t), c);
      }
      __syncthreads();  // A
      if (act) {
#pragma unroll
        for (int w = 1; w < W; ++w) {
          const double* q = part + ((size_t)(w - 1) * N + t) * 5;
          c.lx0 += q[0]; c.lx1 += q[1]; c.l00 += q[2]; c.l01 += q[3]; c.l11 += q[4];
        }
        if (UNC) {  // the map term joins the finished sums (I/Constraints.cpp:188-201)
          const double* qu = umap + (size_t)t * 5;
          c.lx0 += qu[0]; c.lx1 += qu[1]; c.l00 += qu[2]; c.l01 += qu[3]; c.l11 += qu[4];
        }
        double* r = rec + t * RECF;
        const double ih = 2.0 / kpl.dt;  // production records: acceleration in units of (dt/2)·u0 (riccati_mfma)
        r[0] = c.lx0; r[1] = c.lx1; r[2] = c.lx2; r[3] = c.l00; r[4] = c.l01; r[5] = c.l11;
        r[6] = c.lu0 * ih; r[7] = c.lu1; r[8] = c.luu0 * (ih * ih); r[9] = c.luu1;
        r[10] = c.al; r[11] = c.be; r[12] = c.ga; r[13] = c.de;
      }
      J_new = wave_sum_uniform(Jt);
      j_valid = true;
    }
    CILQR_STAMP(c_L)
    if (DIAG) ++n_L;
    const bool accept = J_new < J_old;
    bool stop = false;
    if (!accept) {
      if (J_new != J_new) {
        status = CILQR_EXIT_NUMERIC;
      } else {
        for (;;) {
          lamb = lamb * kp.lamb_factor;
          if (lamb > kp.lamb_max) { status = CILQR_EXIT_LAMBDA_MAX; break; }
          if (++it >= max_it) { status = CILQR_EXIT_MAX_ITER; break; }
          ++iters;
        }
      }
      stop = true;
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the records are in LDS (one wavefront reads what its own lanes wrote)
      if (!riccati_mfma<false, false>(N, rec, rec, nullptr, cst, 2.0 / kp.dt, lamb, nullptr)) {
        handover = true;
        stop = true;
      } else {
        CILQR_STAMP(c_R)
        ++n_pass;
        if (DIAG) ++n_R;
        if (!forward_fast<RECF>(KParams(phase_params()), N, Xa, Ua, rec, Xa, Ua)) {
          handover = true;
          stop = true;
        } else {
          j_valid = false;
          lamb = lamb / kp.lamb_factor;
          if (fabs(J_new - J_old) < kp.tolerance) { status = CILQR_EXIT_TOLERANCE; stop = true; }
          J_old = J_new;
          if (it + 1 >= max_it) stop = true;
        }
      }
    }
    if (stop && lane == 0) *reinterpret_cast<volatile int*>(cmd) = CMD_EXIT;
    __syncthreads();  // B
    CILQR_STAMP(c_F)
    if (stop) break;
  }

  int32_t* const hint = phase_args().hint_passes;
  if (lane == 0) {
    phase_args().redo[b] = handover ? 1 : 0;
    if (handover && hint) hint[b] = 63;
  }
  if (handover) return;  // outputs (and the in/out U) untouched: the GENERAL kernel starts from the same inputs

  // ---- epilogue: X_result / U_result (:243-244) ------------------------------------------------------------------------------
  const SolveArgs ae = phase_args();
  double* Uo = ae.U + (size_t)b * 2 * N;
  for (int i = lane; i < 2 * N; i += WAVE) Uo[i] = Ua[i];
  double* Xg = ae.X_out + (size_t)b * 4 * (N + 1);
  for (int i = lane; i < 4 * (N + 1); i += WAVE) Xg[i] = Xa[(i >> 2) * XR + (i & 3)];
  if (ae.J_out) {
    if (!j_valid) J_new = wave_sum_uniform(cost_only(ae.kp, N, lane, samp, S, grid, Xa, Ua));
    if (lane == 0) ae.J_out[b] = J_new;
  }
  if (lane == 0) {
    if (ae.iters_out) ae.iters_out[b] = iters;
    if (ae.status_out) ae.status_out[b] = status;
    if (ae.passes) ae.passes[b] = n_pass;
    if (ae.hint_passes) ae.hint_passes[b] = n_pass;
  }
  if (DIAG && lane == 0 && a.diag) {
    const unsigned long long now_ = __builtin_readcyclecounter();
    unsigned long long* o = a.diag + (size_t)b * DIAG_SLOTS;
    o[0] = c_pro; o[1] = c_L; o[2] = c_R; o[3] = c_F; o[4] = now_ - tk; o[5] = n_L; o[6] = n_R; o[7] = now_ - tk0;
    for (int q = 8; q < DIAG_SLOTS; ++q) o[q] = 0;
  }
#undef CILQR_STAMP
}

// ==== Static obstacles, two or three wavefronts per solve that share phase L (BASELINE config 2's shape, up to two solves per SIMD) ====
// A batch of at most one solve per SIMD ends when its LONGEST solve does (config 2: 13 of 1024 solves run all 20 passes, the mean
// solve 7.8), so what counts is the length of one pass — R + F + L = 18.6 k + 17.8 k + 8.3 k ticks at N = 50, M = 4 — and R and F
// are serial chains at the issue rate of a lone wavefront.  Phase L is not one chain: the closest-sample search with the tracking
// terms (≈ 3.1 k ticks per lane) is independent of the obstacle, control-barrier and Jacobian terms (≈ 3.4 k), which need the
// headings' cos / sin but not the closest sample.  Here a solve is a workgroup of two wavefronts, lanes = timesteps in both:
//   wavefront 0 ("main"): rollout, R, F as in cilqr_solve_kernel; of phase L the forward-pass rows, the closest-sample search, the
//                         tracking terms, J and the control barrier; then, behind barrier A, l_x / l_xx = tracking terms + the
//                         other wavefront's sums;
//   wavefront 1 ("aux"):  cos / sin of every heading (one state per lane; a step's next heading comes from the lane above), the
//                         obstacle terms summed from zero into `part`, the Jacobians' record slots straight into the records;
//                         then it sleeps at barrier B through R and F.
// Unlike cilqr_solve_pair_kernel (above: the aux wavefront working BEHIND the forward pass, measured slower) nothing is pipelined:
// both wavefronts work on phase L at the same time and only then, the aux wavefront issues nothing while main runs R and F, and the
// two meet at two barriers per pass.  While every SIMD still holds a main wavefront the aux wavefronts take issue slots from the
// main wavefronts of OTHER solves (work-conserving: no gain, no loss to speak of); once the short solves have ended — the larger
// part of the launch — the long ones have their SIMDs' partners to themselves and a pass is ≈ 4.5 k ticks shorter.
// W = 3 (up to three quarters of a solve per SIMD: cilqr_api.cpp, pick_share): the obstacle terms are split once more — wavefront 1 sums the entries of EVEN index, wavefront 2
// those of odd index, each in one chain, in order, which is obstacle_loop's own order of summation (cilqr_device.hpp, SPLIT); wavefront
// 2 also takes the Jacobian slots and the control barrier, and main adds odd sums to even sums as obstacle_loop does.  The solves that
// decide a launch are the ones with every obstacle close (their aux wavefront took 6.1 k ticks per call where the mean took 4.3 k).
// Bits: the statements are lin_step's, piece by piece (cilqr_device.hpp: obstacle sums from zero, state_terms), so a record does
// not depend on which wavefront formed which slot — results are bit-identical to cilqr_solve_kernel's (tests/test_gpu_parity.py,
// test_share_kernel_changes_no_bit).  Horizons up to 64 on one step per lane, up to 127 on two (LONG); obstacle table in LDS; with or
// without an uncertainty map (UNC); early-exit mode.
// DIAG: a.diag[b] = {prologue, L (main's share + the wait at barrier A + the combine), R, F, epilogue, #L, #R, total, aux: busy
// ticks, aux: calls, main: ticks waiting at barrier A, 0…}.
// LONG: horizons 65 … 127, two steps per lane (two wavefronts only: the registers of a second step do not fit three wavefronts per SIMD).
// UNC: an uncertainty map is set (cilqr_set_uncertainty_map*) — the reference's own mode of operation.  The map term (footprint probes,
// bilinear lookups, one exponential each: ≈ 4 k ticks per call, as much as all obstacles of config 2) goes to the LAST aux wavefront,
// beside the Jacobians (and, with three wavefronts, the control barrier); the obstacle terms — all of them, even and odd chain — to
// wavefront 1.  It leaves its five scaled sums per step in LDS and main adds them to l_x / l_xx behind the obstacle sums, the order of
// the one-wavefront kernel (unc_cost_add adds w·vx, w·mx to the finished record, I/Constraints.cpp:188-201).  Two wavefronts per SIMD
// (the map code needs more than 168 registers), so three wavefronts per solve up to half a solve per SIMD.
template <int W, bool LONG, bool DIAG, bool UNC = false>
__global__ __launch_bounds__(W * WAVE, UNC ? 2 : W) void cilqr_solve_share_kernel(SolveArgs a) {  // (W wavefronts per SIMD: ≤ 256 / 168 vector registers)
  static_assert(!(LONG && W != 2), "the long-horizon form is built for two wavefronts");
  constexpr int STEPS = LONG ? 2 : 1;  // steps per lane
  unsigned long long tk0 = 0, tk = 0, c_pro = 0, c_L = 0, c_R = 0, c_F = 0, n_L = 0, n_R = 0, c_wait = 0;
  if (DIAG) tk0 = tk = __builtin_readcyclecounter();
#define CILQR_STAMP(acc)                                 \
  if (DIAG) {                                            \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    acc += now_ - tk;                                    \
    tk = now_;                                           \
  }
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if ((int)blockIdx.x >= a.B) return;
  const int b = __builtin_amdgcn_readfirstlane(a.order ? a.order[blockIdx.x] : (int)blockIdx.x);
  if (b >= a.B) return;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const KParams kp = a.kp;
  const int N = a.N, M = a.M, S = kp.n_samples;
  double* samp = lds;  // the compact layout of cilqr_solve_kernel, then `part` and the command word
  double* Xa = samp + ((S + 1) & ~1);
  double* Ua = Xa + (N + 1) * XR;
  double* rec = Ua + 2 * N;
  double* cst = rec + N * RECF;
  double* tab = cst + RCST;
  double* part = tab + (size_t)M * TABF * N;  // [W - 1][N][5]: the aux wavefronts' sums of a step's obstacle terms
  double* umap = part + (size_t)(W - 1) * 5 * N;      // UNC: [N][5], the map term's scaled sums of a step
  double* ctl = part + ((((W - 1) + (UNC ? 1 : 0)) * 5 * N + 1) & ~1);  // {command word, dmax, curvature bound, -}
  int* const cmd = reinterpret_cast<int*>(ctl);
  double* fwd = a.fwd + (size_t)b * (N + 1) * FREC;

  // ---- prologue: the controls by both wavefronts; then the nominal rollout on the main wavefront while the aux wavefront fills the
  // path samples and the obstacle table (the rollout needs neither) ------------------------------------------------------------------
  SampleGrid grid;
  make_sample_grid(grid, a.xplan_fl[2 * b], a.xplan_fl[2 * b + 1], S);
  const double* Ug = a.U + (size_t)b * 2 * N;
  for (int i = tid; i < 2 * N; i += W * WAVE) Ua[i] = Ug[i];
  if (tid < 16 && (tid & 7) < 5) cst[(tid & 7) + (tid >> 3) * RECF] = (tid & 7) == 0 ? 0.0 : (tid & 7) == 1 ? 1.0 : (tid & 7) == 2 ? kp.dt : (tid & 7) == 3 ? kp.w_vel * 2 : 2.0;
  if (tid == 0) cmd[0] = 0;
  const double* wts = a.obs_weight ? a.obs_weight + (size_t)b * M : nullptr;
  UncPose upose{0, 0, 1, 0};
  if (UNC) upose = unc_pose(a.unc, b);
  __syncthreads();  // the controls are in LDS
  bool handover = false;
  if (wave == 0) {
    handover = !rollout_fast(kp, N, a.x0 + (size_t)b * 4, Ua, Xa) || (a.flags & CILQR_FLAG_GENERAL_ONLY) != 0;  // nominal rollout, I/iLQR.cpp:51-62
    if (handover && lane == 0) *reinterpret_cast<volatile int*>(cmd) = CMD_EXIT;
  } else {
    // (W = 3: the two aux wavefronts share the table; samples and their largest step on the first)
    const int al = lane + (wave - 1) * WAVE, an = (W - 1) * WAVE;
    for (int i = al; i < M * N; i += an) {  // obstacle table, I/Obstacle.cpp:41-62 (entry i = m·N + t, as the inputs lie)
      const ObsEntry e = make_obs_entry(kp, a.obs_pose + ((size_t)b * M * N + i) * 4, a.obs_dim + ((size_t)b * M * N + i) * 2);
      double* o = tab + (size_t)i * TABF;
      o[0] = e.ox; o[1] = e.oy; o[2] = e.co; o[3] = e.so; o[4] = e.ia2; o[5] = e.ib2;
    }
    if (wave == 1) {
      const double* pc = a.poly + (size_t)b * CILQR_POLY_COEFFS;
      for (int s = lane; s < S; s += WAVE) {
        double xs;
        sample_xy(grid, pc, s, xs, samp[s]);
      }
      // largest step between adjacent path samples in y: closest_sample's second window (cilqr_device.hpp); to the main wavefront through LDS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the samples: written above by lanes of this wavefront)
      double m = 0.0;
      for (int q = lane; q + 1 < S; q += WAVE) {
        const double d = fabs(samp[q + 1] - samp[q]);
        m = fmax(m, d == d ? d : __builtin_huge_val());
      }
      m = wave_max_uniform(m);
      const double p2 = path_curvature_wave(grid, pc, S, lane, cst);  // (and the coefficients into LDS: closest_sample's Newton search)
      if (lane == 0) { ctl[1] = m; ctl[2] = p2; }
    }
  }
  __syncthreads();  // trajectory, samples and table are in LDS
  grid.dmax = ctl[1];
  grid.p2 = ctl[2];
  grid.pc = cst + CST_PC;

  if (wave != 0) {
    // ---- the aux wavefront: cos / sin, obstacle sums, control barrier, Jacobians — pass after pass -------------------------------------
    unsigned long long busy = 0, calls = 0;
    for (;;) {
      if (*reinterpret_cast<volatile int*>(cmd) == CMD_EXIT) break;
      unsigned long long t0 = 0;
      if (DIAG) t0 = __builtin_readcyclecounter();
      const KParams kpl = phase_params();
      // one step of this wavefront's share: (cA, sA) = cos / sin of the step's heading, (cn, sn) of the next one
      auto aux_step = [&](int t, double cA, double sA, double cn, double sn) {
        const double* xr = Xa + t * XR;
        const double u0 = Ua[2 * t], u1 = Ua[2 * t + 1];
        StepSums s5{0.0, 0.0, 0.0, 0.0, 0.0};
        const ObsConsts oc = make_obs_consts(kpl, xr[0], xr[1], cA, sA);
        if (W == 2 || (UNC && wave == 1)) {  // every entry: even and odd chains, as lin_step forms them
          obstacle_loop<true, false, true>(oc, M, TabSource<false>{tab, wts, N, kpl.w_obstacle}.at(t), s5);
        } else if (!UNC) {                   // this wavefront's chain: entries wave - 1, wave + 1, …
          const int first = wave - 1;
          obstacle_loop<true, false, true, false>(oc, (M - first + 1) / 2, TabObstaclesStrided{tab + (size_t)t * TABF, wts, N, first, 2, kpl.w_obstacle}, s5);
        }
        double* q = part + ((size_t)(wave - 1) * N + t) * 5;
        q[0] = s5.lx0; q[1] = s5.lx1; q[2] = s5.h00; q[3] = s5.h01; q[4] = s5.h11;
        if (UNC && wave == W - 1) {  // the map term, from zero: w·vx, w·mx exactly as unc_cost_add would add them to a finished record
          double g0 = 0.0, g1 = 0.0, h00 = 0.0, h01 = 0.0, h11 = 0.0;
          unc_cost_add(phase_args().unc, upose, b, xr[0], xr[1], cA, sA, g0, g1, h00, h01, h11);
          double* qu = umap + (size_t)t * 5;
          qu[0] = g0; qu[1] = g1; qu[2] = h00; qu[3] = h01; qu[4] = h11;
        }
        if (wave == W - 1) {  // the last aux wavefront: Jacobians
          Rec c;
          ab_terms(kpl, u0, Xa[(t + 1) * XR + 2], cn, sn, c);
          double* r = rec + t * RECF;
          r[10] = c.al; r[11] = c.be; r[12] = c.ga; r[13] = c.de;
        }
        if (W == 3 && wave == (UNC ? 1 : 2)) {  // the control barrier: with W = 2 on main; with a map set beside the obstacles (the map term is the longer share)
          Rec c;
          double a1, a2, a3, a4;
          ctrl_args(kpl, u0, u1, xr[2], a1, a2, a3, a4);
          const double e1 = exp_fast(a1);
          const double e2 = exp_fast(a2);
          const double e3 = exp_fast(a3);
          const double e4 = exp_fast(a4);
          ctrl_terms(kpl, u0, u1, e1, e2, e3, e4, c);
          double* r = rec + t * RECF;
          const double ih = 2.0 / kpl.dt;  // production records: acceleration in units of (dt/2)·u0 (riccati_mfma)
          r[6] = c.lu0 * ih; r[7] = c.lu1; r[8] = c.luu0 * (ih * ih); r[9] = c.luu1;
        }
      };
      double cn = 0.0;
      if (!LONG) {  // one state per lane: a step's next heading comes from the lane above
        double sA = 0.0, cA = 1.0;
        if (lane <= N) sincos_loop(Xa[lane * XR + 3], sA, cA);  // (headings within sincos_loop's range: rollout_fast, MAX_TURN)
        cn = __shfl_down(cA, 1, WAVE);
        double sn = __shfl_down(sA, 1, WAVE);
        if (N == WAVE && lane == WAVE - 1 && wave == W - 1) sincos_loop(Xa[N * XR + 3], sn, cn);  // (N = 64: the last state has no lane of its own)
        if (lane < N) aux_step(lane, cA, sA, cn, sn);
      } else {         // horizons up to 127, two steps per lane: the last aux wavefront evaluates the next heading itself
        for (int t = lane; t < N; t += WAVE) {
          double sA, cA, sn = 0.0;
          sincos_loop(Xa[t * XR + 3], sA, cA);
          cn = 1.0;
          if (wave == W - 1) sincos_loop(Xa[(t + 1) * XR + 3], sn, cn);
          aux_step(t, cA, sA, cn, sn);
        }
      }
      if (DIAG) { busy += stamp_after(cn) - t0; ++calls; }
      __syncthreads();  // A: sums and record slots are in LDS
      __syncthreads();  // B: main has decided, and, going on, has written the next trajectory
    }
    if (DIAG && lane == 0 && a.diag && wave == 1) {
      unsigned long long* o = a.diag + (size_t)b * DIAG_SLOTS;
      o[8] = busy; o[9] = calls;
    }
    // where this wavefront ran: HW_ID (wave slot, SIMD, CU, shader array, shader engine) | XCC_ID << 32 (tools/wave_placement.py)
    if (DIAG && lane == 0 && a.diag)
      a.diag[(size_t)b * DIAG_SLOTS + 12 + wave] = (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) |
                                                    ((unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) << 32);
    if (DIAG && lane == 0 && a.diag && wave == 2) a.diag[(size_t)b * DIAG_SLOTS + 11] = busy;
    return;
  }

  // ---- the main wavefront ----------------------------------------------------------------------------------------------------------
  CILQR_STAMP(c_pro)
  double J_old = DBL_MAX, lamb = 1.0, J_new = 0.0;
  int iters = 0, status = CILQR_EXIT_MAX_ITER, n_pass = 0;
  bool j_valid = false;
  const int max_it = kp.max_iterations;
  for (int it = 0; it < max_it && !handover; ++it) {
    ++iters;
    {  // phase L: forward-pass rows, closest sample, tracking terms, J; then the aux wavefront's sums on top
      const KParams kpl = phase_params();
      double dx[STEPS], dy[STEPS], lx2[STEPS], Jt = 0.0;  // (LONG: steps lane and lane + 64)
#pragma unroll
      for (int c2 = 0; c2 < STEPS; ++c2) {
        dx[c2] = 0.0; dy[c2] = 0.0; lx2[c2] = 0.0;
        const int t = lane + c2 * WAVE;
        if (t < N) {
          const double* xr = Xa + t * XR;
          const double px = xr[0], py = xr[1], v = xr[2];
          const double u0 = Ua[2 * t], u1 = Ua[2 * t + 1];
          double2* q = reinterpret_cast<double2*>(fwd + t * FREC + 10);  // the forward pass reads the old state and control through the scalar path
          q[0] = make_double2(px, py);
          q[1] = make_double2(v, xr[3]);
          q[2] = make_double2(u0, u1);
          const int cs = closest_sample(S, grid, px, py, LdsSamples{samp, grid.xf, grid.dxs});
          dx[c2] = px - fma(grid.dxs, (double)cs, grid.xf);
          dy[c2] = py - samp[cs];
          const double dv = v - kpl.desired_speed;
          lx2[c2] = (2 * kpl.w_vel) * dv;
          Jt += stage_cost(kpl, dx[c2], dy[c2], dv, u0, u1);
          if (W == 2) {  // control barrier (I/Constraints.cpp:110-131); with three wavefronts: on the last one
            Rec c;
            double a1, a2, a3, a4;
            ctrl_args(kpl, u0, u1, v, a1, a2, a3, a4);
            const double e1 = exp_fast(a1);
            const double e2 = exp_fast(a2);
            const double e3 = exp_fast(a3);
            const double e4 = exp_fast(a4);
            ctrl_terms(kpl, u0, u1, e1, e2, e3, e4, c);
            double* r = rec + t * RECF;
            const double ih = 2.0 / kpl.dt;  // production records: acceleration in units of (dt/2)·u0 (riccati_mfma)
            r[6] = c.lu0 * ih; r[7] = c.lu1; r[8] = c.luu0 * (ih * ih); r[9] = c.luu1;
          }
        }
      }
      J_new = wave_sum_uniform(Jt);
      unsigned long long w0 = 0;
      if (DIAG) w0 = stamp_after(J_new);
      __syncthreads();  // A
      if (DIAG) c_wait += __builtin_readcyclecounter() - w0;
#pragma unroll
      for (int c2 = 0; c2 < STEPS; ++c2) {
        const int t = lane + c2 * WAVE;
        if (t < N) {
          const double* q = part + (size_t)t * 5;
          StepSums s5{q[0], q[1], q[2], q[3], q[4]};
          if (W == 3 && !UNC) {  // odd sums onto even sums: obstacle_loop's last statement
            const double* qo = q + (size_t)N * 5;
            s5.lx0 += qo[0]; s5.lx1 += qo[1]; s5.h00 += qo[2]; s5.h01 += qo[3]; s5.h11 += qo[4];
          }
          double* r = rec + t * RECF;
          double lx0, lx1, l00, l01, l11;
          state_terms(kpl, dx[c2], dy[c2], s5, lx0, lx1, l00, l01, l11);
          if (UNC) {  // the map term joins the finished sums (I/Constraints.cpp:188-201)
            const double* qu = umap + (size_t)t * 5;
            lx0 += qu[0]; lx1 += qu[1]; l00 += qu[2]; l01 += qu[3]; l11 += qu[4];
          }
          r[0] = lx0; r[1] = lx1; r[2] = lx2[c2]; r[3] = l00; r[4] = l01; r[5] = l11;
        }
      }
      j_valid = true;
    }
    CILQR_STAMP(c_L)
    if (DIAG) ++n_L;
    const bool accept = J_new < J_old;
    bool stop = false;
    if (!accept) {
      if (J_new != J_new) {
        status = CILQR_EXIT_NUMERIC;
      } else {
        for (;;) {
          lamb = lamb * kp.lamb_factor;
          if (lamb > kp.lamb_max) { status = CILQR_EXIT_LAMBDA_MAX; break; }
          if (++it >= max_it) { status = CILQR_EXIT_MAX_ITER; break; }
          ++iters;
        }
      }
      stop = true;
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wavefront's record slots are in LDS (the aux wavefront's: barrier A)
      if (!riccati_mfma<true, false>(N, rec, rec, fwd, cst, 2.0 / kp.dt, lamb, nullptr)) {
        handover = true;
        stop = true;
      } else {
        CILQR_STAMP(c_R)
        ++n_pass;
        if (DIAG) ++n_R;
        if (!forward_smem(KParams(phase_params()), N, Xa, fwd, Xa, Ua)) {
          handover = true;
          stop = true;
        } else {
          j_valid = false;
          lamb = lamb / kp.lamb_factor;
          if (fabs(J_new - J_old) < kp.tolerance) { status = CILQR_EXIT_TOLERANCE; stop = true; }
          J_old = J_new;
          if (it + 1 >= max_it) stop = true;
        }
      }
    }
    if (stop && lane == 0) *reinterpret_cast<volatile int*>(cmd) = CMD_EXIT;
    __syncthreads();  // B
    CILQR_STAMP(c_F)
    if (stop) break;
  }

  int32_t* const hint = phase_args().hint_passes;
  if (lane == 0) {
    phase_args().redo[b] = handover ? 1 : 0;
    if (handover && hint) hint[b] = 63;
  }
  if (handover) return;  // outputs (and the in/out U) untouched: the GENERAL kernel starts from the same inputs

  // ---- epilogue: X_result / U_result (:243-244) ------------------------------------------------------------------------------
  const SolveArgs ae = phase_args();
  double* Uo = ae.U + (size_t)b * 2 * N;
  for (int i = lane; i < 2 * N; i += WAVE) Uo[i] = Ua[i];
  double* Xg = ae.X_out + (size_t)b * 4 * (N + 1);
  for (int i = lane; i < 4 * (N + 1); i += WAVE) Xg[i] = Xa[(i >> 2) * XR + (i & 3)];
  if (ae.J_out) {
    if (!j_valid) J_new = wave_sum_uniform(cost_only(ae.kp, N, lane, samp, S, grid, Xa, Ua));
    if (lane == 0) ae.J_out[b] = J_new;
  }
  if (lane == 0) {
    if (ae.iters_out) ae.iters_out[b] = iters;
    if (ae.status_out) ae.status_out[b] = status;
    if (ae.passes) ae.passes[b] = n_pass;
    if (ae.hint_passes) ae.hint_passes[b] = n_pass;
  }
  if (DIAG && lane == 0 && a.diag) {
    const unsigned long long now_ = __builtin_readcyclecounter();
    unsigned long long* o = a.diag + (size_t)b * DIAG_SLOTS;
    o[0] = c_pro; o[1] = c_L; o[2] = c_R; o[3] = c_F; o[4] = now_ - tk; o[5] = n_L; o[6] = n_R; o[7] = now_ - tk0;
    o[10] = c_wait;
    o[12] = (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) |
            ((unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) << 32);
    if (W == 2) { o[11] = 0; o[14] = 0; }
    o[15] = 0;
  }
#undef CILQR_STAMP
}

__global__ void quu_inverse_kernel(int n, const double* q, const double* lamb, double* out, int general) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double a = q[4 * i], b = 0.5 * (q[4 * i + 1] + q[4 * i + 2]), d = q[4 * i + 3];
  double i00 = 0, i01 = 0, i11 = 0;
  bool ok = true;
  if (general) ok = quu_inverse_general(a, b, d, lamb[i], i00, i01, i11);
  else quu_inverse_psd(a, b, d, lamb[i], b * b, i00, i01, i11);
  const double nan = __builtin_nan("");
  out[4 * i] = ok ? i00 : nan; out[4 * i + 1] = ok ? i01 : nan; out[4 * i + 2] = ok ? i01 : nan; out[4 * i + 3] = ok ? i11 : nan;
}

// Test hook: the closest-sample search on n independent queries (one lane each), beside the plain scan over ALL samples.
// in[i] = {poly[6], x_first, x_last, px, py}; out[i] = {index by closest_sample, index by the full scan, 1 if the Newton search decided}.
__global__ void closest_sample_kernel(int n, int S, const double* in, int32_t* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* q = in + (size_t)i * 10;
  double pc[CILQR_POLY_COEFFS];
  for (int j = 0; j < CILQR_POLY_COEFFS; ++j) pc[j] = q[j];
  SampleGrid g;
  make_sample_grid(g, q[6], q[7], S);
  auto at = [&](int s, double& x, double& y) { sample_xy(g, pc, s, x, y); };
  double m = 0.0, A = 0.0, B = 0.0, yp = 0.0;
  for (int s = 0; s < S; ++s) {  // what a solve kernel's prologue forms once per solve
    double x, y, a2, a3;
    at(s, x, y);
    if (s > 0) { const double d = fabs(y - yp); m = fmax(m, d == d ? d : __builtin_huge_val()); }
    yp = y;
    path_curvature_terms(pc, x, a2, a3);
    A = fmax(A, a2 == a2 ? a2 : __builtin_huge_val());
    B = fmax(B, a3 == a3 ? a3 : __builtin_huge_val());
  }
  g.dmax = m;
  g.p2 = path_curvature_bound(g, pc, S, A, B);
  g.pc = pc;
  const double px = q[8], py = q[9];
  XWindow w = closest_window_x(S, g, px, py, at);
  closest_window_y(S, g, px, py, w);
  const int by_newton = w.hi - w.lo >= NEWTON_MIN_WINDOW ? closest_newton(S, g, px, py, at, w) : -1;
  out[3 * (size_t)i] = closest_sample(S, g, px, py, at);
  out[3 * (size_t)i + 1] = closest_scan(0, S - 1, px, py, at);
  out[3 * (size_t)i + 2] = by_newton >= 0 ? 1 : 0;
}

__global__ void unc_cost_kernel(UncArgs u, int n, const double* states, double* cost, double* vx, double* mx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const UncPose po = unc_pose(u, 0);
  const double* st = states + 4 * (size_t)i;
  double sn, cs;
  sincos_fast(st[3], &sn, &cs);
  double g0 = 0.0, g1 = 0.0, h00 = 0.0, h01 = 0.0, h11 = 0.0;
  const double c = unc_cost_add(u, po, 0, st[0], st[1], cs, sn, g0, g1, h00, h01, h11);
  const double inv = u.scale != 0.0 ? 1.0 / (u.scale * (double)(u.nl * u.nw)) : 0.0;  // undo w_uncertainty: report the bare cost terms
  cost[i] = c;
  vx[2 * i] = g0 * inv; vx[2 * i + 1] = g1 * inv;
  mx[3 * i] = h00 * inv; mx[3 * i + 1] = h01 * inv; mx[3 * i + 2] = h11 * inv;
}

// Bytes of the per-solve arrays: `compact` = the production kernel without CILQR_FLAG_FAITHFUL_ITERS (forward pass in place,
// no candidate buffers); otherwise with candidate buffers.
size_t core_lds_bytes(int N, int n_samples, bool compact) {
  const size_t traj = (size_t)(N + 1) * XR + (size_t)2 * N;
  const size_t doubles = (((size_t)n_samples + 1) & ~(size_t)1) + (compact ? 1 : 2) * traj + (size_t)N * (compact ? RECF : REC) + RCST;  // gains overlay the records
  return doubles * sizeof(double);
}

// `extra`: bytes behind the per-solve arrays (obstacle table or sample records), the same for both kernels of the pair.
template <bool DIAG, int TAB, bool UNC>
hipError_t launch_pair_unc(const SolveArgs& a, size_t extra, hipStream_t stream) {
  const bool faithful = (a.flags & CILQR_FLAG_FAITHFUL_ITERS) != 0;
  const size_t lds_fast = core_lds_bytes(a.N, a.kp.n_samples, !faithful) + extra;
  const size_t lds_general = core_lds_bytes(a.N, a.kp.n_samples, false) + extra;
  if (lds_general > 64 * 1024) {  // long horizons: opt in to more than the default 64 KiB of dynamic LDS (the CU has 160 KiB)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cilqr_solve_kernel<DIAG, TAB, false, UNC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_general);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cilqr_solve_kernel<DIAG, TAB, true, UNC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_general);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((cilqr_solve_kernel<DIAG, TAB, false, UNC>), dim3(a.B), dim3(WAVE), lds_fast, stream, a);
  hipLaunchKernelGGL((cilqr_solve_kernel<DIAG, TAB, true, UNC>), dim3(a.B), dim3(WAVE), lds_general, stream, a);
  return hipGetLastError();
}
// The two-wavefront kernel (table in LDS, no map, early exit) with the GENERAL kernel of the one-wavefront family behind it.
template <bool DIAG>
hipError_t launch_two_wavefronts(const SolveArgs& a, size_t tab_bytes, hipStream_t stream) {
  const size_t lds_fast = core_lds_bytes(a.N, a.kp.n_samples, true) + tab_bytes + ((((size_t)a.N + 1) & ~(size_t)1) + PAIR_CTL) * sizeof(double);
  const size_t lds_general = core_lds_bytes(a.N, a.kp.n_samples, false) + tab_bytes;
  hipLaunchKernelGGL((cilqr_solve_pair_kernel<DIAG>), dim3(a.B), dim3(2 * WAVE), lds_fast, stream, a);
  hipLaunchKernelGGL((cilqr_solve_kernel<DIAG, 1, true, false>), dim3(a.B), dim3(WAVE), lds_general, stream, a);
  return hipGetLastError();
}
// The shared-phase-L kernel (table in LDS, early exit, N ≤ 127; with or without an uncertainty map) with the GENERAL kernel of the
// one-wavefront family behind it.
template <int W, bool LONG, bool DIAG, bool UNC>
hipError_t launch_shared_L(const SolveArgs& a, size_t tab_bytes, hipStream_t stream) {
  const size_t lds_fast = core_lds_bytes(a.N, a.kp.n_samples, true) + tab_bytes +
                          ((((size_t)((W - 1) + (UNC ? 1 : 0)) * 5 * a.N + 1) & ~(size_t)1) + 4) * sizeof(double);
  const size_t lds_general = core_lds_bytes(a.N, a.kp.n_samples, false) + tab_bytes;
  if (lds_fast > 64 * 1024 || lds_general > 64 * 1024) {  // few solves per CU with large tables: more than the default 64 KiB of dynamic LDS
    const int want = (int)(lds_fast > lds_general ? lds_fast : lds_general);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cilqr_solve_share_kernel<W, LONG, DIAG, UNC>), hipFuncAttributeMaxDynamicSharedMemorySize, want);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cilqr_solve_kernel<DIAG, 1, true, UNC>), hipFuncAttributeMaxDynamicSharedMemorySize, want);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((cilqr_solve_share_kernel<W, LONG, DIAG, UNC>), dim3(a.B), dim3(W * WAVE), lds_fast, stream, a);
  hipLaunchKernelGGL((cilqr_solve_kernel<DIAG, 1, true, UNC>), dim3(a.B), dim3(WAVE), lds_general, stream, a);
  return hipGetLastError();
}
template <bool DIAG, bool UNC>
hipError_t launch_shared_L_by_shape(const SolveArgs& a, size_t tab_bytes, hipStream_t stream) {
  if (a.N > WAVE) return launch_shared_L<2, true, DIAG, UNC>(a, tab_bytes, stream);
  if (a.pair == 3 && (UNC || a.M >= 2)) return launch_shared_L<3, false, DIAG, UNC>(a, tab_bytes, stream);
  return launch_shared_L<2, false, DIAG, UNC>(a, tab_bytes, stream);
}
template <bool DIAG, int TAB>
hipError_t launch_pair(const SolveArgs& a, size_t extra, hipStream_t stream) {
  return a.unc.layer ? launch_pair_unc<DIAG, TAB, true>(a, extra, stream) : launch_pair_unc<DIAG, TAB, false>(a, extra, stream);
}

}  // namespace

size_t solve_lds_bytes(int N, int n_samples) { return core_lds_bytes(N, n_samples, false); }  // the larger of the two layouts
// Static obstacles: the table stays in LDS while a workgroup stays within 32 KiB (≥ 5 solves resident per CU of 160 KiB).  (M = 0: an empty table fits)
// `budget` (bytes; 0: those 32 KiB): what a workgroup may take where fewer solves than that share a CU — a batch of at most four solves
// per CU leaves each of them a quarter of the CU's LDS, and the table in LDS is worth more than the residency nobody uses
// (cilqr_api.cpp, lds_table_budget).
bool solve_table_in_lds(int N, int M, int n_samples, int budget) {
  return solve_lds_bytes(N, n_samples) + (size_t)M * TABF * N * sizeof(double) <= (size_t)(budget > 0 ? budget : 32 * 1024);
}
// … and whether a solve of this shape can take the shared-phase-L kernel (cilqr_solve_share_kernel): table in LDS, at most two steps per lane
bool solve_share_applies(int N, int M, int n_samples, int budget) { return N < 2 * WAVE && solve_table_in_lds(N, M, n_samples, budget); }

size_t solve_sampled_lds_bytes(int n_obs, int n_samples) {
  return ((size_t)n_obs * n_samples * OFFF + (size_t)2 * n_obs) * sizeof(double);  // offset records + rmax + constant-shape flags
}
size_t solve_sampled_tab_doubles(int n_obs, int N) { return (size_t)n_obs * NOMF * N; }

// Dispatch order for the next call: solve indices sorted by pass count, descending (counting sort, one workgroup; ties in any
// order — every permutation gives the same results, the order only decides which solves start first).
__global__ __launch_bounds__(1024) void schedule_order_kernel(const int32_t* passes, int B, int32_t* order) {
  __shared__ int cnt[64], base[64];
  const int tid = threadIdx.x;
  if (tid < 64) cnt[tid] = 0;
  __syncthreads();
  for (int i = tid; i < B; i += 1024) {
    const int p = passes[i];
    atomicAdd(&cnt[63 - (p < 0 ? 0 : p > 63 ? 63 : p)], 1);  // bucket 0 = the longest
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int k = 0; k < 64; ++k) { base[k] = acc; acc += cnt[k]; }
  }
  __syncthreads();
  for (int i = tid; i < B; i += 1024) {
    const int p = passes[i];
    order[atomicAdd(&base[63 - (p < 0 ? 0 : p > 63 ? 63 : p)], 1)] = i;
  }
}

hipError_t launch_schedule_order(const int32_t* passes, int B, int32_t* order, hipStream_t stream) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(schedule_order_kernel, dim3(1), dim3(1024), 0, stream, passes, B, order);
  return hipGetLastError();
}

hipError_t launch_unc_cost(const UncArgs& u, int n, const double* states, double* cost, double* vx, double* mx, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(unc_cost_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, u, n, states, cost, vx, mx);
  return hipGetLastError();
}

hipError_t launch_closest_sample(int n, int S, const double* in, int32_t* out, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(closest_sample_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, n, S, in, out);
  return hipGetLastError();
}

hipError_t launch_quu_inverse(int n, const double* q, const double* lamb, double* out, int general, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(quu_inverse_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, n, q, lamb, out, general);
  return hipGetLastError();
}

hipError_t launch_solve_wave(const SolveArgs& a, hipStream_t stream) {
  if (a.B <= 0) return hipSuccess;
  const size_t lds = solve_lds_bytes(a.N, a.kp.n_samples);  // the larger layout: limits and the table decision hold for both kernels
  const size_t tab_bytes = (size_t)a.M * TABF * a.N * sizeof(double);
  const bool tab_lds = a.n_samples == 0 && solve_table_in_lds(a.N, a.M, a.kp.n_samples, a.tab_budget);
  if (a.n_samples > 0) {  // sampled obstacles: offset records in LDS, nominal records in the global workspace
    const size_t extra = solve_sampled_lds_bytes(a.M, a.n_samples);
    if (lds + extra > SOLVE_LDS_MAX) return hipErrorInvalidValue;  // checked by the caller
    if (a.split >= 2 && a.N <= WAVE && a.M >= a.split && (a.flags & CILQR_FLAG_FAITHFUL_ITERS) == 0) {
      // a.split wavefronts per solve share phase L (cilqr_solve_split_kernel); the GENERAL kernel of the one-wavefront family behind
      const int W = a.split >= 4 ? 4 : 2;
      const bool unc = a.unc.layer != nullptr;
      const size_t lds_fast = core_lds_bytes(a.N, a.kp.n_samples, true) + extra + ((((size_t)((W - 1) + (unc ? 1 : 0)) * 5 * a.N + 1) & ~(size_t)1) + 2) * sizeof(double);
      const size_t lds_general = core_lds_bytes(a.N, a.kp.n_samples, false) + extra;
      if (lds_fast <= 64 * 1024 && lds_general <= 64 * 1024) {
#define CILQR_SPLIT_LAUNCH(WW, DD, UU)                                                                                          \
  hipLaunchKernelGGL((cilqr_solve_split_kernel<WW, DD, UU>), dim3(a.B), dim3(WW * WAVE), lds_fast, stream, a);                 \
  hipLaunchKernelGGL((cilqr_solve_kernel<DD, 2, true, UU>), dim3(a.B), dim3(WAVE), lds_general, stream, a);
        if (a.diag) {
          if (unc) { if (W == 4) { CILQR_SPLIT_LAUNCH(4, true, true) } else { CILQR_SPLIT_LAUNCH(2, true, true) } }
          else { if (W == 4) { CILQR_SPLIT_LAUNCH(4, true, false) } else { CILQR_SPLIT_LAUNCH(2, true, false) } }
        } else {
          if (unc) { if (W == 4) { CILQR_SPLIT_LAUNCH(4, false, true) } else { CILQR_SPLIT_LAUNCH(2, false, true) } }
          else { if (W == 4) { CILQR_SPLIT_LAUNCH(4, false, false) } else { CILQR_SPLIT_LAUNCH(2, false, false) } }
        }
#undef CILQR_SPLIT_LAUNCH
        return hipGetLastError();
      }
    }
    return a.diag ? launch_pair<true, 2>(a, extra, stream) : launch_pair<false, 2>(a, extra, stream);
  }
  if (lds > SOLVE_LDS_MAX) return hipErrorInvalidValue;  // checked by the caller
  const size_t extra = tab_lds ? tab_bytes : 0;
  if (tab_lds && a.pair == 1 && !a.unc.layer && (a.flags & CILQR_FLAG_FAITHFUL_ITERS) == 0)
    return a.diag ? launch_two_wavefronts<true>(a, tab_bytes, stream) : launch_two_wavefronts<false>(a, tab_bytes, stream);
  if (a.pair >= 2 && a.n_samples == 0 && solve_share_applies(a.N, a.M, a.kp.n_samples, a.tab_budget) && (a.flags & CILQR_FLAG_FAITHFUL_ITERS) == 0) {
    if (a.unc.layer) return a.diag ? launch_shared_L_by_shape<true, true>(a, tab_bytes, stream) : launch_shared_L_by_shape<false, true>(a, tab_bytes, stream);
    return a.diag ? launch_shared_L_by_shape<true, false>(a, tab_bytes, stream) : launch_shared_L_by_shape<false, false>(a, tab_bytes, stream);
  }
  if (a.diag) return tab_lds ? launch_pair<true, 1>(a, extra, stream) : launch_pair<true, 0>(a, extra, stream);
  return tab_lds ? launch_pair<false, 1>(a, extra, stream) : launch_pair<false, 0>(a, extra, stream);
}

}  // namespace cilqr
