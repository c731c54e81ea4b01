// cilqr_solve_groups.hip — batched constrained-iLQR solve for gfx950 (MI355X), G LANES PER SOLVE, 64/G solves per
// wavefront.  The kernel family for batches larger than about one solve per SIMD (cilqr_solve.hip holds the
// one-wavefront-per-solve, LDS-resident family; cilqr_solve_batch_device picks).
//
// Why a second mapping.  A wavefront issues one instruction per ≈5 ticks whatever it is.  In the one-wavefront-per-solve
// kernel the sequential phases (backward Riccati recursion R, forward pass F; I/iLQR.cpp:133-191, 68-86) keep 1 useful
// lane in 64, and LDS (≈20-30 KiB per solve) caps residency at 5-8 solves per CU, so a large batch runs in many rounds at
// a few per cent lane use.  Here a wavefront carries S = 64/G solves: R and F run once for S solves (lanes of a group
// compute redundantly, groups differ) and the lane-parallel linearisation L (I/Constraints.cpp:145-227, 86-137) spreads a
// solve's steps over its G lanes.  G ≈ 65536/B (power of two), i.e. about one wavefront per SIMD.
//
// Where the data lives (round 2).  Trajectories X/U (double-buffered) and the gains k, K stay in a global workspace laid out
// [array][step][field][solve-in-wavefront] (a wavefront instruction touches S consecutive doubles per row).  The per-step
// linearisation records — 16 of the 26 doubles per step that used to make a round trip through that workspace every
// iteration — no longer leave the chip: the backward recursion consumes the horizon in chunks from the end, so phase L
// linearises one chunk of steps straight into LDS and phase R runs over that chunk before L produces the next one
// ("L→R hand-over").  The forward pass reads its operands (old X, old U, gains) from LDS, filled one chunk of steps ahead by
// asynchronous direct-to-LDS loads.
//
// Control flow.  The iteration loop is WAVE-UNIFORM: every lane stays in it until the last solve of the wavefront has
// finished; a solve that has finished (or whose index is past the batch) only switches its lanes' `active` flag off, and every
// phase runs under `if (active)`.  Whole-wavefront jobs (the chunk copies, the LDS hand-over barriers) therefore sit at points
// where the wavefront is whole and run with the hardware's own EXEC — nothing forces EXEC or touches registers of lanes the
// compiler believes inactive.
//
// Lane sharing in phase L (round 3).  The solves of a wavefront differ widely in length (config 5: 35 % run all 20 iterations,
// the rest end anywhere from the 2nd on), and a wavefront stays in its loop until its last solve has finished: measured with
// tools/group_idle.py, its lane groups are busy 62 % of the loop at S = 8 (config 5), 41 % at S = 16, 37 % at S = 64.  R and F
// are sequential per solve and cost the same however many solves are left — but phase L, the largest phase, is a set of
// independent (solve, step) items, so the lanes of finished solves take steps of the unfinished ones: with k solves active every
// one of them gets floor(64 / k) lanes instead of G.  What a helper lane needs of a solve (polynomial, sample grid, held-obstacle
// mask, buffer in use) lies in a small per-solve block in LDS; the stage cost of a step goes through LDS to the solve's OWN lanes,
// which add the steps up in the order they always did — the result of a solve does not depend on who shares its wavefront
// (permutation tests), and is bit-identical to the unshared mapping (CILQR_NO_LANE_SHARING at cilqr_create switches it off).
//
// Arithmetic is shared with the LDS family (cilqr_device.hpp): same per-step functions, same samples, same fast/GENERAL
// split with the redo hand-over.  Path samples are recomputed where needed (12 instructions) instead of stored.
#include "cilqr_device.hpp"

namespace cilqr {

using namespace dev;

namespace {

// Per-wavefront workspace block, in rows of S doubles (one column per solve of the wavefront), STEP-major:
// row(field f of an array, step t) = array base + t*(fields of that array) + f.  The serial phases address a step's operands
// with ONE base register and immediate offsets f·S·8.
struct WsLayout {
  int N, M;
  __device__ __host__ int xa() const { return 0; }
  __device__ __host__ int xb() const { return (N + 1) * XR; }
  __device__ __host__ int ua() const { return 2 * (N + 1) * XR; }
  __device__ __host__ int ub() const { return ua() + 2 * N; }
  __device__ __host__ int kk() const { return ub() + 2 * N; }
  __device__ __host__ int rows() const { return kk() + KR * N; }
};

template <int G>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}

__device__ __forceinline__ void mem_sync() {
  // stores of one lane are read by other lanes of the same wavefront through global memory: drain and re-order
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

extern __shared__ __attribute__((aligned(16))) double cilqr_groups_lds[];
constexpr int F_ROWS = XR + 2 + KR;  // forward pass: 6 state + 2 control + 10 gain rows per step
// Per-solve block in LDS (phase L's lane sharing): {poly[6], xf, dxs, inv_dxs, dmax, windowed, held mask, map pose[4], xc | uc}
constexpr int PAR = 18;
constexpr int P_XF = 6, P_DXS = 7, P_INV = 8, P_DMAX = 9, P_WIN = 10, P_HELD = 11, P_POSE = 12, P_BUF = 16, P_P2 = 17;

// n_doubles (even) contiguous doubles from src (global, wave-uniform) to dst (LDS, wave-uniform) by the asynchronous
// direct-to-LDS load: 16 bytes per lane and instruction, no data registers.  Called only where the wavefront is whole (see the
// header): the lanes of the last, partial piece are selected by an ordinary branch.  stage_wait() is the loads' only wait.
__device__ __forceinline__ void stage_copy(const double* src, double* dst, int n_doubles) {
  const int bytes = n_doubles * 8;
  const unsigned lds0 = __builtin_amdgcn_groupstaticsize() + (unsigned)((dst - cilqr_groups_lds) * sizeof(double));
  const unsigned lane16 = threadIdx.x * 16u;
  for (int off = 0; off < bytes; off += WAVE * 16) {
    // (wave-uniform by construction; said so to the compiler, which otherwise may hold it in vector registers — the "s" operand below)
    const unsigned long long piece0 = reinterpret_cast<unsigned long long>(src) + (unsigned long long)off;
    const unsigned long long piece = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(piece0 >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)piece0);
    const unsigned lds_byte = lds0 + (unsigned)off;
    if ((int)lane16 < bytes - off) {
      unsigned m0_save;
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %2\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %1, %3\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(m0_save)
          : "v"(lane16), "s"(lds_byte), "s"(piece)
          : "memory");
    }
  }
}
__device__ __forceinline__ void stage_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- the production kernel ---------------------------------------------------------------------------------------------------
// chunk_r: steps per L→R hand-over chunk (a multiple of G); chunk_f: steps per forward-pass staging chunk (two buffers).
// UNC: an uncertainty map is set — a separate instantiation, so that without a map none of its code is in the kernel.
template <int G, bool UNC>
__global__ __launch_bounds__(WAVE) void cilqr_solve_groups_fast(SolveArgs a, double* ws_base, int chunk_r, int chunk_f, int lds_main) {
  double* lds = cilqr_groups_lds;
  constexpr int S = WAVE / G;
  double* const Jt = lds + lds_main;                  // [chunk_r][S]: stage cost of each step of the chunk in hand
  double* const par = Jt + (size_t)chunk_r * S;       // [S][PAR]: the per-solve blocks
  int* const alist = reinterpret_cast<int*>(par + (size_t)S * PAR);  // [S]: the solves phase L works on in this iteration
  const int lane = threadIdx.x, grp = lane / G, g = lane % G;
  const int b = blockIdx.x * S + grp;
  const bool live = b < a.B;  // lanes past the batch stay in the wavefront, switched off (whole groups)
  const SolveArgs* aq = &phase_args();
  const KParams kp = a.kp;
  const int N = a.N, M = a.M, NS = kp.n_samples;
  const WsLayout L{N, M};

  // column `grp` of this wavefront's block: element (row r) at ws[r * S]
  const double* wave_ws = ws_base + (size_t)blockIdx.x * L.rows() * S;
  double* ws = ws_base + (size_t)blockIdx.x * L.rows() * S + grp;
  // obstacle table of this wavefront: entry (m, t, solve) = 6 contiguous doubles at ((m*N + t)*S + solve)*6
  double* tab = a.obs_tab + (size_t)blockIdx.x * ((size_t)M * N * TABF) * S + (size_t)grp * TABF;
#define XF(base, t, f) ws[(size_t)((base) + (t) * XR + (f)) * S]  /* state arrays: (N+1) steps x 6 fields */
#define UF(base, t, f) ws[(size_t)((base) + (t) * 2 + (f)) * S]   /* control arrays: N steps x 2 fields */
#define KF(t, f) ws[(size_t)(L.kk() + (t) * KR + (f)) * S]        /* gains: N steps x 10 fields */

  // ---- prologue -------------------------------------------------------------------------------------------
  double pc[CILQR_POLY_COEFFS] = {0, 0, 0, 0, 0, 0};
  SampleGrid grid;
  make_sample_grid(grid, 0.0, 1.0, NS);
  unsigned long long held = 0;
  bool handover = false;
  auto sample_at = [&](int s, double& x, double& y) { sample_xy(grid, pc, s, x, y); };
  auto store_state = [&](int base, int t, const State& s) {
    if (g == 0) {
      double* xr = &XF(base, t, 0);
      xr[0] = s.x; xr[S] = s.y; xr[2 * S] = s.v; xr[3 * S] = s.th; xr[4 * S] = s.c; xr[5 * S] = s.s;
    }
  };
  if (live) {
#pragma unroll
    for (int j = 0; j < CILQR_POLY_COEFFS; ++j) pc[j] = a.poly[(size_t)b * CILQR_POLY_COEFFS + j];
    make_sample_grid(grid, a.xplan_fl[2 * b], a.xplan_fl[2 * b + 1], NS);
    {  // largest step between adjacent path samples in y: closest_sample's second window (cilqr_device.hpp)
      double m = 0.0;
      for (int q = g; q + 1 < NS; q += G) {
        double x0_, y0_, x1_, y1_;
        sample_xy(grid, pc, q, x0_, y0_);
        sample_xy(grid, pc, q + 1, x1_, y1_);
        const double d = fabs(y1_ - y0_);
        m = fmax(m, d == d ? d : __builtin_huge_val());
      }
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, WAVE));
      grid.dmax = m;
    }
    {  // an upper bound of the path's second derivative over the samples: closest_sample's Newton search (cilqr_device.hpp)
      double A = 0.0, Bq = 0.0;
      for (int q = g; q < NS; q += G) {
        double a2, a3;
        path_curvature_terms(pc, fma(grid.dxs, (double)q, grid.xf), a2, a3);
        A = fmax(A, a2 == a2 ? a2 : __builtin_huge_val());
        Bq = fmax(Bq, a3 == a3 ? a3 : __builtin_huge_val());
      }
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) {
        A = fmax(A, __shfl_xor(A, o, WAVE));
        Bq = fmax(Bq, __shfl_xor(Bq, o, WAVE));
      }
      grid.p2 = path_curvature_bound(grid, pc, NS, A, Bq);
      grid.pc = pc;
    }
    double* Ug = a.U + (size_t)b * 2 * N;
    for (int t = g; t < N; t += G) {
      UF(L.ua(), t, 0) = Ug[2 * t];
      UF(L.ua(), t, 1) = Ug[2 * t + 1];
    }
    // Obstacle table, I/Obstacle.cpp:41-62.  An obstacle whose pose and dimensions are the same in every column (how the
    // reference node feeds static obstacles: one pose replicated over the horizon, I/ilqr_uncertainty_node.cpp:175-185) gets a
    // bit in `held`: phase L then reads its step-0 row for every step — the same values, but one cache-resident line per
    // field instead of a stream of N·6 doubles per obstacle, solve and iteration from HBM.
    for (int m = 0; m < M; ++m) {
      const double* pose0 = a.obs_pose + ((size_t)b * M + m) * N * 4;
      const double* dim0 = a.obs_dim + ((size_t)b * M + m) * N * 2;
      bool same = true;
      for (int t = g; t < N; t += G) {
        const double* pose = pose0 + (size_t)t * 4;
        const double* dim = dim0 + (size_t)t * 2;
        same = same && pose[0] == pose0[0] && pose[1] == pose0[1] && pose[2] == pose0[2] && pose[3] == pose0[3] &&
               dim[0] == dim0[0] && dim[1] == dim0[1];
      }
      int all = same ? 1 : 0;
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) all &= __shfl_xor(all, o, WAVE);
      const bool is_held = all && m < 64;
      if (is_held) held |= 1ull << m;
      // a held obstacle needs its step-0 row only (lane g == 0 of the group writes it); the others need every row
      for (int t = g; t < (is_held ? 1 : N); t += G) {
        const ObsEntry e = make_obs_entry(kp, pose0 + (size_t)t * 4, dim0 + (size_t)t * 2);
        double* o = tab + (size_t)(m * N + t) * S * TABF;
        o[0] = e.ox; o[1] = e.oy; o[2] = e.co; o[3] = e.so; o[4] = e.ia2; o[5] = e.ib2;
      }
    }
  }
  UncPose upose{0, 0, 1, 0};
  if (UNC && live) upose = unc_pose(a.unc, b);
  if (live && g == 0) {  // the per-solve block of phase L's lane sharing
    double* q = par + (size_t)grp * PAR;
#pragma unroll
    for (int j = 0; j < CILQR_POLY_COEFFS; ++j) q[j] = pc[j];
    q[P_XF] = grid.xf; q[P_DXS] = grid.dxs; q[P_INV] = grid.inv_dxs; q[P_DMAX] = grid.dmax; q[P_WIN] = grid.windowed ? 1.0 : 0.0; q[P_P2] = grid.p2;
    reinterpret_cast<unsigned long long*>(q)[P_HELD] = held;
    q[P_POSE] = upose.px; q[P_POSE + 1] = upose.py; q[P_POSE + 2] = upose.cp; q[P_POSE + 3] = upose.sp;
    reinterpret_cast<int*>(q + P_BUF)[0] = L.xa(); reinterpret_cast<int*>(q + P_BUF)[1] = L.ua();
  }
  mem_sync();
  if (live) {  // nominal rollout, I/iLQR.cpp:51-62
    const double* x0 = a.x0 + (size_t)b * 4;
    State s;
    s.x = x0[0]; s.y = x0[1]; s.v = x0[2]; s.th = x0[3];
    FwdConst k;
    make_fwd_const(k, kp);
    const bool th0_ok = fabs(s.th) < MAX_HEADING0;
    double max_turn = 0.0;
    sincos_loop(s.th, s.s, s.c);
    store_state(L.xa(), 0, s);
    for (int i = 0; i < N; ++i) {
      dyn_step_loop(k, s, UF(L.ua(), i, 0), UF(L.ua(), i, 1), max_turn);
      store_state(L.xa(), i + 1, s);
    }
    handover = !(th0_ok && max_turn <= MAX_TURN) || (a.flags & CILQR_FLAG_GENERAL_ONLY) != 0;
  }
  mem_sync();

  // ---- iteration loop, I/iLQR.cpp:204-239: wave-uniform, every solve under its own `active` flag ------------------------------
  bool active = live && !handover;
  int xc = L.xa(), xn = L.xb(), uc = L.ua(), un = L.ub();  // this group's current / candidate trajectory buffers
  double J_old = DBL_MAX, lamb = 1.0, J_new = 0.0;
  int iters = 0, status = CILQR_EXIT_MAX_ITER, n_pass = 0;
  bool j_valid = false;
  const bool faithful = (a.flags & CILQR_FLAG_FAITHFUL_ITERS) != 0;
  const int max_it = kp.max_iterations;
  const double dt = kp.dt, two_wvel = kp.w_vel * 2;

  unsigned long long tL = 0, tR = 0, tF = 0, nL = 0, nR = 0;
  const unsigned long long t_begin = __builtin_readcyclecounter();
  for (int it = 0; it < max_it; ++it) {
    if (__builtin_amdgcn_ballot_w64(active) == 0) break;
    if (active) ++iters;
    // ---- who linearises what in this iteration: the k active solves, floor(64 / k) lanes each (lanes j, j + k, … work for the
    // j-th active solve, so that neighbouring lanes touch neighbouring columns of a workspace row).  Without lane sharing: every
    // solve of the wavefront is listed and keeps its own G lanes; the lanes of a finished one find it inactive and idle.
    int k_act, lanes_per, my_slot = -1, my_sub = 0;
    {
      const unsigned long long lead = __builtin_amdgcn_ballot_w64(active && g == 0);
      if (a.steal) {
        k_act = __builtin_popcountll(lead);
        if (active && g == 0) alist[__builtin_popcountll(lead & ((1ull << lane) - 1ull))] = grp;
      } else {
        k_act = S;
        if (g == 0) alist[grp] = active ? grp : -1;
      }
      __syncthreads();  // (one wavefront per workgroup: orders the LDS traffic, waits for nobody)
      lanes_per = WAVE / (k_act > 0 ? k_act : 1);
      if (k_act > 0 && lane < k_act * lanes_per) {
        my_slot = alist[lane % k_act];
        my_sub = lane / k_act;
      }
    }
    // ---- phases L and R, fused: chunks of steps from the end of the horizon; the records of a chunk go L → LDS → R
    double Jpart = 0.0;
    bool r_ok = true;
    Value V;
    value_terminal(V, Rec{}, 0.0);
    for (int hi = N - 1; hi >= 0;) {
      const int lo = hi - chunk_r + 1 < 0 ? 0 : hi - chunk_r + 1;
      unsigned long long t0 = a.diag ? __builtin_readcyclecounter() : 0;
      if (my_slot >= 0) {
        // this lane's steps lo + my_sub, lo + my_sub + lanes_per, … ≤ hi of solve my_slot: the ten operands of the next one are
        // requested while the current computes
        const double* q = par + (size_t)my_slot * PAR;
        double pcs[CILQR_POLY_COEFFS];
#pragma unroll
        for (int j = 0; j < CILQR_POLY_COEFFS; ++j) pcs[j] = q[j];
        SampleGrid gs;
        gs.xf = q[P_XF]; gs.dxs = q[P_DXS]; gs.inv_dxs = q[P_INV]; gs.dmax = q[P_DMAX]; gs.windowed = q[P_WIN] != 0.0;
        gs.p2 = q[P_P2]; gs.pc = pcs;
        const unsigned long long held_s = reinterpret_cast<const unsigned long long*>(q)[P_HELD];
        const int xcs = reinterpret_cast<const int*>(q + P_BUF)[0], ucs = reinterpret_cast<const int*>(q + P_BUF)[1];
        const int bs = blockIdx.x * S + my_slot;
        const double* wss = wave_ws + my_slot;  // column my_slot of this wavefront's workspace block
        const double* tabs = a.obs_tab + (size_t)blockIdx.x * ((size_t)M * N * TABF) * S + (size_t)my_slot * TABF;
        const double* wtss = a.obs_weight ? a.obs_weight + (size_t)bs * M : nullptr;
        auto sample_s = [&](int s_, double& x, double& y) { sample_xy(gs, pcs, s_, x, y); };
        struct LIn { double px, py, v, ct, st, u0, u1, vn, cn, sn; };
        auto load_in = [&](LIn& o, int t) {
          const double* xr = wss + (size_t)(xcs + t * XR) * S;
          const double* xq = wss + (size_t)(xcs + (t + 1) * XR) * S;
          const double* ur = wss + (size_t)(ucs + t * 2) * S;
          o.px = xr[0]; o.py = xr[S]; o.v = xr[2 * S]; o.ct = xr[4 * S]; o.st = xr[5 * S];
          o.u0 = ur[0]; o.u1 = ur[S];
          o.vn = xq[2 * S]; o.cn = xq[4 * S]; o.sn = xq[5 * S];
        };
        const KParams kpl = phase_params();  // this phase's own read of the parameter block (cilqr_device.hpp)
        const UncArgs* unc = UNC ? &aq->unc : nullptr;
        LIn cur, nxt;
        if (lo + my_sub <= hi) load_in(cur, lo + my_sub);
        for (int t = lo + my_sub; t <= hi; t += lanes_per) {
          if (t + lanes_per <= hi) load_in(nxt, t + lanes_per);
          const int cs = closest_sample(NS, gs, cur.px, cur.py, sample_s);
          double cx, cy;
          sample_xy(gs, pcs, cs, cx, cy);
          // one 48-byte entry = three 16-byte loads from one address; a held obstacle reads its step-0 entry
          auto obs = [&](int m, ObsEntry& e, double& w) {
            const int row = (m < 64 && ((held_s >> m) & 1)) ? 0 : t;
            const double2* p = reinterpret_cast<const double2*>(tabs + ((size_t)m * N + row) * S * TABF);
            const double2 q0 = p[0], q1 = p[1], q2 = p[2];
            e.ox = q0.x; e.oy = q0.y; e.co = q1.x; e.so = q1.y; e.ia2 = q2.x; e.ib2 = q2.y;
            w = wtss ? wtss[m] : kpl.w_obstacle;
            return true;
          };
          Rec c;
          Jt[(size_t)(t - lo) * S + my_slot] =
              lin_step(kpl, cur.px, cur.py, cur.v, cur.ct, cur.st, cur.u0, cur.u1, cur.vn, cur.cn, cur.sn, cx, cy, M, obs, c);
          double* r = lds + (size_t)(t - lo) * REC * S + my_slot;  // the record of step t, column my_slot of the chunk in LDS
          r[0] = c.lx0; r[S] = c.lx1; r[2 * S] = c.lx2; r[3 * S] = c.l00; r[4 * S] = c.l01; r[5 * S] = c.l11;
          r[6 * S] = c.lu0; r[7 * S] = c.lu1; r[8 * S] = c.luu0; r[9 * S] = c.luu1;
          r[10 * S] = c.al; r[11 * S] = c.be; r[12 * S] = c.ga; r[13 * S] = c.de; r[14 * S] = c.p; r[15 * S] = c.q;
          cur = nxt;
        }
        if (unc) {  // the map's term joins l_x, l_xx after the obstacles' (I/Constraints.cpp:188-201), in a loop of its own
          const UncPose ups{q[P_POSE], q[P_POSE + 1], q[P_POSE + 2], q[P_POSE + 3]};
          for (int t = lo + my_sub; t <= hi; t += lanes_per) {
            const double* xr = wss + (size_t)(xcs + t * XR) * S;
            double* r = lds + (size_t)(t - lo) * REC * S + my_slot;
            double lx0 = r[0], lx1 = r[S], h00 = r[3 * S], h01 = r[4 * S], h11 = r[5 * S];
            unc_cost_add(*unc, ups, bs, xr[0], xr[S], xr[4 * S], xr[5 * S], lx0, lx1, h00, h01, h11);
            r[0] = lx0; r[S] = lx1; r[3 * S] = h00; r[4 * S] = h01; r[5 * S] = h11;
          }
        }
      }
      __syncthreads();  // the chunk's records are in LDS (one wavefront per workgroup: this orders LDS traffic, it waits for nobody)
      if (a.diag && active) { const unsigned long long t1 = __builtin_readcyclecounter(); tL += t1 - t0; t0 = t1; }
      if (active) {
        // the stage costs of this chunk's steps, summed by the solve's own lanes in the order of the unshared mapping
        for (int t = lo + g; t <= hi; t += G) Jpart += Jt[(size_t)(t - lo) * S + grp];
        auto lds_rec = [&](Rec& o, int t_rel) {
          const double* r = lds + (size_t)t_rel * REC * S + grp;
          o.lx0 = r[0]; o.lx1 = r[S]; o.lx2 = r[2 * S]; o.l00 = r[3 * S]; o.l01 = r[4 * S]; o.l11 = r[5 * S];
          o.lu0 = r[6 * S]; o.lu1 = r[7 * S]; o.luu0 = r[8 * S]; o.luu1 = r[9 * S];
          o.al = r[10 * S]; o.be = r[11 * S]; o.ga = r[12 * S]; o.de = r[13 * S]; o.p = r[14 * S]; o.q = r[15 * S];
        };
        Gains gn;
        bool ok;
        auto step = [&](const Rec& c, int j) {
          riccati_step<true>(c, V, dt, two_wvel, lamb, gn, ok);
          r_ok = r_ok && ok;
          if (g == 0) {
            double* kp_ = &KF(j, 0);
#pragma unroll
            for (int i = 0; i < KR; ++i) kp_[i * S] = gn.g[i];
          }
        };
        Rec ra, rb;
        int j = hi;
        lds_rec(ra, j - lo);
        if (j == N - 1) value_terminal(V, ra, two_wvel);  // I/iLQR.cpp:108-113
        for (; j >= lo + 1; j -= 2) {  // two steps per trip, the next record's LDS reads under this step's arithmetic
          lds_rec(rb, j - 1 - lo);
          step(ra, j);
          lds_rec(ra, j - 2 >= lo ? j - 2 - lo : 0);
          step(rb, j - 1);
        }
        if (j == lo) step(ra, lo);
      }
      __syncthreads();  // the chunk has been consumed: L may overwrite it
      if (a.diag && active) tR += __builtin_readcyclecounter() - t0;
      hi = lo - 1;
    }
    if (a.diag && active) { ++nL; ++nR; }

    bool accept = false;
    if (active) {
      J_new = group_sum<G>(Jpart);
      j_valid = true;
      accept = J_new < J_old;
      if (!accept && !faithful) {
        // A rejection leaves X, U untouched, so every later iteration of the reference loop recomputes the same J_new and
        // rejects again until lamb > lamb_max or the iteration cap: only lamb and the counter change.  The
        // backward pass just run belongs to the rejected iteration and is discarded (the reference discards its gains too).
        if (J_new != J_new) {
          status = CILQR_EXIT_NUMERIC;
        } else {
          int it2 = it;
          for (;;) {
            lamb = lamb * kp.lamb_factor;
            if (lamb > kp.lamb_max) { status = CILQR_EXIT_LAMBDA_MAX; break; }
            if (++it2 >= max_it) { status = CILQR_EXIT_MAX_ITER; break; }
            ++iters;
          }
        }
        active = false;
      } else if (!r_ok) {
        handover = true;
        active = false;
      } else {
        ++n_pass;
      }
    }
    mem_sync();  // the gains (stored by one lane of the group) are visible to the forward pass and to the chunk copies

    // ---- phase F: forward pass
    unsigned long long t0 = a.diag ? __builtin_readcyclecounter() : 0;
    {
      State s;
      s.x = 0; s.y = 0; s.v = 0; s.th = 0; s.c = 1; s.s = 0;
      FwdConst k;
      make_fwd_const(k, KParams(phase_params()));
      double max_turn = 0.0;
      auto step = [&](const FwdIn& c, int i) {
        double u0, u1;
        forward_step(k, c, s, max_turn, u0, u1);
        if (g == 0) { double* ur = &UF(un, i, 0); ur[0] = u0; ur[S] = u1; }
        store_state(xn, i + 1, s);
      };
      if (active) {
        s.x = XF(xc, 0, 0); s.y = XF(xc, 0, 1); s.v = XF(xc, 0, 2); s.th = XF(xc, 0, 3); s.c = XF(xc, 0, 4); s.s = XF(xc, 0, 5);
        store_state(xn, 0, s);
      }
      if (faithful) {
        // With CILQR_FLAG_FAITHFUL_ITERS a solve that has rejected keeps iterating without swapping its trajectory buffers
        // while its neighbours in the wavefront go on swapping theirs, so the groups no longer agree on which buffer is
        // current and the whole-wavefront chunk copy below (one source for all) does not apply: operands come straight from
        // each group's own rows, one step ahead.
        if (active) {
          auto load_fwd = [&](FwdIn& o, int i) {
            const double* xr = &XF(xc, i, 0);
            const double* ur = &UF(uc, i, 0);
            const double* kr = &KF(i, 0);
            o.x = xr[0]; o.y = xr[S]; o.v = xr[2 * S]; o.th = xr[3 * S];
            o.u0 = ur[0]; o.u1 = ur[S];
#pragma unroll
            for (int q = 0; q < KR; ++q) o.g[q] = kr[q * S];
          };
          FwdIn fa, fb;
          load_fwd(fa, 0);
          int i = 0;
          for (; i + 1 < N; i += 2) {
            load_fwd(fb, i + 1);
            step(fa, i);
            load_fwd(fa, i + 2 < N ? i + 2 : i + 1);
            step(fb, i + 1);
          }
          if (i < N) step(fa, i);
        }
      } else if (__builtin_amdgcn_ballot_w64(active) != 0) {  // (wave-uniform) somebody still needs a forward pass
        // Every active solve of the wavefront has accepted all `it` iterations so far (a rejection ends it), so they all hold
        // their current trajectory in the same buffer, by the parity of `it`: the chunk copies have one source.
        const int xw = (it & 1) ? L.xb() : L.xa(), uw = (it & 1) ? L.ub() : L.ua();
        const int buf_doubles = chunk_f * F_ROWS * S;
        // chunks of chunk_f steps, oldest first: X rows, U rows and gain rows of a chunk side by side in a staging buffer
        auto copy_chunk = [&](int lo, int hi, double* dst) {
          const int n = hi - lo + 1;
          stage_copy(wave_ws + (size_t)(xw + lo * XR) * S, dst, n * XR * S);
          stage_copy(wave_ws + (size_t)(uw + lo * 2) * S, dst + chunk_f * XR * S, n * 2 * S);
          stage_copy(wave_ws + (size_t)(L.kk() + lo * KR) * S, dst + chunk_f * (XR + 2) * S, n * KR * S);
        };
        auto lds_fwd = [&](FwdIn& o, const double* base, int t_rel) {
          const double* xr = base + (size_t)t_rel * XR * S + grp;
          const double* ur = base + (size_t)(chunk_f * XR + t_rel * 2) * S + grp;
          const double* kr = base + (size_t)(chunk_f * (XR + 2) + t_rel * KR) * S + grp;
          o.x = xr[0]; o.y = xr[S]; o.v = xr[2 * S]; o.th = xr[3 * S];
          o.u0 = ur[0]; o.u1 = ur[S];
#pragma unroll
          for (int q = 0; q < KR; ++q) o.g[q] = kr[q * S];
        };
        int lo = 0, hi = chunk_f - 1 < N - 1 ? chunk_f - 1 : N - 1, buf = 0;
        copy_chunk(lo, hi, lds);
        stage_wait();
        __syncthreads();
        while (lo < N) {
          const int nlo = hi + 1, nhi = nlo + chunk_f - 1 < N - 1 ? nlo + chunk_f - 1 : N - 1;
          if (nlo < N) copy_chunk(nlo, nhi, lds + (buf ^ 1) * buf_doubles);
          if (active) {
            const double* cur = lds + buf * buf_doubles;
            FwdIn fa, fb;
            int i = lo;
            lds_fwd(fa, cur, 0);
            for (; i + 1 <= hi; i += 2) {
              lds_fwd(fb, cur, i + 1 - lo);
              step(fa, i);
              lds_fwd(fa, cur, i + 2 <= hi ? i + 2 - lo : 0);
              step(fb, i + 1);
            }
            if (i == hi) step(fa, hi);
          }
          stage_wait();  // the next chunk has landed (and this chunk's stores have been accepted)
          __syncthreads();
          lo = nlo;
          hi = nhi;
          buf ^= 1;
        }
      }
      if (active && !(max_turn <= MAX_TURN)) { handover = true; active = false; }
    }
    mem_sync();
    if (a.diag && active) tF += __builtin_readcyclecounter() - t0;

    if (active) {
      if (accept) {
        int sx = xc; xc = xn; xn = sx;
        int su = uc; uc = un; un = su;
        if (g == 0) { int* qb = reinterpret_cast<int*>(par + (size_t)grp * PAR + P_BUF); qb[0] = xc; qb[1] = uc; }
        j_valid = false;
        lamb = lamb / kp.lamb_factor;
        if (fabs(J_new - J_old) < kp.tolerance) { status = CILQR_EXIT_TOLERANCE; active = false; }
      } else {
        lamb = lamb * kp.lamb_factor;
        if (lamb > kp.lamb_max) { status = CILQR_EXIT_LAMBDA_MAX; active = false; }
      }
      J_old = J_new;
    }
  }

  if (!live) return;
  if (g == 0) aq->redo[b] = handover ? 1 : 0;
  if (handover) return;  // outputs (and the in/out U) untouched: the GENERAL kernel starts from the same inputs

  // ---- epilogue: X_result / U_result (:243-244)
  const SolveArgs ae = *aq;  // output pointers read here, not carried through the loop (cilqr_device.hpp::phase_args)
  double* Uo = ae.U + (size_t)b * 2 * N;
  for (int t = g; t < N; t += G) {
    Uo[2 * t] = UF(uc, t, 0);
    Uo[2 * t + 1] = UF(uc, t, 1);
  }
  double* Xg = ae.X_out + (size_t)b * 4 * (N + 1);
  for (int t = g; t <= N; t += G) {
#pragma unroll
    for (int r = 0; r < 4; ++r) Xg[4 * t + r] = XF(xc, t, r);
  }
  if (ae.J_out) {
    if (!j_valid) {
      double Jp = 0.0;
      for (int t = g; t < N; t += G) {
        const double px = XF(xc, t, 0), py = XF(xc, t, 1);
        const int cs = closest_sample<false>(NS, grid, px, py, sample_at);
        double cx, cy;
        sample_xy(grid, pc, cs, cx, cy);
        Jp += stage_cost(ae.kp, px - cx, py - cy, XF(xc, t, 2) - ae.kp.desired_speed, UF(uc, t, 0), UF(uc, t, 1));
      }
      J_new = group_sum<G>(Jp);
    }
    if (g == 0) ae.J_out[b] = J_new;
  }
  if (a.diag && g == 0) {  // {prologue, L, R, F, epilogue, #L, #R, total}: prologue/epilogue not separated here
    unsigned long long* d = a.diag + (size_t)b * DIAG_SLOTS;
    d[0] = 0; d[1] = tL; d[2] = tR; d[3] = tF; d[4] = 0; d[5] = nL; d[6] = nR; d[7] = __builtin_readcyclecounter() - t_begin;
  }
  if (g == 0) {
    if (ae.iters_out) ae.iters_out[b] = iters;
    if (ae.status_out) ae.status_out[b] = status;
    if (ae.passes) ae.passes[b] = n_pass;
  }
#undef XF
#undef UF
#undef KF
}

// ---- the GENERAL kernel: branching passes for the solves the production kernel handed over (redo[b] != 0) --------------------
// Rare by construction (Q_uu not positive semi-definite or not finite, a heading beyond the in-loop sincos range), so it is
// written for clarity: one divergent loop per group, every operand straight from the global workspace, the linearisation
// records through a workspace region of their own (`rec_base`, [N][16][S] per wavefront).
template <int G>
__global__ __launch_bounds__(WAVE) void cilqr_solve_groups_general(SolveArgs a, double* ws_base, double* rec_base) {
  constexpr int S = WAVE / G;
  const int lane = threadIdx.x, grp = lane / G, g = lane % G;
  const int b = blockIdx.x * S + grp;
  if (b >= a.B) return;  // whole groups leave together
  if (a.redo[b] == 0) return;
  const KParams kp = a.kp;
  const int N = a.N, M = a.M, NS = kp.n_samples;
  const WsLayout L{N, M};
  double* ws = ws_base + (size_t)blockIdx.x * L.rows() * S + grp;
  double* rws = rec_base + (size_t)blockIdx.x * ((size_t)N * REC) * S + grp;
  double* tab = a.obs_tab + (size_t)blockIdx.x * ((size_t)M * N * TABF) * S + (size_t)grp * TABF;
#define XF(base, t, f) ws[(size_t)((base) + (t) * XR + (f)) * S]
#define UF(base, t, f) ws[(size_t)((base) + (t) * 2 + (f)) * S]
#define RF(t, f) rws[(size_t)((t) * REC + (f)) * S]
#define KF(t, f) ws[(size_t)(L.kk() + (t) * KR + (f)) * S]
  double pc[CILQR_POLY_COEFFS];
#pragma unroll
  for (int j = 0; j < CILQR_POLY_COEFFS; ++j) pc[j] = a.poly[(size_t)b * CILQR_POLY_COEFFS + j];
  SampleGrid grid;
  make_sample_grid(grid, a.xplan_fl[2 * b], a.xplan_fl[2 * b + 1], NS);  // dmax stays +inf: the x-side window only
  auto sample_at = [&](int s, double& x, double& y) { sample_xy(grid, pc, s, x, y); };
  double* Ug = a.U + (size_t)b * 2 * N;
  for (int t = g; t < N; t += G) {
    UF(L.ua(), t, 0) = Ug[2 * t];
    UF(L.ua(), t, 1) = Ug[2 * t + 1];
  }
  const double* wts = a.obs_weight ? a.obs_weight + (size_t)b * M : nullptr;
  for (int m = 0; m < M; ++m)  // the production kernel fills only the step-0 row of a horizon-constant obstacle: fill them all
    for (int t = g; t < N; t += G) {
      const ObsEntry e = make_obs_entry(kp, a.obs_pose + (((size_t)b * M + m) * N + t) * 4, a.obs_dim + (((size_t)b * M + m) * N + t) * 2);
      double* o = tab + (size_t)(m * N + t) * S * TABF;
      o[0] = e.ox; o[1] = e.oy; o[2] = e.co; o[3] = e.so; o[4] = e.ia2; o[5] = e.ib2;
    }
  mem_sync();
  auto store_state = [&](int base, int t, const State& s) {
    if (g == 0) {
      double* xr = &XF(base, t, 0);
      xr[0] = s.x; xr[S] = s.y; xr[2 * S] = s.v; xr[3 * S] = s.th; xr[4 * S] = s.c; xr[5 * S] = s.s;
    }
  };
  {  // nominal rollout, I/iLQR.cpp:51-62
    const double* x0 = a.x0 + (size_t)b * 4;
    State s;
    s.x = x0[0]; s.y = x0[1]; s.v = x0[2]; s.th = x0[3];
    sincos_fast(s.th, &s.s, &s.c);
    store_state(L.xa(), 0, s);
    for (int i = 0; i < N; ++i) {
      s = dyn_step(kp, s, UF(L.ua(), i, 0), UF(L.ua(), i, 1));
      store_state(L.xa(), i + 1, s);
    }
  }
  mem_sync();
  int xc = L.xa(), xn = L.xb(), uc = L.ua(), un = L.ub();
  double J_old = DBL_MAX, lamb = 1.0, J_new = 0.0;
  int iters = 0, status = CILQR_EXIT_MAX_ITER, n_pass = 0;
  bool j_valid = false;
  const bool faithful = (a.flags & CILQR_FLAG_FAITHFUL_ITERS) != 0;
  const int max_it = kp.max_iterations;
  const double dt = kp.dt, two_wvel = kp.w_vel * 2;
  const UncArgs* unc = a.unc.layer ? &phase_args().unc : nullptr;
  UncPose upose{0, 0, 1, 0};
  if (unc) upose = unc_pose(a.unc, b);
  for (int it = 0; it < max_it; ++it) {
    ++iters;
    double Jpart = 0.0;
    for (int t = g; t < N; t += G) {  // phase L
      const double px = XF(xc, t, 0), py = XF(xc, t, 1);
      const int cs = closest_sample<false>(NS, grid, px, py, sample_at);
      double cx, cy;
      sample_xy(grid, pc, cs, cx, cy);
      auto obs = [&](int m, ObsEntry& e, double& w) {
        const double* p = tab + ((size_t)m * N + t) * S * TABF;
        e.ox = p[0]; e.oy = p[1]; e.co = p[2]; e.so = p[3]; e.ia2 = p[4]; e.ib2 = p[5];
        w = wts ? wts[m] : kp.w_obstacle;
        return true;
      };
      Rec c;
      Jpart += lin_step(kp, px, py, XF(xc, t, 2), XF(xc, t, 4), XF(xc, t, 5), UF(uc, t, 0), UF(uc, t, 1), XF(xc, t + 1, 2),
                        XF(xc, t + 1, 4), XF(xc, t + 1, 5), cx, cy, M, obs, c);
      if (unc) unc_cost_add(*unc, upose, b, px, py, XF(xc, t, 4), XF(xc, t, 5), c.lx0, c.lx1, c.l00, c.l01, c.l11);
      double* r = &RF(t, 0);
      r[0] = c.lx0; r[S] = c.lx1; r[2 * S] = c.lx2; r[3 * S] = c.l00; r[4 * S] = c.l01; r[5 * S] = c.l11;
      r[6 * S] = c.lu0; r[7 * S] = c.lu1; r[8 * S] = c.luu0; r[9 * S] = c.luu1;
      r[10 * S] = c.al; r[11 * S] = c.be; r[12 * S] = c.ga; r[13 * S] = c.de; r[14 * S] = c.p; r[15 * S] = c.q;
    }
    J_new = group_sum<G>(Jpart);
    j_valid = true;
    mem_sync();
    const bool accept = J_new < J_old;
    if (!accept && !faithful) {
      if (J_new != J_new) { status = CILQR_EXIT_NUMERIC; break; }
      for (;;) {
        lamb = lamb * kp.lamb_factor;
        if (lamb > kp.lamb_max) { status = CILQR_EXIT_LAMBDA_MAX; break; }
        if (++it >= max_it) { status = CILQR_EXIT_MAX_ITER; break; }
        ++iters;
      }
      break;
    }
    bool r_ok = true;
    {  // phase R
      Value V;
      Rec c;
      Gains gn;
      for (int j = N - 1; j >= 0 && r_ok; --j) {
        const double* r = &RF(j, 0);
        c.lx0 = r[0]; c.lx1 = r[S]; c.lx2 = r[2 * S]; c.l00 = r[3 * S]; c.l01 = r[4 * S]; c.l11 = r[5 * S];
        c.lu0 = r[6 * S]; c.lu1 = r[7 * S]; c.luu0 = r[8 * S]; c.luu1 = r[9 * S];
        c.al = r[10 * S]; c.be = r[11 * S]; c.ga = r[12 * S]; c.de = r[13 * S]; c.p = r[14 * S]; c.q = r[15 * S];
        if (j == N - 1) value_terminal(V, c, two_wvel);
        bool ok;
        riccati_step<false>(c, V, dt, two_wvel, lamb, gn, ok);
        r_ok = r_ok && ok;
        if (g == 0 && ok) {
          double* kp_ = &KF(j, 0);
#pragma unroll
          for (int i = 0; i < KR; ++i) kp_[i * S] = gn.g[i];
        }
      }
    }
    if (!r_ok) { status = CILQR_EXIT_NUMERIC; break; }
    mem_sync();
    ++n_pass;
    {  // phase F
      State s;
      s.x = XF(xc, 0, 0); s.y = XF(xc, 0, 1); s.v = XF(xc, 0, 2); s.th = XF(xc, 0, 3); s.c = XF(xc, 0, 4); s.s = XF(xc, 0, 5);
      store_state(xn, 0, s);
      for (int i = 0; i < N; ++i) {
        const double* kr = &KF(i, 0);
        const double d0 = s.x - XF(xc, i, 0), d1 = s.y - XF(xc, i, 1), d2 = s.v - XF(xc, i, 2), d3 = s.th - XF(xc, i, 3);
        const double u0 = fma(kr[5 * S], d3, fma(kr[4 * S], d2, fma(kr[3 * S], d1, fma(kr[2 * S], d0, UF(uc, i, 0) + kr[0]))));
        const double u1 = fma(kr[9 * S], d3, fma(kr[8 * S], d2, fma(kr[7 * S], d1, fma(kr[6 * S], d0, UF(uc, i, 1) + kr[S]))));
        s = dyn_step(kp, s, u0, u1);
        if (g == 0) { double* ur = &UF(un, i, 0); ur[0] = u0; ur[S] = u1; }
        store_state(xn, i + 1, s);
      }
    }
    mem_sync();
    if (accept) {
      int sx = xc; xc = xn; xn = sx;
      int su = uc; uc = un; un = su;
      j_valid = false;
      lamb = lamb / kp.lamb_factor;
      if (fabs(J_new - J_old) < kp.tolerance) { status = CILQR_EXIT_TOLERANCE; break; }
    } else {
      lamb = lamb * kp.lamb_factor;
      if (lamb > kp.lamb_max) { status = CILQR_EXIT_LAMBDA_MAX; break; }
    }
    J_old = J_new;
  }
  double* Uo = a.U + (size_t)b * 2 * N;
  for (int t = g; t < N; t += G) {
    Uo[2 * t] = UF(uc, t, 0);
    Uo[2 * t + 1] = UF(uc, t, 1);
  }
  double* Xg = a.X_out + (size_t)b * 4 * (N + 1);
  for (int t = g; t <= N; t += G) {
#pragma unroll
    for (int r = 0; r < 4; ++r) Xg[4 * t + r] = XF(xc, t, r);
  }
  if (a.J_out) {
    if (!j_valid) {
      double Jp = 0.0;
      for (int t = g; t < N; t += G) {
        const double px = XF(xc, t, 0), py = XF(xc, t, 1);
        const int cs = closest_sample<false>(NS, grid, px, py, sample_at);
        double cx, cy;
        sample_xy(grid, pc, cs, cx, cy);
        Jp += stage_cost(kp, px - cx, py - cy, XF(xc, t, 2) - kp.desired_speed, UF(uc, t, 0), UF(uc, t, 1));
      }
      J_new = group_sum<G>(Jp);
    }
    if (g == 0) a.J_out[b] = J_new;
  }
  if (g == 0) {
    if (a.iters_out) a.iters_out[b] = iters;
    if (a.status_out) a.status_out[b] = status;
    if (a.passes) a.passes[b] = n_pass;
  }
#undef XF
#undef UF
#undef RF
#undef KF
}

constexpr int LDS_BUDGET = 40 * 1024;  // per wavefront, everything: four wavefronts per CU, one per SIMD, within the CU's 160 KiB

template <int G>
void launch_g(const SolveArgs& a, double* ws, hipStream_t stream) {
  constexpr int S = WAVE / G;
  const int blocks = (a.B + S - 1) / S;
  // behind the chunk area: the per-solve blocks and the list of active solves (phase L's lane sharing), and one stage cost per
  // step of the L→R chunk in hand
  const int fixed = (S * PAR + (S + 1) / 2) * (int)sizeof(double);
  // L→R hand-over chunk: as many steps as fit (16 record doubles + 1 stage cost per step and solve), a whole number of rounds of
  // the group's G lanes
  int chunk_r = (LDS_BUDGET - fixed) / ((REC + 1) * S * (int)sizeof(double));
  chunk_r = chunk_r / G * G;
  if (chunk_r < G) chunk_r = G;
  if (chunk_r > a.N) chunk_r = (a.N + G - 1) / G * G;
  // forward-pass staging chunk: two buffers
  int chunk_f = (LDS_BUDGET - fixed - chunk_r * S * (int)sizeof(double)) / (2 * F_ROWS * S * (int)sizeof(double));
  if (chunk_f < 1) chunk_f = 1;
  if (chunk_f > a.N) chunk_f = a.N;
  const size_t lds_r = (size_t)chunk_r * REC * S * sizeof(double), lds_f = (size_t)2 * chunk_f * F_ROWS * S * sizeof(double);
  const size_t lds_main = lds_r > lds_f ? lds_r : lds_f;
  const size_t lds = lds_main + (size_t)chunk_r * S * sizeof(double) + (size_t)fixed;
  const int lds_main_doubles = (int)(lds_main / sizeof(double));
  const WsLayout L{a.N, a.M};
  double* rec_ws = ws + ((size_t)a.B + WAVE - 1) / WAVE * WAVE * (size_t)L.rows();  // behind the trajectory/gain blocks
  if (a.unc.layer) hipLaunchKernelGGL((cilqr_solve_groups_fast<G, true>), dim3(blocks), dim3(WAVE), lds, stream, a, ws, chunk_r, chunk_f, lds_main_doubles);
  else hipLaunchKernelGGL((cilqr_solve_groups_fast<G, false>), dim3(blocks), dim3(WAVE), lds, stream, a, ws, chunk_r, chunk_f, lds_main_doubles);
  hipLaunchKernelGGL((cilqr_solve_groups_general<G>), dim3(blocks), dim3(WAVE), 0, stream, a, ws, rec_ws);
}

}  // namespace

size_t solve_groups_ws_doubles(int B, int N) {
  // padded to whole wavefronts of the narrowest grouping (G = 1 → 64 solves per wavefront): trajectories + gains, then the
  // GENERAL kernel's record region
  const size_t Bp = ((size_t)B + WAVE - 1) / WAVE * WAVE;
  const WsLayout L{N, 0};
  return Bp * ((size_t)L.rows() + (size_t)N * REC);
}

hipError_t launch_solve_groups(const SolveArgs& a, int G, double* ws, hipStream_t stream) {
  if (a.B <= 0) return hipSuccess;
  switch (G) {
    case 1: launch_g<1>(a, ws, stream); break;
    case 2: launch_g<2>(a, ws, stream); break;
    case 4: launch_g<4>(a, ws, stream); break;
    case 8: launch_g<8>(a, ws, stream); break;
    case 16: launch_g<16>(a, ws, stream); break;
    case 32: launch_g<32>(a, ws, stream); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace cilqr
