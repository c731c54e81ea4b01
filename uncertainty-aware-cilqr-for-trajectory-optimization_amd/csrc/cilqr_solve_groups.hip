// cilqr_solve_groups.hip — batched constrained-iLQR solve for gfx950 (MI355X), G LANES PER SOLVE, 64/G solves per
// wavefront, per-solve arrays in a global-memory workspace.  The kernel family for batches larger than about one solve
// per SIMD (cilqr_solve.hip holds the one-wavefront-per-solve, LDS-resident family; launch_solve picks).
//
// Why a second mapping.  A wavefront issues one instruction per ≈5 ticks whatever it is.  In the one-wavefront-per-solve
// kernel the sequential phases (backward Riccati recursion R, forward pass F; I/iLQR.cpp:133-191, 68-86) keep 1 useful
// lane in 64, and LDS (≈20-30 KiB per solve) caps residency at 5-8 solves per CU, so a large batch runs in many rounds at
// a few per cent lane use.  Here a wavefront carries S = 64/G solves: R and F run once for S solves (lanes of a group
// compute redundantly, groups differ), the lane-parallel linearisation L (I/Constraints.cpp:145-227, 86-137) spreads each
// solve's N steps over its G lanes, and the arrays live in global memory laid out [row][solve-in-wavefront] so that a
// wavefront instruction touches S consecutive doubles per row (L2 / Infinity-Cache resident for the batch sizes this is
// chosen for).  launch_solve picks G ≈ 65536/B (power of two) so that the grid is about one wavefront per SIMD.
//
// Arithmetic is shared with the LDS family (cilqr_device.hpp): same per-step functions, same samples, same fast/GENERAL
// split with the redo hand-over.  Path samples are recomputed where needed (12 instructions) instead of stored.
#include "cilqr_device.hpp"

namespace cilqr {

using namespace dev;

namespace {

// Per-wavefront workspace block, in rows of S doubles (one column per solve of the wavefront), STEP-major:
// row(field f of an array, step t) = array base + t*(fields of that array) + f.  The serial phases R and F, which take most
// of a solve, then address a step's 16 + 10 operands with ONE base register and immediate offsets f·S·8 (field-major rows
// needed a 64-bit address computation per operand: R was 1 712 ticks per step against 930 in the LDS family, and requesting
// operands further ahead did not help — it was issue, not latency).  Phase L pays with strided rows: the lanes of a load
// (G steps × S solves) touch G separate S·8-byte segments instead of 512 contiguous bytes, 26 such accesses per step
// against thousands of arithmetic instructions.
struct WsLayout {
  int N, M;
  __device__ __host__ int xa() const { return 0; }
  __device__ __host__ int xb() const { return (N + 1) * XR; }
  __device__ __host__ int ua() const { return 2 * (N + 1) * XR; }
  __device__ __host__ int ub() const { return ua() + 2 * N; }
  __device__ __host__ int rec() const { return ub() + 2 * N; }
  __device__ __host__ int kk() const { return rec() + REC * N; }
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

// Staging of the serial phases' operands (fast kernel).  Loads and stores of a wavefront retire through one in-order counter
// on this hardware, so operands requested one step ahead from global memory queue behind the previous step's stores and
// every step waits out a store round trip (R measured 1 580 ticks per step against 930 for the LDS-resident family).  Here
// the operands of a whole chunk of steps are copied global → LDS by the asynchronous direct-to-LDS load (no registers), one
// chunk ahead; the step loops read LDS and only store to global; the one vmcnt wait per chunk finds the copy long finished.
extern __shared__ __attribute__((aligned(16))) double cilqr_groups_stage[];
__device__ __forceinline__ double* stage_base() { return cilqr_groups_stage; }
constexpr int STAGE_ROWS = XR + 2 + KR;  // per step: the larger of R's 16 record rows and F's 6 + 2 + 10 state/control/gain rows
// n_doubles (even) contiguous doubles from src (global, wave-uniform) to dst (LDS, wave-uniform): 16 bytes per lane and
// instruction, 1 KiB per instruction.  The copy is a job of the whole wavefront, but the solves of a wavefront finish at
// different iterations and their lanes are switched off from then on — so each piece is issued with EXEC forced to the lanes
// it needs (and restored), the lane's byte offset formed inside from v_mbcnt, the LDS base in M0 (tools/ubench_lds_direct.hip
// checks this sequence from divergent code).  The compiler does not know these loads: stage_wait() is their only wait.
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void stage_copy(const double* src, double* dst, int n_doubles) {
  // every quantity here is the same in all active lanes; readfirstlane tells the compiler so (the loop counters they come from
  // live in vector registers because the iteration loop is divergent)
  const int bytes = __builtin_amdgcn_readfirstlane(n_doubles * 8);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_groupstaticsize() + (unsigned)((dst - stage_base()) * sizeof(double)));
  const unsigned long long src0 = uniform64(reinterpret_cast<unsigned long long>(src));
  for (int off = 0; off < bytes; off += WAVE * 16) {
    const int lanes = (bytes - off) / 16;
    const unsigned long long mask = uniform64(lanes >= WAVE ? ~0ull : ((1ull << lanes) - 1));
    const unsigned long long piece = uniform64(src0 + (unsigned long long)off);
    const unsigned lds_byte = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)off));
    unsigned long long save;
    unsigned vtmp, m0_save;
    asm volatile(
        "s_mov_b64 %0, exec\n\t"
        "s_mov_b32 %2, m0\n\t"
        "s_mov_b64 exec, %4\n\t"
        "v_mbcnt_lo_u32_b32 %1, -1, 0\n\t"
        "v_mbcnt_hi_u32_b32 %1, -1, %1\n\t"
        "v_lshlrev_b32 %1, 4, %1\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_mov_b64 exec, %0"
        : "=&s"(save), "=&v"(vtmp), "=&s"(m0_save)
        : "s"(piece), "s"(mask), "s"(lds_byte)
        : "memory");
  }
}
__device__ __forceinline__ void stage_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int G, bool GENERAL>
__global__ __launch_bounds__(WAVE) void cilqr_solve_groups_kernel(SolveArgs a, double* ws_base, int chunk) {
  double* stage = stage_base();  // fast kernel: 2 staging buffers of chunk·STAGE_ROWS·S doubles (dynamic LDS)
  constexpr int S = WAVE / G;
  const int lane = threadIdx.x, grp = lane / G, g = lane % G;
  const int b = blockIdx.x * S + grp;
  if (b >= a.B) return;  // whole groups leave together
  if (GENERAL && a.redo[b] == 0) return;
  const SolveArgs* aq = &phase_args();  // taken while the wavefront is still whole; read in the epilogue
  const KParams kp = a.kp;
  const int N = a.N, M = a.M, NS = kp.n_samples;
  const WsLayout L{N, M};

  // column `grp` of this wavefront's block: element (row r) at ws[r * S]
  double* ws = ws_base + (size_t)blockIdx.x * L.rows() * S + grp;
  // obstacle table of this wavefront: entry (m, t, solve) = 6 contiguous doubles at ((m*N + t)*S + solve)*6
  double* tab = a.obs_tab + (size_t)blockIdx.x * ((size_t)M * N * TABF) * S + (size_t)grp * TABF;
#define XF(base, t, f) ws[(size_t)((base) + (t) * XR + (f)) * S]  /* state arrays: (N+1) steps x 6 fields */
#define UF(base, t, f) ws[(size_t)((base) + (t) * 2 + (f)) * S]   /* control arrays: N steps x 2 fields */
#define RF(t, f) ws[(size_t)(L.rec() + (t) * REC + (f)) * S]      /* linearisation: N steps x 16 fields */
#define KF(t, f) ws[(size_t)(L.kk() + (t) * KR + (f)) * S]        /* gains: N steps x 10 fields */

  // ---- prologue -------------------------------------------------------------------------------------------
  double pc[CILQR_POLY_COEFFS];
#pragma unroll
  for (int j = 0; j < CILQR_POLY_COEFFS; ++j) pc[j] = a.poly[(size_t)b * CILQR_POLY_COEFFS + j];
  SampleGrid grid;
  make_sample_grid(grid, a.xplan_fl[2 * b], a.xplan_fl[2 * b + 1], NS);
  auto sample_at = [&](int s, double& x, double& y) { sample_xy(grid, pc, s, x, y); };
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

  double* Ug = a.U + (size_t)b * 2 * N;
  for (int t = g; t < N; t += G) {
    UF(L.ua(), t, 0) = Ug[2 * t];
    UF(L.ua(), t, 1) = Ug[2 * t + 1];
  }
  const double* wts = a.obs_weight ? a.obs_weight + (size_t)b * M : nullptr;
  // Obstacle table, I/Obstacle.cpp:41-62.  An obstacle whose pose and dimensions are the same in every column (how the
  // reference node feeds static obstacles: one pose replicated over the horizon, I/ilqr_uncertainty_node.cpp:175-185) gets a
  // bit in `held`: phase L then reads its step-0 row for every step — the same values, but one cache-resident line per
  // field instead of a stream of N·6 doubles per obstacle, solve and iteration from HBM.
  unsigned long long held = 0;
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
  mem_sync();

  auto store_state = [&](int base, int t, const State& s) {
    if (g == 0) {
      double* xr = &XF(base, t, 0);
      xr[0] = s.x; xr[S] = s.y; xr[2 * S] = s.v; xr[3 * S] = s.th; xr[4 * S] = s.c; xr[5 * S] = s.s;
    }
  };

  bool handover = false;
  {  // nominal rollout, I/iLQR.cpp:51-62
    const double* x0 = a.x0 + (size_t)b * 4;
    State s;
    s.x = x0[0]; s.y = x0[1]; s.v = x0[2]; s.th = x0[3];
    if (GENERAL) {
      sincos_fast(s.th, &s.s, &s.c);
      store_state(L.xa(), 0, s);
      for (int i = 0; i < N; ++i) {
        s = dyn_step(kp, s, UF(L.ua(), i, 0), UF(L.ua(), i, 1));
        store_state(L.xa(), i + 1, s);
      }
    } else {
      FwdConst k;
      make_fwd_const(k, kp);
      double max_th = fabs(s.th);
      sincos_loop(k, s.th, s.s, s.c);
      store_state(L.xa(), 0, s);
      for (int i = 0; i < N; ++i) {
        dyn_step_loop(k, s, UF(L.ua(), i, 0), UF(L.ua(), i, 1), max_th);
        store_state(L.xa(), i + 1, s);
      }
      handover = !(max_th < 1.0e6);
    }
  }
  mem_sync();

  // ---- iteration loop, I/iLQR.cpp:204-239 (per group; groups of one wavefront diverge freely) -----------------------
  int xc = L.xa(), xn = L.xb(), uc = L.ua(), un = L.ub();
  double J_old = DBL_MAX, lamb = 1.0, J_new = 0.0;
  int iters = 0, status = CILQR_EXIT_MAX_ITER, n_pass = 0;
  bool j_valid = false;
  const bool faithful = (a.flags & CILQR_FLAG_FAITHFUL_ITERS) != 0;
  const int max_it = kp.max_iterations;
  const double dt = kp.dt, two_wvel = kp.w_vel * 2;

  unsigned long long tL = 0, tR = 0, tF = 0, nL = 0, nR = 0;
  const unsigned long long t_begin = __builtin_readcyclecounter();
  for (int it = 0; it < max_it && !handover; ++it) {
    ++iters;
    // ---- phase L: this group's N steps over its G lanes
    unsigned long long t0 = a.diag ? __builtin_readcyclecounter() : 0;
    double Jpart = 0.0;
    {
      // this lane's steps t = g, g+G, …: the ten operands of the next one are requested while the current one computes
      struct LIn { double px, py, v, ct, st, u0, u1, vn, cn, sn; };
      auto load_in = [&](LIn& o, int t) {
        const double* xr = &XF(xc, t, 0);
        const double* xq = &XF(xc, t + 1, 0);
        const double* ur = &UF(uc, t, 0);
        o.px = xr[0]; o.py = xr[S]; o.v = xr[2 * S]; o.ct = xr[4 * S]; o.st = xr[5 * S];
        o.u0 = ur[0]; o.u1 = ur[S];
        o.vn = xq[2 * S]; o.cn = xq[4 * S]; o.sn = xq[5 * S];
      };
      const KParams kpl = phase_params();  // this phase's own read of the parameter block (cilqr_device.hpp)
      UncProbe probe;
      const UncProbe* unc = nullptr;
      if (aq->unc.layer) {  // a map is set (cilqr_set_uncertainty_map*): the same for every solve of the launch
        probe = make_unc_probe(aq->unc, b);
        unc = &probe;
      }
      LIn cur, nxt;
      if (g < N) load_in(cur, g);
      for (int t = g; t < N; t += G) {
        if (t + G < N) load_in(nxt, t + G);
        const int cs = closest_sample(NS, grid, cur.px, cur.py, sample_at);
        double cx, cy;
        sample_xy(grid, pc, cs, cx, cy);
        // one 48-byte entry = three 16-byte loads from one address; a held obstacle reads its step-0 entry
        auto obs = [&](int m, ObsEntry& e, double& w) {
          const int row = (m < 64 && ((held >> m) & 1)) ? 0 : t;
          const double2* p = reinterpret_cast<const double2*>(tab + ((size_t)m * N + row) * S * TABF);
          const double2 q0 = p[0], q1 = p[1], q2 = p[2];
          e.ox = q0.x; e.oy = q0.y; e.co = q1.x; e.so = q1.y; e.ia2 = q2.x; e.ib2 = q2.y;
          w = wts ? wts[m] : kpl.w_obstacle;
          return true;
        };
        Rec c;
        Jpart += lin_step(kpl, cur.px, cur.py, cur.v, cur.ct, cur.st, cur.u0, cur.u1, cur.vn, cur.cn, cur.sn, cx, cy, M, obs, c, unc);
        double* r = &RF(t, 0);
        r[0] = c.lx0; r[S] = c.lx1; r[2 * S] = c.lx2; r[3 * S] = c.l00; r[4 * S] = c.l01; r[5 * S] = c.l11;
        r[6 * S] = c.lu0; r[7 * S] = c.lu1; r[8 * S] = c.luu0; r[9 * S] = c.luu1;
        r[10 * S] = c.al; r[11 * S] = c.be; r[12 * S] = c.ga; r[13 * S] = c.de; r[14 * S] = c.p; r[15 * S] = c.q;
        cur = nxt;
      }
    }
    J_new = group_sum<G>(Jpart);
    j_valid = true;
    mem_sync();
    if (a.diag) { const unsigned long long t1 = __builtin_readcyclecounter(); tL += t1 - t0; ++nL; t0 = t1; }

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

    // ---- phase R: backward recursion (all lanes of the group compute the same values)
    bool r_ok = true;
    {
      // one 64-bit base per step, the 16 operands at immediate offsets f·S doubles from it
      auto load_rec = [&](Rec& o, int j) {
        const double* r = &RF(j, 0);
        o.lx0 = r[0]; o.lx1 = r[S]; o.lx2 = r[2 * S]; o.l00 = r[3 * S]; o.l01 = r[4 * S]; o.l11 = r[5 * S];
        o.lu0 = r[6 * S]; o.lu1 = r[7 * S]; o.luu0 = r[8 * S]; o.luu1 = r[9 * S];
        o.al = r[10 * S]; o.be = r[11 * S]; o.ga = r[12 * S]; o.de = r[13 * S]; o.p = r[14 * S]; o.q = r[15 * S];
      };
      Rec ra, rb;
      Value V;
      if (GENERAL) {
        load_rec(ra, N - 1);
        value_terminal(V, ra, two_wvel);
      }
      Gains gn;
      bool ok;
      auto step = [&](const Rec& c, int j) {
        riccati_step<!GENERAL>(c, V, dt, two_wvel, lamb, gn, ok);
        r_ok = r_ok && ok;
        if (g == 0 && (!GENERAL || ok)) {
          double* kp_ = &KF(j, 0);
#pragma unroll
          for (int i = 0; i < KR; ++i) kp_[i * S] = gn.g[i];
        }
      };
      if (GENERAL) {
        int j = N - 1;
        for (; j >= 1 && r_ok; j -= 2) {
          load_rec(rb, j - 1);
          step(ra, j);
          load_rec(ra, j >= 2 ? j - 2 : 0);
          if (r_ok) step(rb, j - 1);
        }
        if (j == 0 && r_ok) step(ra, 0);
      } else {
        // chunks of `chunk` steps, newest first; chunk c+1 is copied while chunk c is computed
        const double* wave_ws = ws_base + (size_t)blockIdx.x * L.rows() * S;  // this wavefront's block, column 0
        const int buf_doubles = chunk * STAGE_ROWS * S;
        auto rows_of = [&](int lo) { return wave_ws + (size_t)(L.rec() + lo * REC) * S; };
        auto lds_rec = [&](Rec& o, const double* base, int t_rel) {
          const double* r = base + (size_t)t_rel * REC * S + grp;
          o.lx0 = r[0]; o.lx1 = r[S]; o.lx2 = r[2 * S]; o.l00 = r[3 * S]; o.l01 = r[4 * S]; o.l11 = r[5 * S];
          o.lu0 = r[6 * S]; o.lu1 = r[7 * S]; o.luu0 = r[8 * S]; o.luu1 = r[9 * S];
          o.al = r[10 * S]; o.be = r[11 * S]; o.ga = r[12 * S]; o.de = r[13 * S]; o.p = r[14 * S]; o.q = r[15 * S];
        };
        int hi = N - 1, lo = hi - chunk + 1 < 0 ? 0 : hi - chunk + 1, buf = 0;
        stage_copy(rows_of(lo), stage, (hi - lo + 1) * REC * S);
        stage_wait();
        while (hi >= 0) {
          const int nhi = lo - 1, nlo = nhi - chunk + 1 < 0 ? 0 : nhi - chunk + 1;
          if (nhi >= 0) stage_copy(rows_of(nlo), stage + (buf ^ 1) * buf_doubles, (nhi - nlo + 1) * REC * S);
          const double* cur = stage + buf * buf_doubles;
          int j = hi;
          lds_rec(ra, cur, j - lo);
          if (j == N - 1) value_terminal(V, ra, two_wvel);
          for (; j >= lo + 1; j -= 2) {  // two steps per trip, the next record's LDS reads under this step's arithmetic
            lds_rec(rb, cur, j - 1 - lo);
            step(ra, j);
            lds_rec(ra, cur, j - 2 >= lo ? j - 2 - lo : 0);
            step(rb, j - 1);
          }
          if (j == lo) step(ra, lo);
          stage_wait();  // the next chunk has landed (and this chunk's gain stores have been accepted)
          hi = nhi;
          lo = nlo;
          buf ^= 1;
        }
      }
    }
    if (!r_ok) {
      if (GENERAL) { status = CILQR_EXIT_NUMERIC; break; }
      handover = true;
      break;
    }
    mem_sync();
    ++n_pass;
    if (a.diag) { const unsigned long long t1 = __builtin_readcyclecounter(); tR += t1 - t0; ++nR; t0 = t1; }

    // ---- phase F: forward pass
    {
      State s;
      s.x = XF(xc, 0, 0); s.y = XF(xc, 0, 1); s.v = XF(xc, 0, 2); s.th = XF(xc, 0, 3); s.c = XF(xc, 0, 4); s.s = XF(xc, 0, 5);
      store_state(xn, 0, s);
      auto load_fwd = [&](FwdIn& o, int i) {
        const double* xr = &XF(xc, i, 0);
        const double* ur = &UF(uc, i, 0);
        const double* kr = &KF(i, 0);
        o.x = xr[0]; o.y = xr[S]; o.v = xr[2 * S]; o.th = xr[3 * S];
        o.u0 = ur[0]; o.u1 = ur[S];
#pragma unroll
        for (int k = 0; k < KR; ++k) o.g[k] = kr[k * S];
      };
      if (GENERAL) {
        for (int i = 0; i < N; ++i) {
          FwdIn c;
          load_fwd(c, i);
          const double d0 = s.x - c.x, d1 = s.y - c.y, d2 = s.v - c.v, d3 = s.th - c.th;
          const double u0 = fma(c.g[5], d3, fma(c.g[4], d2, fma(c.g[3], d1, fma(c.g[2], d0, c.u0 + c.g[0]))));
          const double u1 = fma(c.g[9], d3, fma(c.g[8], d2, fma(c.g[7], d1, fma(c.g[6], d0, c.u1 + c.g[1]))));
          s = dyn_step(kp, s, u0, u1);
          if (g == 0) { double* ur = &UF(un, i, 0); ur[0] = u0; ur[S] = u1; }
          store_state(xn, i + 1, s);
        }
      } else {
        FwdConst k;
        const KParams kpf = phase_params();
        make_fwd_const(k, kpf);
        double max_th = fabs(s.th);
        FwdIn fa, fb;
        auto step = [&](const FwdIn& c, int i) {
          double u0, u1;
          forward_step(k, c, s, max_th, u0, u1);
          if (g == 0) { double* ur = &UF(un, i, 0); ur[0] = u0; ur[S] = u1; }
          store_state(xn, i + 1, s);
        };
        if (faithful) {
          // With CILQR_FLAG_FAITHFUL_ITERS a solve that has rejected keeps iterating without swapping its trajectory
          // buffers while its neighbours in the wavefront go on swapping theirs, so the groups no longer agree on which
          // buffer is current and the whole-wavefront chunk copy below (one source address for all) does not apply:
          // operands come straight from each group's own rows, one step ahead.
          load_fwd(fa, 0);
          int i = 0;
          for (; i + 1 < N; i += 2) {
            load_fwd(fb, i + 1);
            step(fa, i);
            load_fwd(fa, i + 2 < N ? i + 2 : i + 1);
            step(fb, i + 1);
          }
          if (i < N) step(fa, i);
        } else {
          // chunks of `chunk` steps, oldest first: X rows, U rows and gain rows of a chunk side by side in a staging buffer
          const double* wave_ws = ws_base + (size_t)blockIdx.x * L.rows() * S;
          const int buf_doubles = chunk * STAGE_ROWS * S;
          auto copy_chunk = [&](int lo, int hi, double* dst) {
            const int n = hi - lo + 1;
            stage_copy(wave_ws + (size_t)(xc + lo * XR) * S, dst, n * XR * S);
            stage_copy(wave_ws + (size_t)(uc + lo * 2) * S, dst + chunk * XR * S, n * 2 * S);
            stage_copy(wave_ws + (size_t)(L.kk() + lo * KR) * S, dst + chunk * (XR + 2) * S, n * KR * S);
          };
          auto lds_fwd = [&](FwdIn& o, const double* base, int t_rel) {
            const double* xr = base + (size_t)t_rel * XR * S + grp;
            const double* ur = base + (size_t)(chunk * XR + t_rel * 2) * S + grp;
            const double* kr = base + (size_t)(chunk * (XR + 2) + t_rel * KR) * S + grp;
            o.x = xr[0]; o.y = xr[S]; o.v = xr[2 * S]; o.th = xr[3 * S];
            o.u0 = ur[0]; o.u1 = ur[S];
  #pragma unroll
            for (int q = 0; q < KR; ++q) o.g[q] = kr[q * S];
          };
          int lo = 0, hi = chunk - 1 < N - 1 ? chunk - 1 : N - 1, buf = 0;
          copy_chunk(lo, hi, stage);
          stage_wait();
          while (lo < N) {
            const int nlo = hi + 1, nhi = nlo + chunk - 1 < N - 1 ? nlo + chunk - 1 : N - 1;
            if (nlo < N) copy_chunk(nlo, nhi, stage + (buf ^ 1) * buf_doubles);
            const double* cur = stage + buf * buf_doubles;
            int i = lo;
            lds_fwd(fa, cur, 0);
            for (; i + 1 <= hi; i += 2) {
              lds_fwd(fb, cur, i + 1 - lo);
              step(fa, i);
              lds_fwd(fa, cur, i + 2 <= hi ? i + 2 - lo : 0);
              step(fb, i + 1);
            }
            if (i == hi) step(fa, hi);
            stage_wait();
            lo = nlo;
            hi = nhi;
            buf ^= 1;
          }
        }
        if (!(max_th < 1.0e6)) { handover = true; break; }
      }
    }
    mem_sync();
    if (a.diag) tF += __builtin_readcyclecounter() - t0;

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

  if (!GENERAL) {
    if (g == 0) aq->redo[b] = handover ? 1 : 0;
    if (handover) return;
  }

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
      double Jpart = 0.0;
      for (int t = g; t < N; t += G) {
        const double px = XF(xc, t, 0), py = XF(xc, t, 1);
        const int cs = closest_sample(NS, grid, px, py, sample_at);
        double cx, cy;
        sample_xy(grid, pc, cs, cx, cy);
        Jpart += stage_cost(ae.kp, px - cx, py - cy, XF(xc, t, 2) - ae.kp.desired_speed, UF(uc, t, 0), UF(uc, t, 1));
      }
      J_new = group_sum<G>(Jpart);
    }
    if (g == 0) ae.J_out[b] = J_new;
  }
  if (a.diag && g == 0) {  // {prologue, L, R, F, epilogue, #L, #R, total}: prologue/epilogue not separated here
    unsigned long long* d = a.diag + (size_t)b * 8;
    d[0] = 0; d[1] = tL; d[2] = tR; d[3] = tF; d[4] = 0; d[5] = nL; d[6] = nR; d[7] = __builtin_readcyclecounter() - t_begin;
  }
  if (g == 0) {
    if (ae.iters_out) ae.iters_out[b] = iters;
    if (ae.status_out) ae.status_out[b] = status;
    if (ae.passes) ae.passes[b] = n_pass;
  }
#undef XF
#undef UF
#undef RF
#undef KF
}

// Steps per staging chunk: two buffers within 36 KiB of LDS per wavefront (four wavefronts per CU, one per SIMD).
template <int G>
int stage_chunk(int N) {
  constexpr int S = WAVE / G;
  int c = (36 * 1024) / (2 * STAGE_ROWS * S * (int)sizeof(double));
  if (c < 1) c = 1;
  return c < N ? c : N;
}

template <int G>
void launch_g(const SolveArgs& a, double* ws, hipStream_t stream) {
  constexpr int S = WAVE / G;
  const int blocks = (a.B + S - 1) / S;
  const int chunk = stage_chunk<G>(a.N);
  const size_t lds = (size_t)2 * chunk * STAGE_ROWS * S * sizeof(double);
  hipLaunchKernelGGL((cilqr_solve_groups_kernel<G, false>), dim3(blocks), dim3(WAVE), lds, stream, a, ws, chunk);
  hipLaunchKernelGGL((cilqr_solve_groups_kernel<G, true>), dim3(blocks), dim3(WAVE), 0, stream, a, ws, chunk);
}

}  // namespace

size_t solve_groups_ws_doubles(int B, int N) {
  // padded to whole wavefronts of the narrowest grouping (G = 1 → 64 solves per wavefront)
  const size_t Bp = ((size_t)B + WAVE - 1) / WAVE * WAVE;
  const WsLayout L{N, 0};
  return Bp * (size_t)L.rows();
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
