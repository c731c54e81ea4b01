// cilqr_solve.hip — batched constrained-iLQR solve for gfx950 (MI355X), one wavefront per solve.
//
// Hot path of the reference planner: iLQR::get_optimal_control_seq (I/iLQR.cpp:201-245) with everything it
// calls — nominal rollout (:51-62), Constraints::get_state_cost / get_control_cost / get_J
// (I/Constraints.cpp:145-227, 86-137, 534-561), Obstacle::get_obstalce_cost (I/Obstacle.cpp:39-112),
// Model::get_A_matrix / get_B_matrix (I/Model.cpp:100-155), the backward Riccati recursion (I/iLQR.cpp:133-191)
// and the forward pass (:68-86).  I/ = CILQR/src/ilqr/include/ilqr/ of the reference.
//
// Mapping: workgroup = one 64-lane wavefront = one solve, whole ≤20-iteration loop inside one
// launch.  Everything a solve touches between its first load and its last store lives in LDS:
//   samp  [S][2]        the S = 200 path samples (depend only on poly / x_local_plan, I/Constraints.cpp:28-42)
//   X a/b [(N+1)][6]    state records {x, y, v, theta, cos theta, sin theta}, double-buffered (X / X_new)
//   U a/b [N][2]        controls, double-buffered (U / U_new)
//   rec   [N][16]       per-step linearisation {l_x(3), l_xx(3), l_u(2), l_uu(2), A/B entries(6)}
//   kK    [N][10]       feed-forward k and feedback K of the backward pass
// Phases per iteration:
//   L  lanes = timesteps: closest path sample, tracking + obstacle + control barrier derivatives, A/B entries,
//      the stage cost of get_J, wavefront-shuffle reduction of J;
//   R  backward Riccati recursion, sequential in t, fp64 VALU in registers, per-step operands broadcast from LDS;
//   F  forward pass, sequential in t.
// The obstacle table (per obstacle and step: centre, heading cos/sin, 1/a², 1/b²) is built once per solve into a
// global workspace laid out [m][field][t] so that lanes = timesteps read it coalesced.
// No MFMA: the largest contraction is 4×4×4.
#include <float.h>

#include "cilqr_internal.h"

namespace cilqr {

namespace {

constexpr int WAVE = 64;
constexpr int XR = 6;    // doubles per state record
constexpr int REC = 16;  // doubles per linearisation record
constexpr int KR = 10;   // doubles per gain record
constexpr int TABF = 6;  // fields per obstacle-table entry

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

struct State {
  double x, y, v, th, c, s;
};

// Model::forward_simulate, I/Model.cpp:17-30 (the clamps act on a copy of the control, :19-20).
__device__ __forceinline__ State dyn_step(const KParams& kp, const State& st, double u0, double u1) {
  const double a = fmax(fmin(u0, kp.acc_max), kp.acc_min);
  const double hi = st.v * kp.tan_steer_max / kp.wheelbase;
  const double lo = st.v * kp.tan_steer_min / kp.wheelbase;
  const double w = fmax(fmin(u1, hi), lo);
  const double adv = st.v * kp.dt + a * kp.dt * kp.dt / 2.0;
  State n;
  n.x = st.x + st.c * adv;
  n.y = st.y + st.s * adv;
  n.v = fmin(fmax(st.v + a * kp.dt, 0.0), kp.speed_max);
  n.th = st.th + w * kp.dt;
  sincos(n.th, &n.s, &n.c);
  return n;
}

__device__ __forceinline__ void store_state(double* X, int t, const State& s) {
  double* r = X + t * XR;
  r[0] = s.x; r[1] = s.y; r[2] = s.v; r[3] = s.th; r[4] = s.c; r[5] = s.s;
}

// Closest path sample to (px, py): strict-< first minimum (I/Constraints.cpp:43-56).
__device__ __forceinline__ void closest_sample(const double* samp, int S, double px, double py, double& cx, double& cy) {
  double bx = samp[0], by = samp[1];
  double md = (bx - px) * (bx - px) + (by - py) * (by - py);
  for (int s = 0; s < S; ++s) {
    const double sx = samp[2 * s], sy = samp[2 * s + 1];
    const double d = (sx - px) * (sx - px) + (sy - py) * (sy - py);
    if (d < md) { md = d; bx = sx; by = sy; }
  }
  cx = bx;
  cy = by;
}

// Stage cost of Constraints::get_J (I/Constraints.cpp:534-561) for one step.
__device__ __forceinline__ double stage_cost(const KParams& kp, double dx, double dy, double dv, double u0, double u1) {
  const double xc = (dx * kp.w_pos) * dx + (dy * kp.w_pos) * dy + (dv * kp.w_vel) * dv;
  const double uc = (u0 * kp.w_acc) * u0 + (u1 * kp.w_yawrate) * u1;
  return xc + uc;
}

// Phase L.  Returns this lane's partial of J over its timesteps.
__device__ __forceinline__ double linearize(const KParams& kp, int N, int M, int lane, const double* samp, int S,
                                            const double* X, const double* U, double* rec, const double* tab,
                                            const double* wts) {
  double Jpart = 0.0;
  const double dt = kp.dt;
  for (int t = lane; t < N; t += WAVE) {
    const double* xr = X + t * XR;
    const double px = xr[0], py = xr[1], v = xr[2], ct = xr[4], st = xr[5];
    const double u0 = U[2 * t], u1 = U[2 * t + 1];

    // --- tracking cost (I/Constraints.cpp:163-174)
    double cx, cy;
    closest_sample(samp, S, px, py, cx, cy);
    const double dx = px - cx, dy = py - cy, dv = v - kp.desired_speed;
    double lx0 = (2 * kp.w_pos) * dx;
    double lx1 = (2 * kp.w_pos) * dy;
    const double lx2 = (2 * kp.w_vel) * dv;
    double h00 = kp.w_pos * 2, h01 = 0.0, h11 = kp.w_pos * 2;
    Jpart += stage_cost(kp, dx, dy, dv, u0, u1);

    // --- obstacles (I/Constraints.cpp:177-187, I/Obstacle.cpp:39-112)
    const double fxp = px + ct * kp.ego_front, fyp = py + st * kp.ego_front;
    const double rxp = px - ct * kp.ego_rear, ryp = py - st * kp.ego_rear;
    for (int m = 0; m < M; ++m) {
      const double* e = tab + (size_t)m * TABF * N + t;
      const double ox = e[0], oy = e[N], co = e[2 * N], so = e[3 * N], ia2 = e[4 * N], ib2 = e[5 * N];
      const double w = wts ? wts[m] : kp.w_obstacle;
      double gx = 0.0, gy = 0.0, gxx = 0.0, gxy = 0.0, gyy = 0.0;
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        const double ex = (side == 0 ? fxp : rxp) - ox, ey = (side == 0 ? fyp : ryp) - oy;
        const double q1 = side == 0 ? kp.q1_front : kp.q1_rear, q2 = side == 0 ? kp.q2_front : kp.q2_rear;
        const double d0 = co * ex + so * ey;
        const double d1 = co * ey - so * ex;
        const double g0 = d0 * ia2, g1 = d1 * ib2;
        const double c = 1 - (g0 * d0 + g1 * d1);
        const double cd0 = -2 * (co * g0 - so * g1);
        const double cd1 = -2 * (so * g0 + co * g1);
        const double ee = exp(q2 * c);
        const double sv = q2 * q1 * ee;
        const double sm = q2 * q2 * q1 * ee;
        gx += sv * cd0;
        gy += sv * cd1;
        gxx += (sm * cd0) * cd0;
        gxy += (sm * cd0) * cd1;
        gyy += (sm * cd1) * cd1;
      }
      lx0 += gx * w;
      lx1 += gy * w;
      h00 += gxx * w;
      h01 += gxy * w;
      h11 += gyy * w;
    }

    // --- control cost (I/Constraints.cpp:110-131)
    const double e1 = exp(kp.q2_acc * (u0 - kp.acc_max));
    const double e2 = exp(kp.q2_acc * (kp.acc_min - u0));
    const double e3 = exp(kp.q2_yawrate * (u1 - v * kp.tan_steer_max / kp.wheelbase));
    const double e4 = exp(kp.q2_yawrate * (v * kp.tan_steer_min / kp.wheelbase - u1));
    const double sa = kp.q2_acc * kp.q1_acc, sy = kp.q2_yawrate * kp.q1_yawrate;
    const double ma = kp.q2_acc * kp.q2_acc * kp.q1_acc, my = kp.q2_yawrate * kp.q2_yawrate * kp.q1_yawrate;
    const double lu0 = (sa * e1 - sa * e2) + (2 * kp.w_acc) * u0;
    const double lu1 = (sy * e3 - sy * e4) + (2 * kp.w_yawrate) * u1;
    const double luu0 = ma * e1 + ma * e2 + 2 * kp.w_acc;
    const double luu1 = my * e3 + my * e4 + 2 * kp.w_yawrate;

    // --- A/B entries at (v_{t+1}, theta_{t+1}, a_t) (I/iLQR.cpp:102-106, I/Model.cpp:100-155)
    const double* xn = X + (t + 1) * XR;
    const double vn = xn[2], cn = xn[4], sn = xn[5];
    const double adv = vn * dt + 0.5 * u0 * dt * dt;
    double* r = rec + t * REC;
    r[0] = lx0; r[1] = lx1; r[2] = lx2;
    r[3] = h00; r[4] = h01; r[5] = h11;
    r[6] = lu0; r[7] = lu1; r[8] = luu0; r[9] = luu1;
    r[10] = dt * cn;            // alpha: A(2,0)
    r[11] = dt * sn;            // beta : A(2,1)
    r[12] = (-1) * sn * adv;    // gamma: A(3,0)
    r[13] = cn * adv;           // delta: A(3,1)
    r[14] = dt * dt * cn / 2.0; // p    : B(0,0)
    r[15] = dt * dt * sn / 2.0; // q    : B(0,1)
  }
  return Jpart;
}

// get_J only (used once after an accepted last iteration).
__device__ __forceinline__ double cost_only(const KParams& kp, int N, int lane, const double* samp, int S,
                                            const double* X, const double* U) {
  double Jpart = 0.0;
  for (int t = lane; t < N; t += WAVE) {
    const double* xr = X + t * XR;
    double cx, cy;
    closest_sample(samp, S, xr[0], xr[1], cx, cy);
    Jpart += stage_cost(kp, xr[0] - cx, xr[1] - cy, xr[2] - kp.desired_speed, U[2 * t], U[2 * t + 1]);
  }
  return Jpart;
}

// Phase R: iLQR::backward_pass recursion (I/iLQR.cpp:108-191).  All lanes compute the same values.
// Returns false when Q_uu is not finite (the reference's EigenSolver path cannot produce a real
// decomposition there).
__device__ __forceinline__ bool riccati(const KParams& kp, int N, int lane, const double* rec, double* kK, double lamb) {
  const double dt = kp.dt;
  double Vx[4], V[4][4];
  {
    const double* r = rec + (N - 1) * REC;  // :108-113: terminal value = stage N-1
    Vx[0] = r[0]; Vx[1] = r[1]; Vx[2] = r[2]; Vx[3] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) V[i][j] = 0.0;
    V[0][0] = r[3]; V[0][1] = r[4]; V[1][0] = r[4]; V[1][1] = r[5];
    V[2][2] = kp.w_vel * 2;
  }
  bool ok = true;
  for (int j = N - 1; j >= 0; --j) {
    const double* r = rec + j * REC;
    const double lx0 = r[0], lx1 = r[1], lx2 = r[2], l00 = r[3], l01 = r[4], l11 = r[5];
    const double lu0 = r[6], lu1 = r[7], luu0 = r[8], luu1 = r[9];
    const double al = r[10], be = r[11], ga = r[12], de = r[13], p = r[14], q = r[15];

    // Q_x = l_x + fx*V_x ; Q_u = l_u + fu*V_x (:149-150); fx = [[1,0,0,0],[0,1,0,0],[al,be,1,0],[ga,de,0,1]],
    // fu = [[p,q,dt,0],[0,0,0,dt]] (the stored-transposed Jacobians).
    double Qx[4], Qu[2];
    Qx[0] = lx0 + Vx[0];
    Qx[1] = lx1 + Vx[1];
    Qx[2] = lx2 + (al * Vx[0] + be * Vx[1] + Vx[2]);
    Qx[3] = 0.0 + (ga * Vx[0] + de * Vx[1] + Vx[3]);
    Qu[0] = lu0 + (p * Vx[0] + q * Vx[1] + dt * Vx[2]);
    Qu[1] = lu1 + dt * Vx[3];

    // T = fx*V ; Q_xx = l_xx + T*fx' (:151)
    double T[4][4], Qxx[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      T[0][c] = V[0][c];
      T[1][c] = V[1][c];
      T[2][c] = al * V[0][c] + be * V[1][c] + V[2][c];
      T[3][c] = ga * V[0][c] + de * V[1][c] + V[3][c];
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      Qxx[rr][0] = T[rr][0];
      Qxx[rr][1] = T[rr][1];
      Qxx[rr][2] = al * T[rr][0] + be * T[rr][1] + T[rr][2];
      Qxx[rr][3] = ga * T[rr][0] + de * T[rr][1] + T[rr][3];
    }
    Qxx[0][0] += l00; Qxx[0][1] += l01; Qxx[1][0] += l01; Qxx[1][1] += l11;
    Qxx[2][2] += kp.w_vel * 2;

    // E = fu*V ; Q_ux = E*fx' ; Q_uu = l_uu + E*fu' (:152-153)
    double E[2][4], Qux[2][4], Quu[2][2];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      E[0][c] = p * V[0][c] + q * V[1][c] + dt * V[2][c];
      E[1][c] = dt * V[3][c];
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      Qux[rr][0] = E[rr][0];
      Qux[rr][1] = E[rr][1];
      Qux[rr][2] = al * E[rr][0] + be * E[rr][1] + E[rr][2];
      Qux[rr][3] = ga * E[rr][0] + de * E[rr][1] + E[rr][3];
      Quu[rr][0] = p * E[rr][0] + q * E[rr][1] + dt * E[rr][2];
      Quu[rr][1] = dt * E[rr][3];
    }
    Quu[0][0] += luu0;
    Quu[1][1] += luu1;

    // Regularised inverse V*diag(1/(max(eig,0)+lamb))*V' (:155-175) in closed form for the symmetric 2×2:
    // with h = (a-d)/2, rr = sqrt(h²+b²), c2 = h/rr, s2 = b/rr: inverse = (d1+d2)/2·I + (d1-d2)/2·[[c2,s2],[s2,-c2]].
    const double a = Quu[0][0], d = Quu[1][1], b = 0.5 * (Quu[0][1] + Quu[1][0]);
    if (!(a == a) || !(b == b) || !(d == d)) { ok = false; break; }
    double i00, i01, i11;
    {
      const double mm = 0.5 * (a + d), h = 0.5 * (a - d);
      const double rad = sqrt(h * h + b * b);
      const double e_hi = mm + rad, e_lo = mm - rad;
      const double d1 = 1.0 / (fmax(e_hi, 0.0) + lamb), d2 = 1.0 / (fmax(e_lo, 0.0) + lamb);
      const double hs = 0.5 * (d1 + d2), hd = 0.5 * (d1 - d2);
      double c2 = 1.0, s2 = 0.0;
      if (rad > 0.0) { c2 = h / rad; s2 = b / rad; }
      i00 = hs + hd * c2;
      i11 = hs - hd * c2;
      i01 = hd * s2;
    }

    // k = -Qinv*Q_u ; K = -Qinv*Q_ux (:177-178)
    double kj[2], Kj[2][4];
    kj[0] = -(i00 * Qu[0] + i01 * Qu[1]);
    kj[1] = -(i01 * Qu[0] + i11 * Qu[1]);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      Kj[0][c] = -(i00 * Qux[0][c] + i01 * Qux[1][c]);
      Kj[1][c] = -(i01 * Qux[0][c] + i11 * Qux[1][c]);
    }

    // G = K'*Q_uu (unregularised Q_uu) ; V_x = Q_x - G*k ; V_xx = Q_xx - G*K (:180-181)
    double G[4][2];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      G[rr][0] = Kj[0][rr] * Quu[0][0] + Kj[1][rr] * Quu[1][0];
      G[rr][1] = Kj[0][rr] * Quu[0][1] + Kj[1][rr] * Quu[1][1];
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      Vx[rr] = Qx[rr] - (G[rr][0] * kj[0] + G[rr][1] * kj[1]);
#pragma unroll
      for (int c = 0; c < 4; ++c) V[rr][c] = Qxx[rr][c] - (G[rr][0] * Kj[0][c] + G[rr][1] * Kj[1][c]);
    }

    if (lane == 0) {
      double* o = kK + j * KR;
      o[0] = kj[0]; o[1] = kj[1];
      o[2] = Kj[0][0]; o[3] = Kj[0][1]; o[4] = Kj[0][2]; o[5] = Kj[0][3];
      o[6] = Kj[1][0]; o[7] = Kj[1][1]; o[8] = Kj[1][2]; o[9] = Kj[1][3];
    }
  }
  return ok;
}

// Phase F: iLQR::forward_pass (I/iLQR.cpp:68-86).  All lanes compute the same values; lane 0 stores.
__device__ __forceinline__ void forward(const KParams& kp, int N, int lane, const double* X, const double* U,
                                        const double* kK, double* Xn, double* Un) {
  State s;
  s.x = X[0]; s.y = X[1]; s.v = X[2]; s.th = X[3]; s.c = X[4]; s.s = X[5];
  if (lane == 0) store_state(Xn, 0, s);
  for (int i = 0; i < N; ++i) {
    const double* xo = X + i * XR;
    const double* g = kK + i * KR;
    const double d0 = s.x - xo[0], d1 = s.y - xo[1], d2 = s.v - xo[2], d3 = s.th - xo[3];
    const double u0 = U[2 * i] + g[0] + (g[2] * d0 + g[3] * d1 + g[4] * d2 + g[5] * d3);
    const double u1 = U[2 * i + 1] + g[1] + (g[6] * d0 + g[7] * d1 + g[8] * d2 + g[9] * d3);
    s = dyn_step(kp, s, u0, u1);
    if (lane == 0) {
      Un[2 * i] = u0;
      Un[2 * i + 1] = u1;
      store_state(Xn, i + 1, s);
    }
  }
}

__global__ __launch_bounds__(WAVE) void cilqr_solve_kernel(SolveArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const KParams& kp = a.kp;
  const int N = a.N, M = a.M, S = kp.n_samples;
  if (b >= a.B) return;

  double* samp = lds;
  double* Xa = samp + 2 * S;
  double* Xb = Xa + (N + 1) * XR;
  double* Ua = Xb + (N + 1) * XR;
  double* Ub = Ua + 2 * N;
  double* rec = Ub + 2 * N;
  double* kK = rec + N * REC;

  // ---- prologue -------------------------------------------------------------------------------------------
  {  // path samples, I/Constraints.cpp:28-42 (ascending powers by repeated multiplication)
    const double* pc = a.poly + (size_t)b * CILQR_POLY_COEFFS;
    const double xf = a.xplan_fl[2 * b], xl = a.xplan_fl[2 * b + 1];
    const double dxs = (xl - xf) / (double)S;
    for (int s = lane; s < S; s += WAVE) {
      const double x = xf + dxs * s;
      double y = 0.0, pw = 1.0;
#pragma unroll
      for (int j = 0; j < CILQR_POLY_COEFFS; ++j) {
        y += pc[j] * pw;
        pw *= x;
      }
      samp[2 * s] = x;
      samp[2 * s + 1] = y;
    }
  }
  double* Ug = a.U + (size_t)b * 2 * N;
  for (int i = lane; i < 2 * N; i += WAVE) Ua[i] = Ug[i];

  double* tab = a.obs_tab + (size_t)b * M * TABF * N;
  const double* wts = a.obs_weight ? a.obs_weight + (size_t)b * M : nullptr;
  for (int m = 0; m < M; ++m) {  // obstacle table, I/Obstacle.cpp:41-62
    for (int t = lane; t < N; t += WAVE) {
      const double* ps = a.obs_pose + (((size_t)b * M + m) * N + t) * 4;
      const double* dm = a.obs_dim + (((size_t)b * M + m) * N + t) * 2;
      double so, co;
      sincos(ps[3], &so, &co);
      const double ea = dm[0] / 2.0 + fabs(ps[2] * co) * kp.t_safe + kp.s_safe_a + kp.ego_rad;
      const double eb = dm[1] / 2.0 + fabs(ps[2] * so) * kp.t_safe + kp.s_safe_b + kp.ego_rad + 1;
      double* e = tab + (size_t)m * TABF * N + t;
      e[0] = ps[0];
      e[N] = ps[1];
      e[2 * N] = co;
      e[3 * N] = so;
      e[4 * N] = 1.0 / ea / ea;
      e[5 * N] = 1.0 / eb / eb;
    }
  }
  __syncthreads();

  {  // nominal rollout, I/iLQR.cpp:51-62
    const double* x0 = a.x0 + (size_t)b * 4;
    State s;
    s.x = x0[0]; s.y = x0[1]; s.v = x0[2]; s.th = x0[3];
    sincos(s.th, &s.s, &s.c);
    if (lane == 0) store_state(Xa, 0, s);
    for (int i = 0; i < N; ++i) {
      s = dyn_step(kp, s, Ua[2 * i], Ua[2 * i + 1]);
      if (lane == 0) store_state(Xa, i + 1, s);
    }
  }
  __syncthreads();

  // ---- iteration loop, I/iLQR.cpp:204-239 --------------------------------------------------------------------
  double* Xc = Xa;
  double* Uc = Ua;
  double* Xn = Xb;
  double* Un = Ub;
  double J_old = DBL_MAX, lamb = 1.0, J_new = 0.0;
  int iters = 0, status = CILQR_EXIT_MAX_ITER;
  bool j_valid = false;  // J_new is get_J of the current (Xc, Uc)
  const bool faithful = (a.flags & CILQR_FLAG_FAITHFUL_ITERS) != 0;
  const int max_it = kp.max_iterations;
  for (int it = 0; it < max_it; ++it) {
    ++iters;
    // The reference evaluates backward_pass, forward_pass, then J_new = get_J(X, U) on the CURRENT X, U (:213-217).
    // The linearisation and J share their closest-point searches, so they are computed together, first.
    J_new = readfirstlane_f64(wave_sum(linearize(kp, N, M, lane, samp, S, Xc, Uc, rec, tab, wts)));
    j_valid = true;
    __syncthreads();
    const bool accept = J_new < J_old;
    if (!accept && !faithful) {
      // A rejection leaves X, U untouched, so every later iteration recomputes the same J_new == J_old and
      // rejects again until lamb > lamb_max or the iteration cap: only lamb and the counter change.
      for (;;) {
        lamb = lamb * kp.lamb_factor;
        if (lamb > kp.lamb_max) { status = CILQR_EXIT_LAMBDA_MAX; break; }
        if (++it >= max_it) { status = CILQR_EXIT_MAX_ITER; break; }
        ++iters;
      }
      break;
    }
    if (!riccati(kp, N, lane, rec, kK, lamb)) { status = CILQR_EXIT_NUMERIC; break; }
    __syncthreads();
    forward(kp, N, lane, Xc, Uc, kK, Xn, Un);
    __syncthreads();
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

  // ---- epilogue: X_result / U_result (:243-244) ----------------------------------------------------------------
  for (int i = lane; i < 2 * N; i += WAVE) Ug[i] = Uc[i];
  double* Xg = a.X_out + (size_t)b * 4 * (N + 1);
  for (int i = lane; i < 4 * (N + 1); i += WAVE) Xg[i] = Xc[(i >> 2) * XR + (i & 3)];
  if (a.J_out) {
    if (!j_valid) J_new = readfirstlane_f64(wave_sum(cost_only(kp, N, lane, samp, S, Xc, Uc)));
    if (lane == 0) a.J_out[b] = J_new;
  }
  if (lane == 0) {
    if (a.iters_out) a.iters_out[b] = iters;
    if (a.status_out) a.status_out[b] = status;
  }
}

}  // namespace

size_t solve_lds_bytes(int N, int n_samples) {
  const size_t doubles = 2 * (size_t)n_samples + 2 * (size_t)(N + 1) * XR + 2 * (size_t)2 * N + (size_t)N * REC + (size_t)N * KR;
  return doubles * sizeof(double);
}

hipError_t launch_solve(const SolveArgs& a, hipStream_t stream) {
  if (a.B <= 0) return hipSuccess;
  const size_t lds = solve_lds_bytes(a.N, a.kp.n_samples);
  hipLaunchKernelGGL(cilqr_solve_kernel, dim3(a.B), dim3(WAVE), lds, stream, a);
  return hipGetLastError();
}

}  // namespace cilqr
