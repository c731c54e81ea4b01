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

// sin and cos of one fp64 argument, ≤ ~1 ulp each: three-part Cody–Waite reduction by pi/2 (exact first step under fma
// for |x| < 2^20·pi/2) followed by the classic degree-13 / degree-14 minimax kernels on [-pi/4, pi/4].  Larger arguments
// (never met by a heading angle) take the library path.  The serial forward pass spends most of its dependent chain
// here, so the ~35 instructions of this form (against ~160 for the library's general-range sincos) are what bounds a step.
__device__ __forceinline__ void sincos_fast(double x, double* sn, double* cs) {
  if (__builtin_expect(!(fabs(x) < 1.0e6), 0)) {
    sincos(x, sn, cs);
    return;
  }
  const double n = rint(x * 6.36619772367581382433e-01);  // 2/pi
  double r = fma(-n, 1.57079632679489655800e+00, x);      // pi/2 head: exact
  r = fma(-n, 6.12323399573676603587e-17, r);             // pi/2 - head
  r = fma(-n, -1.49738490485916983278e-33, r);            // next part
  const double z = r * r;
  // sin kernel
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                               2.75573137070700676789e-06), -1.98412698298579493134e-04),
                               8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double sr = fma(z * r, ps, r);
  // cos kernel
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                               -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                               -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  const double cr = w + (((1.0 - w) - hz) + z * (z * pc));
  const int q = (int)n;
  const double s0 = (q & 1) ? cr : sr;
  const double c0 = (q & 1) ? sr : cr;
  *sn = (q & 2) ? -s0 : s0;
  *cs = ((q + 1) & 2) ? -c0 : c0;
}

// Model::forward_simulate, I/Model.cpp:17-30 (the clamps act on a copy of the control, :19-20).  The yaw-rate bounds
// v·tan(steer)/wheelbase are taken as v·(tan(steer)/wheelbase) with the quotient formed once on the host.
__device__ __forceinline__ State dyn_step(const KParams& kp, const State& st, double u0, double u1) {
  const double a = fmax(fmin(u0, kp.acc_max), kp.acc_min);
  const double w = fmax(fmin(u1, st.v * kp.yaw_hi), st.v * kp.yaw_lo);
  const double adv = st.v * kp.dt + a * kp.half_dt2;
  State n;
  n.x = st.x + st.c * adv;
  n.y = st.y + st.s * adv;
  n.v = fmin(fmax(st.v + a * kp.dt, 0.0), kp.speed_max);
  n.th = st.th + w * kp.dt;
  sincos_fast(n.th, &n.s, &n.c);
  return n;
}

__device__ __forceinline__ void store_state(double* X, int t, const State& s) {
  double* r = X + t * XR;
  r[0] = s.x; r[1] = s.y; r[2] = s.v; r[3] = s.th; r[4] = s.c; r[5] = s.s;
}

// Uniform (per solve) description of the sample abscissae x_s = xf + dxs*s, for the windowed search below.
struct SampleGrid {
  double xf, inv_dxs;  // inv_dxs = 1/dxs (signed)
  bool windowed;       // false: dxs is 0 or not finite → full scan
};

__device__ __forceinline__ double sample_dist(const double* samp, int s, double px, double py) {
  const double sx = samp[2 * s], sy = samp[2 * s + 1];
  return (sx - px) * (sx - px) + (sy - py) * (sy - py);
}

// Closest path sample to (px, py): index of the strict-< first minimum of the squared distance over ALL S samples
// (I/Constraints.cpp:43-56), found without visiting all of them.  A sample can only reach the distance d_c of the
// sample nearest in x if its own x-offset satisfies (x_s - px)² ≤ d_c, because fl(dx² + dy²) ≥ fl(dx²); the x_s are
// equispaced, so that is an index window around (px - xf)/dxs.  The window is widened by two samples and a 1e-4
// relative margin (≫ any rounding in its own computation), clamped to [0, S-1] and scanned in ascending order with
// strict <, which yields exactly the reference's argmin, ties included.
__device__ __forceinline__ int closest_sample(const double* samp, int S, const SampleGrid& g, double px, double py) {
  int lo = 0, hi = S - 1;
  if (g.windowed) {
    const double fc = (px - g.xf) * g.inv_dxs;
    const double fcc = fmin(fmax(fc, 0.0), (double)(S - 1));  // NaN → 0
    const double dc = sample_dist(samp, (int)(fcc + 0.5), px, py);
    const double hw = (double)(__builtin_sqrtf((float)dc) * 1.0001f) * fabs(g.inv_dxs) * 1.0001 + 2.0;
    if (hw < 1.0e9) {  // false for NaN / overflow: keep the full range
      lo = (int)fmin(fmax(fc - hw, 0.0), (double)(S - 1));
      hi = (int)fmax(fmin(fc + hw + 1.0, (double)(S - 1)), 0.0);
    }
  }
  double md = sample_dist(samp, lo, px, py);
  int best = lo;
  for (int s = lo + 1; s <= hi; s += 4) {
    // four candidates per trip (indices past hi repeat hi: harmless under strict <), loads issued together
    const int s1 = min(s + 1, hi), s2 = min(s + 2, hi), s3 = min(s + 3, hi);
    const double d0 = sample_dist(samp, s, px, py), d1 = sample_dist(samp, s1, px, py);
    const double d2 = sample_dist(samp, s2, px, py), d3 = sample_dist(samp, s3, px, py);
    if (d0 < md) { md = d0; best = s; }
    if (d1 < md) { md = d1; best = s1; }
    if (d2 < md) { md = d2; best = s2; }
    if (d3 < md) { md = d3; best = s3; }
  }
  return best;
}

// Stage cost of Constraints::get_J (I/Constraints.cpp:534-561) for one step.
__device__ __forceinline__ double stage_cost(const KParams& kp, double dx, double dy, double dv, double u0, double u1) {
  const double xc = (dx * kp.w_pos) * dx + (dy * kp.w_pos) * dy + (dv * kp.w_vel) * dv;
  const double uc = (u0 * kp.w_acc) * u0 + (u1 * kp.w_yawrate) * u1;
  return xc + uc;
}

// Phase L.  Returns this lane's partial of J over its timesteps.
__device__ __forceinline__ double linearize(const KParams& kp, int N, int M, int lane, const double* samp, int S,
                                            const SampleGrid& grid, const double* X, const double* U, double* rec,
                                            const double* tab, const double* wts) {
  double Jpart = 0.0;
  const double dt = kp.dt;
  for (int t = lane; t < N; t += WAVE) {
    const double* xr = X + t * XR;
    const double px = xr[0], py = xr[1], v = xr[2], ct = xr[4], st = xr[5];
    const double u0 = U[2 * t], u1 = U[2 * t + 1];

    // --- tracking cost (I/Constraints.cpp:163-174)
    const int cs = closest_sample(samp, S, grid, px, py);
    const double cx = samp[2 * cs], cy = samp[2 * cs + 1];
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
    const double e3 = exp(kp.q2_yawrate * (u1 - v * kp.yaw_hi));
    const double e4 = exp(kp.q2_yawrate * (v * kp.yaw_lo - u1));
    const double sa = kp.q2_acc * kp.q1_acc, sy = kp.q2_yawrate * kp.q1_yawrate;
    const double ma = kp.q2_acc * kp.q2_acc * kp.q1_acc, my = kp.q2_yawrate * kp.q2_yawrate * kp.q1_yawrate;
    const double lu0 = (sa * e1 - sa * e2) + (2 * kp.w_acc) * u0;
    const double lu1 = (sy * e3 - sy * e4) + (2 * kp.w_yawrate) * u1;
    const double luu0 = ma * e1 + ma * e2 + 2 * kp.w_acc;
    const double luu1 = my * e3 + my * e4 + 2 * kp.w_yawrate;

    // --- A/B entries at (v_{t+1}, theta_{t+1}, a_t) (I/iLQR.cpp:102-106, I/Model.cpp:100-155)
    const double* xn = X + (t + 1) * XR;
    const double vn = xn[2], cn = xn[4], sn = xn[5];
    const double adv = vn * dt + u0 * kp.half_dt2;
    double* r = rec + t * REC;
    r[0] = lx0; r[1] = lx1; r[2] = lx2;
    r[3] = h00; r[4] = h01; r[5] = h11;
    r[6] = lu0; r[7] = lu1; r[8] = luu0; r[9] = luu1;
    r[10] = dt * cn;            // alpha: A(2,0)
    r[11] = dt * sn;            // beta : A(2,1)
    r[12] = (-1) * sn * adv;    // gamma: A(3,0)
    r[13] = cn * adv;           // delta: A(3,1)
    r[14] = kp.half_dt2 * cn;   // p    : B(0,0)
    r[15] = kp.half_dt2 * sn;   // q    : B(0,1)
  }
  return Jpart;
}

// get_J only (used once after an accepted last iteration).
__device__ __forceinline__ double cost_only(const KParams& kp, int N, int lane, const double* samp, int S,
                                            const SampleGrid& grid, const double* X, const double* U) {
  double Jpart = 0.0;
  for (int t = lane; t < N; t += WAVE) {
    const double* xr = X + t * XR;
    const int cs = closest_sample(samp, S, grid, xr[0], xr[1]);
    const double cx = samp[2 * cs], cy = samp[2 * cs + 1];
    Jpart += stage_cost(kp, xr[0] - cx, xr[1] - cy, xr[2] - kp.desired_speed, U[2 * t], U[2 * t + 1]);
  }
  return Jpart;
}

// ---- single-instruction helpers -----------------------------------------------------------------------------------
// One wavefront per SIMD issues one instruction every ~4.5 cycles whatever its kind (measured, tools/ubench_fp64.hip),
// so the serial phases are priced in instructions.  These keep hipcc from adding canonicalising v_max around
// fmin/fmax and from re-materialising 64-bit literals with s_mov pairs inside the loops.
__device__ __forceinline__ double vmin(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double vmax(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// Pins a loop-invariant value in a vector register: after this the compiler cannot fold it back into a literal.
#define CILQR_PIN(x) asm volatile("" : "+v"(x))

// 1/x by v_rcp_f64 and two Newton steps (≤ ~1 ulp; x is a well-scaled positive determinant here).
__device__ __forceinline__ double rcp_newton(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// Regularised inverse V diag(1/(max(eig,0)+lamb)) V' of the symmetric 2×2 [[a,b],[b,d]] (I/iLQR.cpp:155-175).
// PSD case (always, when l_xx and l_uu are PSD: barrier Hessians are rank-one PSD): both eigenvalues pass the clamp and
// the result is inv(Q_uu + lamb I), formed from the adjugate with one reciprocal.  bb = b².
__device__ __forceinline__ void quu_inverse_psd(double a, double b, double d, double lamb, double bb, double& i00, double& i01,
                                                double& i11) {
  const double ar = a + lamb, dr = d + lamb;
  const double rdet = rcp_newton(fma(ar, dr, -bb));
  i00 = dr * rdet;
  i11 = ar * rdet;
  i01 = -b * rdet;
}
// General case.  With m = (a+d)/2, h = (a-d)/2, r = sqrt(h²+b²) the eigenvalues are m ± r and
// inverse = (d1+d2)/2·I + (d1-d2)/2·[[h,b],[b,-h]]/r, d_i = 1/(max(eig_i,0)+lamb).  False for a non-finite matrix.
__device__ __forceinline__ bool quu_inverse_general(double a, double b, double d, double lamb, double& i00, double& i01,
                                                    double& i11) {
  const double bb = b * b;
  const double det0 = fma(a, d, -bb);
  if (!(det0 == det0) || !(a + d == a + d)) return false;
  if (det0 >= 0.0 && a + d >= 0.0) {
    quu_inverse_psd(a, b, d, lamb, bb, i00, i01, i11);
    return true;
  }
  const double mm = 0.5 * (a + d), h = 0.5 * (a - d);
  const double rad = sqrt(fma(h, h, bb));
  const double d1 = 1.0 / (fmax(mm + rad, 0.0) + lamb), d2 = 1.0 / (fmax(mm - rad, 0.0) + lamb);
  const double hs = 0.5 * (d1 + d2), hd = 0.5 * (d1 - d2);
  double c2 = 1.0, s2 = 0.0;
  if (rad > 0.0) { c2 = h / rad; s2 = b / rad; }
  i00 = fma(hd, c2, hs);
  i11 = fma(-hd, c2, hs);
  i01 = hd * s2;
  return true;
}

// One per-step linearisation record held in registers.
struct Rec {
  double lx0, lx1, lx2, l00, l01, l11, lu0, lu1, luu0, luu1, al, be, ga, de, p, q;
};
__device__ __forceinline__ void load_rec(Rec& o, const double* rec, int j) {
  const double* r = rec + j * REC;
  o.lx0 = r[0]; o.lx1 = r[1]; o.lx2 = r[2]; o.l00 = r[3]; o.l01 = r[4]; o.l11 = r[5];
  o.lu0 = r[6]; o.lu1 = r[7]; o.luu0 = r[8]; o.luu1 = r[9];
  o.al = r[10]; o.be = r[11]; o.ga = r[12]; o.de = r[13]; o.p = r[14]; o.q = r[15];
}

// Value function carried by the backward recursion: V_x and the upper triangle of the symmetric V_xx.
struct Value {
  double x0, x1, x2, x3;
  double v00, v01, v02, v03, v11, v12, v13, v22, v23, v33;
};

// One step of iLQR::backward_pass (I/iLQR.cpp:133-191).
//
// With fx = [[1,0,0,0],[0,1,0,0],[al,be,1,0],[ga,de,0,1]] and fu = [[p,q,dt,0],[0,0,0,dt]] (the reference's
// stored-transposed Jacobians, I/Model.cpp:100-155) the products of :149-153 reduce to one or two fused multiply-adds
// per entry.  V_xx, Q_xx and Q_uu are carried as symmetric matrices (the reference computes both triangles, which
// agree to rounding).  Returns false when Q_uu is not finite (the reference's EigenSolver cannot give a real
// decomposition there).
// FAST: branch-free positive-semi-definite form; steps whose Q_uu fails the test (or is NaN) are OR-ed into `suspect`
// (a wavefront-uniform lane mask) and the caller redoes the whole pass with FAST = false, which handles them.
template <bool FAST>
__device__ __forceinline__ bool riccati_step(const Rec& c, Value& V, double dt, double two_wvel, double lamb, double* out,
                                             unsigned long long& suspect) {
  const double al = c.al, be = c.be, ga = c.ga, de = c.de, p = c.p, q = c.q;
  const double x0 = V.x0, x1 = V.x1, x2 = V.x2, x3 = V.x3;
  const double v00 = V.v00, v01 = V.v01, v02 = V.v02, v03 = V.v03, v11 = V.v11, v12 = V.v12, v13 = V.v13;
  const double v22 = V.v22, v23 = V.v23, v33 = V.v33;

  // Q_x = l_x + fx V_x ; Q_u = l_u + fu V_x (:149-150)
  const double qx0 = c.lx0 + x0;
  const double qx1 = c.lx1 + x1;
  const double qx2 = fma(al, x0, fma(be, x1, x2 + c.lx2));
  const double qx3 = fma(ga, x0, fma(de, x1, x3));
  const double qu0 = fma(p, x0, fma(q, x1, fma(dt, x2, c.lu0)));
  const double qu1 = fma(dt, x3, c.lu1);

  // T = fx V (rows 2, 3) ; Q_xx = l_xx + T fx' (:151)
  const double t20 = fma(al, v00, fma(be, v01, v02));
  const double t21 = fma(al, v01, fma(be, v11, v12));
  const double t22 = fma(al, v02, fma(be, v12, v22));
  const double t23 = fma(al, v03, fma(be, v13, v23));
  const double t30 = fma(ga, v00, fma(de, v01, v03));
  const double t31 = fma(ga, v01, fma(de, v11, v13));
  const double t33 = fma(ga, v03, fma(de, v13, v33));
  const double q00 = v00 + c.l00, q01 = v01 + c.l01, q11 = v11 + c.l11;
  const double q22 = fma(al, t20, fma(be, t21, t22 + two_wvel));
  const double q23 = fma(ga, t20, fma(de, t21, t23));
  const double q33 = fma(ga, t30, fma(de, t31, t33));

  // E = fu V ; Q_ux = E fx' ; Q_uu = l_uu + E fu' (:152-153)
  const double e00 = fma(p, v00, fma(q, v01, dt * v02));
  const double e01 = fma(p, v01, fma(q, v11, dt * v12));
  const double e02 = fma(p, v02, fma(q, v12, dt * v22));
  const double e03 = fma(p, v03, fma(q, v13, dt * v23));
  const double e10 = dt * v03, e11 = dt * v13, e12 = dt * v23, e13 = dt * v33;
  const double ux02 = fma(al, e00, fma(be, e01, e02));
  const double ux03 = fma(ga, e00, fma(de, e01, e03));
  const double ux12 = fma(al, e10, fma(be, e11, e12));
  const double ux13 = fma(ga, e10, fma(de, e11, e13));
  const double a = fma(p, e00, fma(q, e01, fma(dt, e02, c.luu0)));
  const double b = dt * e03;
  const double d = fma(dt, e13, c.luu1);

  // Regularised inverse V diag(1/(max(eig,0)+lamb)) V' (:155-175).  Positive semi-definite Q_uu (always, when l_xx
  // and l_uu are: barrier Hessians are rank-one PSD): both eigenvalues pass the clamp and the result is
  // inv(Q_uu + lamb I), formed from the adjugate with one reciprocal.  Otherwise the clamped eigen form:
  // with h = (a-d)/2, r = sqrt(h²+b²): inverse = (d1+d2)/2·I + (d1-d2)/2·[[h,b],[b,-h]]/r.
  double i00, i01, i11;
  if (FAST) {
    const double bb = b * b;
    const double det0 = fma(a, d, -bb);
    suspect |= __builtin_amdgcn_ballot_w64(!(det0 >= 0.0)) | __builtin_amdgcn_ballot_w64(!(a + d >= 0.0));
    quu_inverse_psd(a, b, d, lamb, bb, i00, i01, i11);
  } else if (!quu_inverse_general(a, b, d, lamb, i00, i01, i11)) {
    return false;
  }

  // k = -Qinv Q_u ; K = -Qinv Q_ux (:177-178)
  const double k0 = fma(-i00, qu0, -(i01 * qu1));
  const double k1 = fma(-i01, qu0, -(i11 * qu1));
  const double K00 = fma(-i00, e00, -(i01 * e10)), K01 = fma(-i00, e01, -(i01 * e11));
  const double K02 = fma(-i00, ux02, -(i01 * ux12)), K03 = fma(-i00, ux03, -(i01 * ux13));
  const double K10 = fma(-i01, e00, -(i11 * e10)), K11 = fma(-i01, e01, -(i11 * e11));
  const double K12 = fma(-i01, ux02, -(i11 * ux12)), K13 = fma(-i01, ux03, -(i11 * ux13));

  // G = K' Q_uu (unregularised) ; V_x = Q_x - G k ; V_xx = Q_xx - G K (:180-181)
  const double g00 = fma(K00, a, K10 * b), g01 = fma(K00, b, K10 * d);
  const double g10 = fma(K01, a, K11 * b), g11 = fma(K01, b, K11 * d);
  const double g20 = fma(K02, a, K12 * b), g21 = fma(K02, b, K12 * d);
  const double g30 = fma(K03, a, K13 * b), g31 = fma(K03, b, K13 * d);
  V.x0 = fma(-g01, k1, fma(-g00, k0, qx0));
  V.x1 = fma(-g11, k1, fma(-g10, k0, qx1));
  V.x2 = fma(-g21, k1, fma(-g20, k0, qx2));
  V.x3 = fma(-g31, k1, fma(-g30, k0, qx3));
  V.v00 = fma(-g01, K10, fma(-g00, K00, q00));
  V.v01 = fma(-g01, K11, fma(-g00, K01, q01));
  V.v02 = fma(-g01, K12, fma(-g00, K02, t20));
  V.v03 = fma(-g01, K13, fma(-g00, K03, t30));
  V.v11 = fma(-g11, K11, fma(-g10, K01, q11));
  V.v12 = fma(-g11, K12, fma(-g10, K02, t21));
  V.v13 = fma(-g11, K13, fma(-g10, K03, t31));
  V.v22 = fma(-g21, K12, fma(-g20, K02, q22));
  V.v23 = fma(-g21, K13, fma(-g20, K03, q23));
  V.v33 = fma(-g31, K13, fma(-g30, K03, q33));

  if (threadIdx.x == 0) {  // every lane holds the same values; one lane stores (same-address stores from 64 lanes serialise)
    out[0] = k0; out[1] = k1;
    out[2] = K00; out[3] = K01; out[4] = K02; out[5] = K03;
    out[6] = K10; out[7] = K11; out[8] = K12; out[9] = K13;
  }
  return true;
}

// Phase R: iLQR::backward_pass recursion (I/iLQR.cpp:108-191).  All lanes compute the same values; operands are
// broadcast LDS reads issued one step ahead into the register set the next step uses (two steps per trip, no copies).
template <bool FAST>
__device__ __forceinline__ bool riccati_pass(const KParams& kp, int N, const double* rec, double* kK, double lamb_in,
                                             unsigned long long& suspect) {
  double dt = kp.dt, two_wvel = kp.w_vel * 2, lamb = lamb_in;
  CILQR_PIN(dt); CILQR_PIN(two_wvel); CILQR_PIN(lamb);
  Rec ra, rb;
  load_rec(ra, rec, N - 1);
  Value V;  // :108-113: terminal value = stage N-1
  V.x0 = ra.lx0; V.x1 = ra.lx1; V.x2 = ra.lx2; V.x3 = 0.0;
  V.v00 = ra.l00; V.v01 = ra.l01; V.v02 = 0.0; V.v03 = 0.0; V.v11 = ra.l11; V.v12 = 0.0; V.v13 = 0.0;
  V.v22 = two_wvel; V.v23 = 0.0; V.v33 = 0.0;
  int j = N - 1;
  for (; j >= 1; j -= 2) {
    load_rec(rb, rec, j - 1);
    if (!riccati_step<FAST>(ra, V, dt, two_wvel, lamb, kK + j * KR, suspect)) return false;
    load_rec(ra, rec, j >= 2 ? j - 2 : 0);
    if (!riccati_step<FAST>(rb, V, dt, two_wvel, lamb, kK + (j - 1) * KR, suspect)) return false;
  }
  if (j == 0 && !riccati_step<FAST>(ra, V, dt, two_wvel, lamb, kK, suspect)) return false;
  return true;
}

// GENERAL = false: the branch-free fast pass; returns false when any step was suspect (the caller then hands the
// solve to the GENERAL kernel).  GENERAL = true: the branching pass; returns false only for a non-finite Q_uu.
template <bool GENERAL>
__device__ __forceinline__ bool riccati(const KParams& kp, int N, const double* rec, double* kK, double lamb) {
  unsigned long long suspect = 0;
  const bool ok = riccati_pass<!GENERAL>(kp, N, rec, kK, lamb, suspect);
  return GENERAL ? ok : suspect == 0;
}

// ---- forward pass ---------------------------------------------------------------------------------------------------
struct FwdConst {  // loop invariants of the forward pass, pinned in vector registers
  double dt, half_dt2, acc_max, acc_min, yaw_hi, yaw_lo, speed_max, zero;
  double two_over_pi, p1, p2, p3, s1, s2, s3, s4, s5, s6, c1, c2, c3, c4, c5, c6;
};
__device__ __forceinline__ void make_fwd_const(FwdConst& k, const KParams& kp) {
  k.dt = kp.dt; k.half_dt2 = kp.half_dt2; k.acc_max = kp.acc_max; k.acc_min = kp.acc_min;
  k.yaw_hi = kp.yaw_hi; k.yaw_lo = kp.yaw_lo; k.speed_max = kp.speed_max; k.zero = 0.0;
  k.two_over_pi = 6.36619772367581382433e-01;
  k.p1 = 1.57079632679489655800e+00; k.p2 = 6.12323399573676603587e-17; k.p3 = -1.49738490485916983278e-33;
  k.s1 = -1.66666666666666324348e-01; k.s2 = 8.33333333332248946124e-03; k.s3 = -1.98412698298579493134e-04;
  k.s4 = 2.75573137070700676789e-06; k.s5 = -2.50507602534068634195e-08; k.s6 = 1.58969099521155010221e-10;
  k.c1 = 4.16666666666666019037e-02; k.c2 = -1.38888888888741095749e-03; k.c3 = 2.48015872894767294178e-05;
  k.c4 = -2.75573143513906633035e-07; k.c5 = 2.08757232129817482790e-09; k.c6 = -1.13596475577881948265e-11;
  CILQR_PIN(k.dt); CILQR_PIN(k.half_dt2); CILQR_PIN(k.acc_max); CILQR_PIN(k.acc_min); CILQR_PIN(k.yaw_hi);
  CILQR_PIN(k.yaw_lo); CILQR_PIN(k.speed_max); CILQR_PIN(k.zero); CILQR_PIN(k.two_over_pi);
  CILQR_PIN(k.p1); CILQR_PIN(k.p2); CILQR_PIN(k.p3);
  CILQR_PIN(k.s1); CILQR_PIN(k.s2); CILQR_PIN(k.s3); CILQR_PIN(k.s4); CILQR_PIN(k.s5); CILQR_PIN(k.s6);
  CILQR_PIN(k.c1); CILQR_PIN(k.c2); CILQR_PIN(k.c3); CILQR_PIN(k.c4); CILQR_PIN(k.c5); CILQR_PIN(k.c6);
}

// sincos_fast without its range guard and with every constant in a register (same arithmetic, same results for
// |x| < 1e6; the caller tracks max|x| and redoes the pass on the guarded path if that bound was ever exceeded).
__device__ __forceinline__ void sincos_loop(const FwdConst& k, double x, double& sn, double& cs) {
  const double n = rint(x * k.two_over_pi);
  double r = fma(-n, k.p1, x);
  r = fma(-n, k.p2, r);
  r = fma(-n, k.p3, r);
  const double z = r * r;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, k.s6, k.s5), k.s4), k.s3), k.s2), k.s1);
  const double sr = fma(z * r, ps, r);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, k.c6, k.c5), k.c4), k.c3), k.c2), k.c1);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  const double cr = w + (((1.0 - w) - hz) + z * (z * pc));
  const int q = (int)n;
  const bool odd = (q & 1) != 0;
  const double s0 = odd ? cr : sr;
  const double c0 = odd ? sr : cr;
  // sign flips as integer xors on the high words
  const int sgs = (q & 2) << 30, sgc = ((q + 1) & 2) << 30;
  sn = __hiloint2double(__double2hiint(s0) ^ sgs, __double2loint(s0));
  cs = __hiloint2double(__double2hiint(c0) ^ sgc, __double2loint(c0));
}

struct FwdIn {
  double x, y, v, th, u0, u1, g[KR];
};
__device__ __forceinline__ void load_fwd(FwdIn& o, const double* X, const double* U, const double* kK, int i) {
  const double* xo = X + i * XR;
  o.x = xo[0]; o.y = xo[1]; o.v = xo[2]; o.th = xo[3];
  o.u0 = U[2 * i]; o.u1 = U[2 * i + 1];
  const double* g = kK + i * KR;
#pragma unroll
  for (int k = 0; k < KR; ++k) o.g[k] = g[k];
}

// One step of iLQR::forward_pass (I/iLQR.cpp:77-85) with Model::forward_simulate (I/Model.cpp:17-30) inlined.
__device__ __forceinline__ void forward_step(const FwdConst& k, const FwdIn& c, State& s, double& max_th, double* Un_i,
                                             double* Xn_next) {
  const double d0 = s.x - c.x, d1 = s.y - c.y, d2 = s.v - c.v, d3 = s.th - c.th;
  const double u0 = fma(c.g[5], d3, fma(c.g[4], d2, fma(c.g[3], d1, fma(c.g[2], d0, c.u0 + c.g[0]))));
  const double u1 = fma(c.g[9], d3, fma(c.g[8], d2, fma(c.g[7], d1, fma(c.g[6], d0, c.u1 + c.g[1]))));
  const double a = vmax(vmin(u0, k.acc_max), k.acc_min);
  const double w = vmax(vmin(u1, s.v * k.yaw_hi), s.v * k.yaw_lo);
  const double adv = fma(a, k.half_dt2, s.v * k.dt);
  s.x = fma(s.c, adv, s.x);
  s.y = fma(s.s, adv, s.y);
  s.v = vmin(vmax(fma(a, k.dt, s.v), k.zero), k.speed_max);
  s.th = fma(w, k.dt, s.th);
  max_th = vmax(max_th, fabs(s.th));
  sincos_loop(k, s.th, s.s, s.c);
  if (threadIdx.x == 0) {
    Un_i[0] = u0; Un_i[1] = u1;
    Xn_next[0] = s.x; Xn_next[1] = s.y; Xn_next[2] = s.v; Xn_next[3] = s.th; Xn_next[4] = s.c; Xn_next[5] = s.s;
  }
}

// Phase F: iLQR::forward_pass (I/iLQR.cpp:68-86).  All lanes compute the same values; lane 0 stores; the operands of
// step i+1 (old state, old control, gains) are read while step i computes (two steps per trip, no register copies).
// Returns false if a heading left the range of the in-loop sincos (|theta| ≥ 1e6 rad, or NaN): the results are then not
// to be used and the solve is handed to the GENERAL kernel.
__device__ __forceinline__ bool forward_fast(const KParams& kp, int N, const double* X, const double* U, const double* kK,
                                             double* Xn, double* Un) {
  FwdConst k;
  make_fwd_const(k, kp);
  State s;
  s.x = X[0]; s.y = X[1]; s.v = X[2]; s.th = X[3]; s.c = X[4]; s.s = X[5];
  if (threadIdx.x == 0) store_state(Xn, 0, s);
  double max_th = fabs(s.th);
  FwdIn fa, fb;
  load_fwd(fa, X, U, kK, 0);
  int i = 0;
  for (; i + 1 < N; i += 2) {
    load_fwd(fb, X, U, kK, i + 1);
    forward_step(k, fa, s, max_th, Un + 2 * i, Xn + (i + 1) * XR);
    load_fwd(fa, X, U, kK, i + 2 < N ? i + 2 : i + 1);
    forward_step(k, fb, s, max_th, Un + 2 * (i + 1), Xn + (i + 2) * XR);
  }
  if (i < N) forward_step(k, fa, s, max_th, Un + 2 * i, Xn + (i + 1) * XR);
  return max_th < 1.0e6;
}

// The same pass with the range-guarded sincos (library path for huge arguments); GENERAL kernel only.
__device__ __forceinline__ void forward_general(const KParams& kp, int N, const double* X, const double* U, const double* kK,
                                                double* Xn, double* Un) {
  State s;
  s.x = X[0]; s.y = X[1]; s.v = X[2]; s.th = X[3]; s.c = X[4]; s.s = X[5];
  if (threadIdx.x == 0) store_state(Xn, 0, s);
  for (int i = 0; i < N; ++i) {
    const double* xo = X + i * XR;
    const double* g = kK + i * KR;
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
__device__ __forceinline__ bool rollout_fast(const KParams& kp, int N, const double* x0, const double* U, double* X) {
  FwdConst k;
  make_fwd_const(k, kp);
  State s;
  s.x = x0[0]; s.y = x0[1]; s.v = x0[2]; s.th = x0[3];
  double max_th = fabs(s.th);
  sincos_loop(k, s.th, s.s, s.c);
  if (threadIdx.x == 0) store_state(X, 0, s);
  for (int i = 0; i < N; ++i) {
    const double u0 = U[2 * i], u1 = U[2 * i + 1];
    const double a = vmax(vmin(u0, k.acc_max), k.acc_min);
    const double w = vmax(vmin(u1, s.v * k.yaw_hi), s.v * k.yaw_lo);
    const double adv = fma(a, k.half_dt2, s.v * k.dt);
    s.x = fma(s.c, adv, s.x);
    s.y = fma(s.s, adv, s.y);
    s.v = vmin(vmax(fma(a, k.dt, s.v), k.zero), k.speed_max);
    s.th = fma(w, k.dt, s.th);
    max_th = vmax(max_th, fabs(s.th));
    sincos_loop(k, s.th, s.s, s.c);
    if (threadIdx.x == 0) store_state(X, i + 1, s);
  }
  return max_th < 1.0e6;
}

// DIAG: per-solve shader-clock totals by phase, written to a.diag[b][8] = {prologue, L, R, F, epilogue, L count, R count,
// total}.  A separate instantiation so that the production kernel carries no stamps.
// TABLDS: the obstacle table of the solve lives in LDS (chosen by the launcher when it fits beside the rest at the
// wanted residency) instead of the global workspace.
// GENERAL: false = the production kernel: branch-free fast passes.  A solve that meets anything the fast passes do not
// cover (Q_uu not positive semi-definite or not finite; a heading beyond the in-loop sincos range) stops WITHOUT touching
// its outputs and sets a.redo[b]; the GENERAL = true kernel, launched right behind on the same stream, redoes exactly
// those solves from their untouched inputs with the branching passes and returns at once for all others.
template <bool DIAG, bool TABLDS, bool GENERAL>
__global__ __launch_bounds__(WAVE) void cilqr_solve_kernel(SolveArgs a) {
  unsigned long long tk0 = 0, tk = 0, c_pro = 0, c_L = 0, c_R = 0, c_F = 0, n_L = 0, n_R = 0;
  if (DIAG) tk0 = tk = __builtin_readcyclecounter();
#define CILQR_STAMP(acc)                                 \
  if (DIAG) {                                            \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    acc += now_ - tk;                                    \
    tk = now_;                                           \
  }
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const KParams kp = a.kp;
  const int N = a.N, M = a.M, S = kp.n_samples;
  if (b >= a.B) return;
  if (GENERAL && a.redo[b] == 0) return;

  double* samp = lds;
  double* Xa = samp + 2 * S;
  double* Xb = Xa + (N + 1) * XR;
  double* Ua = Xb + (N + 1) * XR;
  double* Ub = Ua + 2 * N;
  double* rec = Ub + 2 * N;
  double* kK = rec + N * REC;
  double* tab = TABLDS ? kK + N * KR : a.obs_tab + (size_t)b * M * TABF * N;

  // ---- prologue -------------------------------------------------------------------------------------------
  SampleGrid grid;
  {  // path samples, I/Constraints.cpp:28-42 (ascending powers by repeated multiplication)
    const double* pc = a.poly + (size_t)b * CILQR_POLY_COEFFS;
    const double xf = a.xplan_fl[2 * b], xl = a.xplan_fl[2 * b + 1];
    const double dxs = (xl - xf) / (double)S;
    grid.xf = xf;
    grid.inv_dxs = 1.0 / dxs;
    grid.windowed = fabs(grid.inv_dxs) < 1.0e300 && fabs(dxs) < 1.0e300 && dxs != 0.0;  // finite, non-zero spacing
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
    handover = !rollout_fast(kp, N, a.x0 + (size_t)b * 4, Ua, Xa);
  }
  __syncthreads();

  CILQR_STAMP(c_pro)
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
  for (int it = 0; it < max_it && !handover; ++it) {
    ++iters;
    // The reference evaluates backward_pass, forward_pass, then J_new = get_J(X, U) on the CURRENT X, U (:213-217).
    // The linearisation and J share their closest-point searches, so they are computed together, first.
    J_new = readfirstlane_f64(wave_sum(linearize(kp, N, M, lane, samp, S, grid, Xc, Uc, rec, tab, wts)));
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
    if (!riccati<GENERAL>(kp, N, rec, kK, lamb)) {
      if (GENERAL) { status = CILQR_EXIT_NUMERIC; break; }
      handover = true;
      break;
    }
    __syncthreads();
    CILQR_STAMP(c_R)
    if (DIAG) ++n_R;
    if (GENERAL) {
      forward_general(kp, N, Xc, Uc, kK, Xn, Un);
    } else if (!forward_fast(kp, N, Xc, Uc, kK, Xn, Un)) {
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
    if (lane == 0) a.redo[b] = handover ? 1 : 0;
    if (handover) return;  // outputs (and the in/out U) untouched: the GENERAL kernel starts from the same inputs
  }

  // ---- epilogue: X_result / U_result (:243-244) ----------------------------------------------------------------
  __syncthreads();
  for (int i = lane; i < 2 * N; i += WAVE) Ug[i] = Uc[i];
  double* Xg = a.X_out + (size_t)b * 4 * (N + 1);
  for (int i = lane; i < 4 * (N + 1); i += WAVE) Xg[i] = Xc[(i >> 2) * XR + (i & 3)];
  if (a.J_out) {
    if (!j_valid) J_new = readfirstlane_f64(wave_sum(cost_only(kp, N, lane, samp, S, grid, Xc, Uc)));
    if (lane == 0) a.J_out[b] = J_new;
  }
  if (lane == 0) {
    if (a.iters_out) a.iters_out[b] = iters;
    if (a.status_out) a.status_out[b] = status;
  }
  if (DIAG && lane == 0 && a.diag) {
    const unsigned long long now_ = __builtin_readcyclecounter();
    unsigned long long* o = a.diag + (size_t)b * 8;
    o[0] = c_pro; o[1] = c_L; o[2] = c_R; o[3] = c_F; o[4] = now_ - tk; o[5] = n_L; o[6] = n_R; o[7] = now_ - tk0;
  }
#undef CILQR_STAMP
}

}  // namespace

size_t solve_lds_bytes(int N, int n_samples) {
  const size_t doubles = 2 * (size_t)n_samples + 2 * (size_t)(N + 1) * XR + 2 * (size_t)2 * N + (size_t)N * REC + (size_t)N * KR;
  return doubles * sizeof(double);
}

namespace {
template <bool DIAG, bool TABLDS>
void launch_pair(const SolveArgs& a, size_t lds, hipStream_t stream) {
  hipLaunchKernelGGL((cilqr_solve_kernel<DIAG, TABLDS, false>), dim3(a.B), dim3(WAVE), lds, stream, a);
  hipLaunchKernelGGL((cilqr_solve_kernel<DIAG, TABLDS, true>), dim3(a.B), dim3(WAVE), lds, stream, a);
}
}  // namespace

namespace {
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
}  // namespace

hipError_t launch_quu_inverse(int n, const double* q, const double* lamb, double* out, int general, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(quu_inverse_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, n, q, lamb, out, general);
  return hipGetLastError();
}

hipError_t launch_solve(const SolveArgs& a, hipStream_t stream) {
  if (a.B <= 0) return hipSuccess;
  size_t lds = solve_lds_bytes(a.N, a.kp.n_samples);
  // Keep the obstacle table in LDS while a workgroup stays within 32 KiB (≥ 5 solves resident per CU of 160 KiB).
  const size_t tab_bytes = (size_t)a.M * TABF * a.N * sizeof(double);
  const bool tab_lds = a.M > 0 && lds + tab_bytes <= 32 * 1024;
  if (tab_lds) lds += tab_bytes;
  if (a.diag) {
    if (tab_lds) launch_pair<true, true>(a, lds, stream);
    else launch_pair<true, false>(a, lds, stream);
  } else {
    if (tab_lds) launch_pair<false, true>(a, lds, stream);
    else launch_pair<false, false>(a, lds, stream);
  }
  return hipGetLastError();
}

}  // namespace cilqr
