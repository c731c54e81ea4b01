// cilqr_device.hpp — device-side arithmetic of the CILQR solve shared by the two kernel families:
//   cilqr_solve.hip         one wavefront per solve, LDS-resident (small batches: one solve per SIMD)
//   cilqr_solve_groups.hip  G lanes per solve, 64/G solves per wavefront, workspace in global memory (large batches)
// Everything here is per (solve, step) and independent of where the data lives.  Reference citations:
// I/ = CILQR/src/ilqr/include/ilqr/ of Leo-Liao-Chao/Uncertainty-Aware-CILQR-for-Trajectory-Optimization.
#pragma once

#include <stddef.h>

#include <float.h>

#include "cilqr_internal.h"

namespace cilqr {
namespace dev {

constexpr int WAVE = 64;
constexpr int XR = 6;    // doubles per state record {x, y, v, theta, cos theta, sin theta}
constexpr int REC = 16;  // doubles per linearisation record
constexpr int KR = 10;   // doubles per gain record {k(2), K(2x4)}
constexpr int TABF = 6;  // fields per obstacle-table entry

// ---- single-instruction helpers ------------------------------------------------------------------------------------
// One wavefront per SIMD issues one instruction every ~5 shader ticks whatever its kind (tools/ubench_issue.hip), so the
// serial phases are priced in instructions.  These keep hipcc from adding canonicalising v_max around fmin/fmax and from
// re-materialising 64-bit literals with s_mov pairs inside the loops.
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
__device__ __forceinline__ double vmax_abs(double a, double b) {  // max(a, |b|): the absolute value is a source modifier
  double r;
  asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// Pins a loop-invariant value in a vector register: after this the compiler cannot fold it back into a literal.
#define CILQR_PIN(x) asm volatile("" : "+v"(x))
#define CILQR_PIN2(x, y) asm volatile("" : "+v"(x), "+v"(y))

// The parameter block as it lies in the kernel-argument segment (SolveArgs is the first kernel parameter of every solve
// kernel), through a pointer the compiler cannot trace back to the preloaded arguments.  A phase that reads its parameters
// through this keeps their scalar registers live for that phase only; read once at kernel entry, the 35 doubles are carried
// — and spilled to vector lanes, one v_readlane per use — across every phase (grouped family: 215 → 122 spilled SGPRs,
// 44 → 16 reloads per pair of obstacle entries).
__device__ __forceinline__ const KParams& phase_params() {
  const KParams* q = reinterpret_cast<const KParams*>(
      reinterpret_cast<const char*>((const void*)__builtin_amdgcn_kernarg_segment_ptr()) + offsetof(SolveArgs, kp));
  asm volatile("" : "+s"(q));
  return *q;
}

// The whole argument block the same way (the epilogue's output pointers need not stay live through the iteration loop).
__device__ __forceinline__ const SolveArgs& phase_args() {
  const SolveArgs* q = reinterpret_cast<const SolveArgs*>((const void*)__builtin_amdgcn_kernarg_segment_ptr());
  asm volatile("" : "+s"(q));
  return *q;
}

// 1/x by v_rcp_f64 and two Newton steps (≤ ~1 ulp; x is a well-scaled positive determinant here).
__device__ __forceinline__ double rcp_newton(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// exp(x) in ≈21 instructions, ≤ ~2 ulp: x = n·ln2 + r with a two-part ln2 (|r| ≤ 0.347), degree-13 Taylor polynomial in
// Horner form, scaling by 2^n with v_ldexp_f64 (gradual underflow to 0 for very negative x).  The library exp costs
// ≈55 instructions; the barrier terms call it 2M + 4 times per (solve, step), which makes it the largest single item of
// the linearisation for M ≥ 4 (and ≈60 % of the whole solve at M = 256).
__device__ __forceinline__ double exp_fast(double x) {
  const double n = rint(x * 1.44269504088896338700e+00);
  double r = fma(-n, 6.93147180369123816490e-01, x);  // ln2 head (low bits zero: n·head exact for |n| < 2^21)
  r = fma(-n, 1.90821492927058770002e-10, r);         // ln2 tail
  double p = 1.0 / 6227020800.0;                       // 1/13!
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  const double e = ldexp(p, (int)n);
  return x > 710.0 ? __builtin_huge_val() : e;  // beyond the overflow threshold the reduction itself is meaningless
}

// ---- dynamics --------------------------------------------------------------------------------------------------------
struct State {
  double x, y, v, th, c, s;
};

// sin and cos of one fp64 argument, ≤ ~1 ulp each: three-part Cody–Waite reduction by pi/2 (exact first step under fma for
// |x| < 2^20·pi/2) followed by the classic degree-13 / degree-14 minimax kernels on [-pi/4, pi/4].  Larger arguments (never
// met by a heading angle) take the library path.
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
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                               2.75573137070700676789e-06), -1.98412698298579493134e-04),
                               8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double sr = fma(z * r, ps, r);
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

struct FwdConst {  // loop invariants of the forward pass, pinned in vector registers
  double dt, half_dt2, acc_max, acc_min, yaw_hi, yaw_lo, speed_max, zero;
  double sn1, sn2, sn3, sn4, sn5, cs1, cs2, cs3, cs4, cs5, cs6;  // Taylor coefficients of sin d / d and (cos d - 1) / d², by powers of d²
};
__device__ __forceinline__ void make_fwd_const(FwdConst& k, const KParams& kp) {
  k.dt = kp.dt; k.half_dt2 = kp.half_dt2; k.acc_max = kp.acc_max; k.acc_min = kp.acc_min;
  k.yaw_hi = kp.yaw_hi; k.yaw_lo = kp.yaw_lo; k.speed_max = kp.speed_max; k.zero = 0.0;
  k.sn1 = -1.0 / 6.0; k.sn2 = 1.0 / 120.0; k.sn3 = -1.0 / 5040.0; k.sn4 = 1.0 / 362880.0; k.sn5 = -1.0 / 39916800.0;
  k.cs1 = -0.5; k.cs2 = 1.0 / 24.0; k.cs3 = -1.0 / 720.0; k.cs4 = 1.0 / 40320.0; k.cs5 = -1.0 / 3628800.0; k.cs6 = 1.0 / 479001600.0;
  CILQR_PIN(k.dt); CILQR_PIN(k.half_dt2); CILQR_PIN(k.acc_max); CILQR_PIN(k.acc_min); CILQR_PIN(k.yaw_hi);
  CILQR_PIN(k.yaw_lo); CILQR_PIN(k.speed_max); CILQR_PIN(k.zero);
  CILQR_PIN(k.sn1); CILQR_PIN(k.sn2); CILQR_PIN(k.sn3); CILQR_PIN(k.sn4); CILQR_PIN(k.sn5);
  CILQR_PIN(k.cs1); CILQR_PIN(k.cs2); CILQR_PIN(k.cs3); CILQR_PIN(k.cs4); CILQR_PIN(k.cs5); CILQR_PIN(k.cs6);
}

// sincos_fast without its range guard (same arithmetic, same results for |x| < 1e6; callers track max|x| and hand the solve to
// the GENERAL kernel if that bound was ever exceeded).  Off the per-step chain since the headings are advanced by rotation
// (rotate_heading): used for a trajectory's first state and for turns of more than 1/4 rad per step.
__device__ __forceinline__ void sincos_loop(double x, double& sn, double& cs) {
  const double n = rint(x * 6.36619772367581382433e-01);
  double r = fma(-n, 1.57079632679489655800e+00, x);
  r = fma(-n, 6.12323399573676603587e-17, r);
  r = fma(-n, -1.49738490485916983278e-33, r);
  const double z = r * r;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                               2.75573137070700676789e-06), -1.98412698298579493134e-04),
                               8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double sr = fma(z * r, ps, r);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                               -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                               -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  const double cr = w + (((1.0 - w) - hz) + z * (z * pc));
  const int q = (int)n;
  const bool odd = (q & 1) != 0;
  const double s0 = odd ? cr : sr;
  const double c0 = odd ? sr : cr;
  const int sgs = (q & 2) << 30, sgc = ((q + 1) & 2) << 30;  // sign flips as integer xors on the high words
  sn = __hiloint2double(__double2hiint(s0) ^ sgs, __double2loint(s0));
  cs = __hiloint2double(__double2hiint(c0) ^ sgc, __double2loint(c0));
}

// cos and sin of the heading advanced by a small turn: (c, s) rotated by delta = w·dt through the addition formulas, with sin
// delta and cos delta from their Taylor series (|delta| ≤ 1/4: truncation below 2e-18 relative) — 17 branch-free instructions
// instead of the 35 of a range-reduced sincos of the new heading, on the serial chain of every forward step.  The rounding of
// each rotation (≈1e-16) accumulates over the horizon instead of being redone from the rounded angle: ≈1e-15 after 80 steps,
// far inside the parity tolerance.  Callers track the largest |delta| (MAX_TURN) and hand the solve to the GENERAL kernel —
// which evaluates sincos of every heading — if a turn was larger (above 8 m/s at full lock with the reference's parameters); a
// first heading beyond MAX_HEADING0 goes there too, so that |theta| stays in the range of sincos_loop over any horizon.
constexpr double MAX_TURN = 0.25, MAX_HEADING0 = 9.99e5;
__device__ __forceinline__ void rotate_heading(const FwdConst& k, double delta, double& sn, double& cs) {
  const double z = delta * delta;
  // sin d = d (1 - z/3! + z²/5! - z³/7! + z⁴/9! - z⁵/11!);  cos d - 1 = z (-1/2 + z/4! - z²/6! + z³/8! - z⁴/10! + z⁵/12!).
  // The two Horner chains are written interleaved and pinned that way: a dependent fp64 instruction issues every 8.5 ticks,
  // an independent one every 5.2 (tools/ubench_exec.hip), and hipcc emits the chains one after the other otherwise.
  double ps = fma(z, k.sn5, k.sn4);
  double pc = fma(z, k.cs6, k.cs5);
  CILQR_PIN2(ps, pc);
  pc = fma(pc, z, k.cs4);
  ps = fma(ps, z, k.sn3);
  CILQR_PIN2(ps, pc);
  pc = fma(pc, z, k.cs3);
  ps = fma(ps, z, k.sn2);
  CILQR_PIN2(ps, pc);
  pc = fma(pc, z, k.cs2);
  ps = fma(ps, z, k.sn1);
  CILQR_PIN2(ps, pc);
  pc = fma(pc, z, k.cs1);
  const double sd = fma(delta * z, ps, delta);
  const double cdm1 = pc * z;
  const double c0 = cs, s0 = sn;
  cs = fma(c0, cdm1, fma(-s0, sd, c0));
  sn = fma(s0, cdm1, fma(c0, sd, s0));
}

// Model::forward_simulate on the in-loop constants, in two halves: position, speed and heading (returns the turn delta = w·dt);
// then cos/sin of the new heading.  max_turn: running maximum of |w·dt| (see rotate_heading).
__device__ __forceinline__ double dyn_pose_loop(const FwdConst& k, State& s, double u0, double u1, double& max_turn) {
  const double a = vmax(vmin(u0, k.acc_max), k.acc_min);
  const double w = vmax(vmin(u1, s.v * k.yaw_hi), s.v * k.yaw_lo);
  const double adv = fma(a, k.half_dt2, s.v * k.dt);
  s.x = fma(s.c, adv, s.x);
  s.y = fma(s.s, adv, s.y);
  s.v = vmin(vmax(fma(a, k.dt, s.v), k.zero), k.speed_max);
  const double delta = w * k.dt;
  s.th = s.th + delta;
  max_turn = vmax_abs(max_turn, delta);
  return delta;
}
__device__ __forceinline__ void dyn_step_loop(const FwdConst& k, State& s, double u0, double u1, double& max_turn) {
  const double delta = dyn_pose_loop(k, s, u0, u1, max_turn);
  rotate_heading(k, delta, s.s, s.c);
}

// One step of iLQR::forward_pass (I/iLQR.cpp:77-85): old state (ox..oth), old control (ou0, ou1), gains g[10].
struct FwdIn {
  double x, y, v, th, u0, u1, g[KR];
};
// SCALED (production one-wavefront kernel): the acceleration gains g[0], g[2..5] are in units of (dt/2)·u0 — the backward pass on
// the matrix cores works with B's first column divided by dt/2 (riccati_mfma) — and `inv_half_dt` = 2/dt brings them back in
// the instruction that adds the old control: as many instructions as the plain form.
template <bool SCALED = false>
__device__ __forceinline__ void forward_controls(const FwdIn& c, const State& s, double& u0, double& u1, double inv_half_dt = 1.0) {
  const double d0 = s.x - c.x, d1 = s.y - c.y, d2 = s.v - c.v, d3 = s.th - c.th;
  if (SCALED) u0 = fma(fma(c.g[5], d3, fma(c.g[4], d2, fma(c.g[3], d1, fma(c.g[2], d0, c.g[0])))), inv_half_dt, c.u0);
  else u0 = fma(c.g[5], d3, fma(c.g[4], d2, fma(c.g[3], d1, fma(c.g[2], d0, c.u0 + c.g[0]))));
  u1 = fma(c.g[9], d3, fma(c.g[8], d2, fma(c.g[7], d1, fma(c.g[6], d0, c.u1 + c.g[1]))));
}
// The same controls (scaled form) for operands that lie in SCALAR registers (forward_smem): a vector instruction takes one
// scalar operand, so a chain must not start from two of them — the products are summed first and the constant terms added
// behind, one scalar each: 12 instructions and no register copies instead of 11 + 6 v_mov.
__device__ __forceinline__ void forward_controls_scalar(const FwdIn& c, const State& s, double& u0, double& u1, double inv_half_dt) {
  const double d0 = s.x - c.x, d1 = s.y - c.y, d2 = s.v - c.v, d3 = s.th - c.th;
  const double t0 = fma(c.g[5], d3, fma(c.g[4], d2, fma(c.g[3], d1, c.g[2] * d0)));
  const double t1 = fma(c.g[9], d3, fma(c.g[8], d2, fma(c.g[7], d1, c.g[6] * d0)));
  // (the old control as the SCALAR addend of a three-operand fma: left to itself hipcc copies it into vector registers for v_fmac)
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(u0) : "v"(t0 + c.g[0]), "v"(inv_half_dt), "s"(c.u0));
  u1 = (t1 + c.g[1]) + c.u1;
}
template <bool SCALED = false>
__device__ __forceinline__ void forward_step(const FwdConst& k, const FwdIn& c, State& s, double& max_turn, double& u0, double& u1,
                                             double inv_half_dt = 1.0) {
  forward_controls<SCALED>(c, s, u0, u1, inv_half_dt);
  dyn_step_loop(k, s, u0, u1, max_turn);
}

// ---- path samples and the closest-point search -----------------------------------------------------------------------
// Sample s of the local path (I/Constraints.cpp:28-42): x_s = xf + dxs*s, y_s = sum_j c_j x_s^j with ascending powers
// formed by repeated multiplication.  One statement of it, used wherever a sample is needed, so that every kernel sees
// bit-identical samples.
struct SampleGrid {
  double xf, dxs, inv_dxs;  // inv_dxs = 1/dxs (signed)
  double dmax;              // largest |y_{s+1} - y_s| over the samples (set once per solve; +inf: not known)
  double p2;                // an upper bound of |p''| over the sampled range (path_curvature_bound; +inf: not known → no Newton search)
  const double* pc;         // the polynomial's coefficients, ascending (null: no Newton search)
  bool windowed;            // false: dxs is 0 or not finite → full scan
};
__device__ __forceinline__ void make_sample_grid(SampleGrid& g, double xf, double xl, int S) {
  g.xf = xf;
  g.dxs = (xl - xf) / (double)S;
  g.inv_dxs = 1.0 / g.dxs;
  g.dmax = __builtin_huge_val();
  g.p2 = __builtin_huge_val();
  g.pc = nullptr;
  g.windowed = fabs(g.inv_dxs) < 1.0e300 && fabs(g.dxs) < 1.0e300 && g.dxs != 0.0;
}
// Value, first and second derivative of the path polynomial at x (Horner from the top; for the Newton search below — the SAMPLES are
// formed by sample_xy alone).
__device__ __forceinline__ void poly_012(const double* pc, double x, double& p, double& d1, double& d2) {
  double b = pc[CILQR_POLY_COEFFS - 1], c = 0.0, d = 0.0;
#pragma unroll
  for (int j = CILQR_POLY_COEFFS - 2; j >= 0; --j) {
    d = fma(d, x, c);
    c = fma(c, x, b);
    b = fma(b, x, pc[j]);
  }
  p = b; d1 = c; d2 = d + d;
}
// One sample's contribution to an upper bound of the second derivative over the sampled range [x_0, x_{S-1}]: |p2| and |p3| (second
// and third derivative) at x.  With A = max_s |p2(x_s)|, B = max_s |p3(x_s)| and C = max |p4| over the range (the fourth derivative
// is linear: its endpoints), Taylor between neighbouring samples gives |p2| ≤ A + (h/2)(B + (h/2) C) on the whole range
// (path_curvature_bound puts them together).
static_assert(CILQR_POLY_COEFFS == 6, "the curvature bound is written for the reference's degree-5 local plan");
__device__ __forceinline__ void path_curvature_terms(const double* pc, double x, double& a2, double& a3) {
  a2 = fabs(fma(fma(fma(20.0 * pc[5], x, 12.0 * pc[4]), x, 6.0 * pc[3]), x, 2.0 * pc[2]));
  a3 = fabs(fma(fma(60.0 * pc[5], x, 24.0 * pc[4]), x, 6.0 * pc[3]));
}
__device__ __forceinline__ double path_curvature_bound(const SampleGrid& g, const double* pc, int S, double A, double B) {
  const double x0 = g.xf, x1 = fma(g.dxs, (double)(S - 1), g.xf), hh = 0.5 * fabs(g.dxs);
  const double C = fmax(fabs(fma(120.0 * pc[5], x0, 24.0 * pc[4])), fabs(fma(120.0 * pc[5], x1, 24.0 * pc[4])));
  const double P = (A + hh * (B + hh * C)) * (1.0 + 1.0e-6);
  return P == P ? P : __builtin_huge_val();  // (a NaN anywhere: not known)
}
__device__ __forceinline__ void sample_xy(const SampleGrid& g, const double* pc, int s, double& x, double& y) {
  x = fma(g.dxs, (double)s, g.xf);
  double acc = pc[0], pw = x;
#pragma unroll
  for (int j = 1; j < CILQR_POLY_COEFFS; ++j) {
    acc = fma(pc[j], pw, acc);
    pw *= x;
  }
  y = acc;
}

// Closest path sample to (px, py): index of the strict-< first minimum of the squared distance over ALL S samples
// (I/Constraints.cpp:43-56), found without visiting all of them.  A sample can only reach the distance d_c of the sample
// nearest in x if its own x-offset satisfies (x_s - px)² ≤ d_c, because fl(dx² + dy²) ≥ fl(dx²); the x_s are equispaced,
// so that is an index window around (px - xf)/dxs.  The window is widened by two samples and a 1e-4 relative margin (≫ any
// rounding in its own computation), clamped to [0, S-1] and scanned in ascending order with strict <, which yields exactly
// the reference's argmin, ties included.  `at(s, x, y)` supplies sample s.
// The squared distance of sample s from (px, py), as every search below forms it.
template <typename SampleAt>
__device__ __forceinline__ double sample_dist(SampleAt at, int s, double px, double py) {
  double sx, sy;
  at(s, sx, sy);
  return (sx - px) * (sx - px) + (sy - py) * (sy - py);
}

// The index window [lo, hi] that contains the argmin (see closest_sample): one statement of it for every search, in two parts.
// First part, from the x side (always a superset of the argmin's neighbourhood).  Also hands back what the refinements need:
// c = the sample nearest in x, its position, its squared distance dc, the half-width hw (in samples; < 0: no window, full range).
struct XWindow {
  int lo, hi, c;
  double scx, scy, dc, hw;
};
template <typename SampleAt>
__device__ __forceinline__ XWindow closest_window_x(int S, const SampleGrid& g, double px, double py, SampleAt at) {
  XWindow w;
  w.lo = 0; w.hi = S - 1; w.c = 0; w.scx = 0.0; w.scy = 0.0; w.dc = 0.0; w.hw = -1.0;
  if (g.windowed) {
    const double fc = (px - g.xf) * g.inv_dxs;
    const double fcc = fmin(fmax(fc, 0.0), (double)(S - 1));  // NaN → 0
    w.c = (int)(fcc + 0.5);
    at(w.c, w.scx, w.scy);
    w.dc = sample_dist(at, w.c, px, py);
    const double hw = (double)(__builtin_sqrtf((float)w.dc) * 1.0001f) * fabs(g.inv_dxs) * 1.0001 + 2.0;
    if (hw < 1.0e9) {  // false for NaN / overflow: keep the full range
      w.lo = (int)fmin(fmax(fc - hw, 0.0), (double)(S - 1));
      w.hi = (int)fmax(fmin(fc + hw + 1.0, (double)(S - 1)), 0.0);
      w.hw = hw;
    }
  }
  return w;
}
// Second, usually much tighter window from the y side.  Sample c + n (or c - n) has |x_s - px| ≥ n·h - e and, by the
// triangle inequality over adjacent samples, |y_s - py| ≥ Ly - n·D, with h = |dxs|, e = |x_c - px|, Ly = |y_c - py|,
// D = dmax.  While both right-hand sides are non-negative, d_s ≥ (n h - e)² + (Ly - n D)²; a sample can only reach
// d_c if that is ≤ d_c, i.e. n ≤ [B + sqrt(B² + A (T - e² - Ly²))] / A with A = h² + D², B = h e + Ly D.  Used only when
// the y-bound stays non-negative across the whole first window (D·hw ≤ Ly) and c really is the sample nearest in x
// (e ≤ h); widened by a 1e-9 relative margin and two samples, like the first window ≫ any rounding in its terms.
__device__ __forceinline__ void closest_window_y(int S, const SampleGrid& g, double px, double py, XWindow& w) {
  if (!(w.hw >= 0.0)) return;
  const double h = fabs(g.dxs), D = g.dmax, e = fabs(w.scx - px), Ly = fabs(w.scy - py);
  if (D * w.hw <= Ly && e <= h) {
    const double A = h * h + D * D, Bq = h * e + Ly * D;
    const double T = w.dc * (1.0 + 1.0e-9);
    const double disc = Bq * Bq + A * (T - e * e - Ly * Ly);
    const double nmax = (Bq + sqrt(fmax(disc, 0.0))) / A * (1.0 + 1.0e-9) + 2.0;
    if (nmax < 1.0e9) {  // false for NaN
      w.lo = max(w.lo, (int)fmax((double)w.c - nmax, 0.0));
      w.hi = min(w.hi, (int)fmin((double)w.c + nmax + 1.0, (double)(S - 1)));
    }
  }
}
// YSIDE = false leaves out the second (y-side) window: still a superset of the argmin's neighbourhood, i.e. the same argmin; worth
// it where the scan is dealt to several lanes and the window's square root and division cost more than the candidates they save.
template <bool YSIDE = true, typename SampleAt>
__device__ __forceinline__ void closest_window(int S, const SampleGrid& g, double px, double py, SampleAt at, int& lo, int& hi) {
  XWindow w = closest_window_x(S, g, px, py, at);
  if (YSIDE) closest_window_y(S, g, px, py, w);
  lo = w.lo;
  hi = w.hi;
}

// The scan over [lo, hi]: ascending, strict <.
template <typename SampleAt>
__device__ __forceinline__ int closest_scan(int lo, int hi, double px, double py, SampleAt at) {
  auto dist = [&](int s) { return sample_dist(at, s, px, py); };
  double md = dist(lo);
  int best = lo;
  for (int s = lo + 1; s <= hi; s += 4) {
    // four candidates per trip (indices past hi repeat hi: harmless under strict <), loads issued together
    const int s1 = min(s + 1, hi), s2 = min(s + 2, hi), s3 = min(s + 3, hi);
    const double d0 = dist(s), d1 = dist(s1), d2 = dist(s2), d3 = dist(s3);
    if (d0 < md) { md = d0; best = s; }
    if (d1 < md) { md = d1; best = s1; }
    if (d2 < md) { md = d2; best = s2; }
    if (d3 < md) { md = d3; best = s3; }
  }
  return best;
}

// Wide windows (a state well off the path: the x-side window is as wide as the state is far, and most of that is lateral) by NEWTON
// instead of a scan (round 3).  f(s) = (x_s - px)² + (p(x_s) - py)² as a function of a continuous s has f'' = 2h²·g with
// g = 1 + p'² + (p - py)·p''.  Over the first window |p - py| ≤ Ly + D·(hw + 1) (adjacent samples differ by at most D = dmax in y)
// and |p''| ≤ P2 (path_curvature_bound), so (Ly + D(hw + 1))·P2 ≤ ½ makes g ≥ ½ there: f is strictly convex over the whole window,
// its values at the samples are a convex sequence with second differences ≥ h² — far above the rounding of a squared distance
// (guarded: h² > 1e-9·(dc + 1)) — and a sample that is no larger than its evaluated neighbours on both sides (or sits at the
// window's edge) is THE minimum over the window, hence over all samples; among equal values the ascending strict-< comparison keeps
// the first, as the scan does.  Newton on phi(x) = (x - px) + (p - py)·p' (phi' = g ≥ ½) from the sample nearest in x, two steps
// (the first is the projection onto the tangent at that sample), clamped to the window; then the four samples around its result are evaluated exactly as the scan evaluates them.  If the best of
// them is not enclosed (Newton off by more than a sample) the scan runs after all.  Returns -1 where the conditions do not hold.
constexpr int NEWTON_MIN_WINDOW = 16;
template <typename SampleAt>
__device__ __forceinline__ int closest_newton(int S, const SampleGrid& g, double px, double py, SampleAt at, const XWindow& w) {
  if (!g.pc || !(w.hw >= 0.0)) return -1;
  const double Ly = fabs(w.scy - py);
  if (!((Ly + g.dmax * (w.hw + 1.0)) * g.p2 <= 0.5) || !(g.dxs * g.dxs > 1.0e-9 * (w.dc + 1.0))) return -1;
  const double xa = fma(g.dxs, (double)w.lo, g.xf), xb = fma(g.dxs, (double)w.hi, g.xf);
  const double xmin = fmin(xa, xb), xmax = fmax(xa, xb);
  double x = w.scx;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    double p, d1, d2;
    poly_012(g.pc, x, p, d1, d2);
    const double phi = fma(p - py, d1, x - px), dphi = fma(p - py, d2, fma(d1, d1, 1.0));
    x = fmin(fmax(fma(-phi, __builtin_amdgcn_rcp(dphi), x), xmin), xmax);
  }
  const double fs = (x - g.xf) * g.inv_dxs;
  const int k = (int)fmin(fmax(floor(fs), (double)w.lo), (double)w.hi);  // NaN → lo
  const int s0 = max(k - 1, w.lo), s1 = k, s2 = min(k + 1, w.hi), s3 = min(k + 2, w.hi);
  const double d0 = sample_dist(at, s0, px, py), e1 = sample_dist(at, s1, px, py), e2 = sample_dist(at, s2, px, py),
               e3 = sample_dist(at, s3, px, py);
  double md = d0;
  int best = s0;
  if (e1 < md) { md = e1; best = s1; }
  if (e2 < md) { md = e2; best = s2; }
  if (e3 < md) { md = e3; best = s3; }
  const bool enclosed = (best > s0 || s0 == w.lo) && (best < s3 || s3 == w.hi);
  return enclosed ? best : -1;
}

// NEWTON = false: windows and scan only — the same argmin; for the instantiations that have no registers to spare for the second
// search (the one-wavefront kernels sit at their 256-register budget) and for the rarely run paths.
template <bool NEWTON = true, typename SampleAt>
__device__ __forceinline__ int closest_sample(int S, const SampleGrid& g, double px, double py, SampleAt at) {
  XWindow w = closest_window_x(S, g, px, py, at);
  closest_window_y(S, g, px, py, w);
  // Newton where the scan would visit more than ≈ 16 samples (its two steps and four candidates cost what ≈ 13 candidates do)
  if (NEWTON && w.hi - w.lo >= NEWTON_MIN_WINDOW) {
    const int fast = closest_newton(S, g, px, py, at, w);
    if (fast >= 0) return fast;
  }
  return closest_scan(w.lo, w.hi, px, py, at);
}

// ---- cost linearisation of one step ----------------------------------------------------------------------------------
// Stage cost of Constraints::get_J (I/Constraints.cpp:534-561) for one step.
__device__ __forceinline__ double stage_cost(const KParams& kp, double dx, double dy, double dv, double u0, double u1) {
  // (fused multiply-adds written out: the same bits wherever a kernel forms it)
  const double xc = __builtin_fma(dv * kp.w_vel, dv, __builtin_fma(dy * kp.w_pos, dy, (dx * kp.w_pos) * dx));
  const double uc = __builtin_fma(u1 * kp.w_yawrate, u1, (u0 * kp.w_acc) * u0);
  return xc + uc;
}

struct ObsEntry {  // one obstacle at one step, as the barrier needs it
  double ox, oy, co, so, ia2, ib2;
};
// I/Obstacle.cpp:41-62: pose = (x, y, v, theta), dim = (length, width) of the obstacle at this step.
__device__ __forceinline__ ObsEntry make_obs_entry(const KParams& kp, const double* pose, const double* dim) {
  ObsEntry e;
  sincos(pose[3], &e.so, &e.co);
  const double ea = dim[0] / 2.0 + fabs(pose[2] * e.co) * kp.t_safe + kp.s_safe_a + kp.ego_rad;
  const double eb = dim[1] / 2.0 + fabs(pose[2] * e.so) * kp.t_safe + kp.s_safe_b + kp.ego_rad + 1;
  e.ox = pose[0];
  e.oy = pose[1];
  e.ia2 = 1.0 / ea / ea;
  e.ib2 = 1.0 / eb / eb;
  return e;
}

// ---- costmap-lookup uncertainty cost ---------------------------------------------------------------------------
// Uncertainty::get_uncertainty_cost(state) → {x, vx, mx}, added to l_x / l_xx with weight w_uncertainty where
// Constraints::get_state_cost does (I/Constraints.cpp:188-201).  The reference's class Uncertainty is absent from its repository:
// the arithmetic is the one include/cilqr.h defines at cilqr_set_uncertainty_map (footprint probes → bilinear lookup of the blurred
// occupancy as GridMap::atPositionLinearInterpolated, G/grid_map_core/src/GridMap.cpp:770-837 → exponential barrier in the form
// of Obstacle::barrier_function, I/Obstacle.cpp:21-32; derivatives w.r.t. (x, y) only).  PARITY UNPINNED: the tests check it
// against an independent plain-C statement of the same definition.
struct UncPose {  // pose of solve b's map frame in the planning frame: position, cos and sin of its heading
  double px, py, cp, sp;
};
// Formed once per solve, in the prologue (a per-solve pose costs a sincos).  `u` is the block in the kernel-argument segment.
__device__ __forceinline__ UncPose unc_pose(const UncArgs& u, int b) {
  UncPose q{u.px, u.py, u.cp, u.sp};
  if (u.poses) {
    const double* po = u.poses + 3 * (size_t)b;
    q.px = po[0]; q.py = po[1];
    sincos(po[2], &q.sp, &q.cp);
  }
  return q;
}
// (x, y): state position; (ct, st): cos / sin of its heading.  Adds w_uncertainty·(vx, mx) to the gradient / Hessian sums and
// returns the mean barrier value (diagnostics only: get_J does not contain it, I/Constraints.cpp:553-557).  The map's constants
// are read from `u` (kernel-argument segment: scalar loads) where they are used — nothing of the map is carried in registers
// or on a stack between calls.
__device__ __forceinline__ double unc_cost_add(const UncArgs& u, const UncPose& po, int b, double x, double y, double ct, double st,
                                               double& lx0, double& lx1, double& h00, double& h01, double& h11) {
#pragma clang fp contract(off)  // probe positions and cell indices as the plain-C statement forms them
  const float* layer = u.layer + (size_t)b * (size_t)u.stride;
  const int rows = u.rows, cols = u.cols, nl = u.nl, nw = u.nw;
  const double x_first = u.x_first, y_first = u.y_first, inv_res = u.inv_res;
  const double la0 = u.la0, la_step = u.la_step, wb0 = u.wb0, wb_step = u.wb_step, q1 = u.q1, q2 = u.q2, scale = u.scale;
  double sx = 0.0, gx = 0.0, gy = 0.0, hxx = 0.0, hxy = 0.0, hyy = 0.0;
  for (int k = 0; k < nl; ++k) {
    const double a = la0 + (double)k * la_step;
    for (int l = 0; l < nw; ++l) {
      const double bb = wb0 + (double)l * wb_step;
      const double Px = x + (a * ct - bb * st), Py = y + (a * st + bb * ct);
      const double dx = Px - po.px, dy = Py - po.py;
      const double qx = po.cp * dx + po.sp * dy, qy = po.cp * dy - po.sp * dx;
      const double fi = (x_first - qx) * inv_res, fj = (y_first - qy) * inv_res;
      if (!(fi >= 0.0) || !(fj >= 0.0) || !(fi < (double)(rows - 1)) || !(fj < (double)(cols - 1))) continue;
      const int i0 = (int)fi, j0 = (int)fj;
      const double ti = fi - (double)i0, tj = fj - (double)j0;
      const float* c0 = layer + (size_t)j0 * rows + i0;
      const double f00 = c0[0], f10 = c0[1], f01 = c0[rows], f11 = c0[rows + 1];
      const double big = 1.0e300;  // finite test without library calls (NaN fails every comparison)
      if (!(fabs(f00) < big && fabs(f10) < big && fabs(f01) < big && fabs(f11) < big)) continue;
      const double a0 = f00 + ti * (f10 - f00), a1 = f01 + ti * (f11 - f01);
      const double o = a0 + tj * (a1 - a0);
      const double di = (f10 - f00) + tj * ((f11 - f01) - (f10 - f00));
      const double dj = a1 - a0;
      const double cqx = (-di * inv_res) * 0.01, cqy = (-dj * inv_res) * 0.01;
      const double cX = po.cp * cqx - po.sp * cqy, cY = po.sp * cqx + po.cp * cqy;
      const double e = q1 * exp_fast(q2 * (o * 0.01 - 1.0));
      const double sv = q2 * e, sm = q2 * q2 * e;
      sx += e;
      gx += sv * cX;
      gy += sv * cY;
      hxx += (sm * cX) * cX;
      hxy += (sm * cX) * cY;
      hyy += (sm * cY) * cY;
    }
  }
  lx0 += gx * scale;
  lx1 += gy * scale;
  h00 += hxx * scale;
  h01 += hxy * scale;
  h11 += hyy * scale;
  return sx / (double)(nl * nw);
}

// One per-step linearisation record.
struct Rec {
  double lx0, lx1, lx2, l00, l01, l11, lu0, lu1, luu0, luu1, al, be, ga, de, p, q;
};

// ---- pieces of one step's linearisation, shared by lin_step (one lane per step) and the four-lanes-per-step mapping of the
// two-wavefront kernel (cilqr_solve.hip, linearize_quads): the same statements, so that both mappings give the same bits.
struct ObsConsts {  // what every obstacle entry of one step needs of the ego state and the parameters
  double fxp, fyp, rxp, ryp;  // the two ego circle centres (I/Obstacle.cpp:65-69, 86-90)
  double q2f, q2r;            // barrier exponents' factors
  double svf, smf, svr, smr;  // gradient / Hessian factors; the reference's factor -2 of c-dot (I/Obstacle.cpp:75-78) is carried
                              // here (exact: powers of two), not applied to the vector
};
// (from here to ab_terms: contraction is switched off and every fused multiply-add is written out — these statements are compiled
// into several kernels, on different wavefronts of one solve, and must round alike in all of them)
__device__ __forceinline__ ObsConsts make_obs_consts(const KParams& kp, double px, double py, double ct, double st) {
#pragma clang fp contract(off)
  ObsConsts k;
  k.fxp = __builtin_fma(ct, kp.ego_front, px); k.fyp = __builtin_fma(st, kp.ego_front, py);
  k.rxp = __builtin_fma(-ct, kp.ego_rear, px); k.ryp = __builtin_fma(-st, kp.ego_rear, py);
  k.q2f = kp.q2_front; k.q2r = kp.q2_rear;
  k.svf = -2 * (kp.q2_front * kp.q1_front); k.smf = 4 * (kp.q2_front * kp.q2_front * kp.q1_front);
  k.svr = -2 * (kp.q2_rear * kp.q1_rear); k.smr = 4 * (kp.q2_rear * kp.q2_rear * kp.q1_rear);
  return k;
}
struct ObsPrep {  // an entry up to its barrier arguments
  double g0[2], g1[2], arg[2];
};
// both circles up to the barrier argument c = 1 - d'Pd (I/Obstacle.cpp:65-73, 86-94)
__device__ __forceinline__ void obs_prep(const ObsConsts& k, const ObsEntry& e, ObsPrep& p) {
#pragma clang fp contract(off)
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    const double ex = (side == 0 ? k.fxp : k.rxp) - e.ox, ey = (side == 0 ? k.fyp : k.ryp) - e.oy;
    const double d0 = __builtin_fma(e.co, ex, e.so * ey);
    const double d1 = __builtin_fma(e.co, ey, -(e.so * ex));
    p.g0[side] = d0 * e.ia2;
    p.g1[side] = d1 * e.ib2;
    const double c = 1 - __builtin_fma(p.g0[side], d0, p.g1[side] * d1);
    p.arg[side] = (side == 0 ? k.q2f : k.q2r) * c;
  }
}
// false: both barrier exponents are at or below -64 — the entry contributes less than e^-64 times O(10) factors at this step
__device__ __forceinline__ bool obs_needed(const ObsPrep& p) { return !(p.arg[0] <= -64.0) || !(p.arg[1] <= -64.0); }
struct ObsTerms {  // one entry's gradient and Hessian terms (front + rear circle), before its weight
  double gx, gy, gxx, gxy, gyy;
};
__device__ __forceinline__ ObsTerms obs_terms(const ObsConsts& k, const ObsEntry& e, const ObsPrep& p) {
#pragma clang fp contract(off)
  double gx = 0.0, gy = 0.0, gxx = 0.0, gxy = 0.0, gyy = 0.0;
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    const double h0 = __builtin_fma(e.co, p.g0[side], -(e.so * p.g1[side]));  // c-dot = -2 (h0, h1)
    const double h1 = __builtin_fma(e.so, p.g0[side], e.co * p.g1[side]);
    const double ee = exp_fast(p.arg[side]);
    const double sv = (side == 0 ? k.svf : k.svr) * ee;
    const double sm = (side == 0 ? k.smf : k.smr) * ee;
    gx = __builtin_fma(sv, h0, gx);
    gy = __builtin_fma(sv, h1, gy);
    gxx = __builtin_fma(sm * h0, h0, gxx);
    gxy = __builtin_fma(sm * h0, h1, gxy);
    gyy = __builtin_fma(sm * h1, h1, gyy);
  }
  return ObsTerms{gx, gy, gxx, gxy, gyy};
}
struct StepSums {  // l_x(0,1), l_xx(00, 01, 11) of one step while its obstacle terms are added
  double lx0, lx1, h00, h01, h11;
};
// (the fused multiply-adds are written out: every mapping of phase L — one lane per step, four lanes per step, a second wavefront
// for the obstacle terms — must round these sums alike, whatever the compiler would contract in its surroundings)
__device__ __forceinline__ void obs_accumulate(StepSums& a, const ObsTerms& g, double w) {
  a.lx0 = __builtin_fma(g.gx, w, a.lx0);
  a.lx1 = __builtin_fma(g.gy, w, a.lx1);
  a.h00 = __builtin_fma(g.gxx, w, a.h00);
  a.h01 = __builtin_fma(g.gxy, w, a.h01);
  a.h11 = __builtin_fma(g.gyy, w, a.h11);
}
// l_x(0,1) and l_xx(00, 01, 11) of a step: the tracking terms (I/Constraints.cpp:163-174) plus the SUM of the obstacle terms — the
// obstacle terms are added up on their own, from zero, in entry order, and join the tracking terms here in one addition each
// (round 3; before, they were added to the tracking terms one by one).  So whoever forms the obstacle sums — the step's own lane or
// a lane of another wavefront (cilqr_solve.hip, cilqr_solve_share_kernel) — the record has the same bits.
__device__ __forceinline__ void state_terms(const KParams& kp, double dx, double dy, const StepSums& a, double& lx0, double& lx1,
                                            double& l00, double& l01, double& l11) {
#pragma clang fp contract(off)
  lx0 = __builtin_fma(2 * kp.w_pos, dx, a.lx0);
  lx1 = __builtin_fma(2 * kp.w_pos, dy, a.lx1);
  l00 = kp.w_pos * 2 + a.h00;
  l01 = a.h01;
  l11 = kp.w_pos * 2 + a.h11;
}
// Control cost (I/Constraints.cpp:110-131): the arguments of its four barrier exponentials, then l_u, l_uu from their values.
__device__ __forceinline__ void ctrl_args(const KParams& kp, double u0, double u1, double v, double& a1, double& a2, double& a3, double& a4) {
#pragma clang fp contract(off)
  a1 = kp.q2_acc * (u0 - kp.acc_max);
  a2 = kp.q2_acc * (kp.acc_min - u0);
  a3 = kp.q2_yawrate * __builtin_fma(-v, kp.yaw_hi, u1);
  a4 = kp.q2_yawrate * __builtin_fma(v, kp.yaw_lo, -u1);
}
__device__ __forceinline__ void ctrl_terms(const KParams& kp, double u0, double u1, double e1, double e2, double e3, double e4, Rec& r) {
#pragma clang fp contract(off)
  const double sa = kp.q2_acc * kp.q1_acc, sy = kp.q2_yawrate * kp.q1_yawrate;
  const double ma = kp.q2_acc * kp.q2_acc * kp.q1_acc, my = kp.q2_yawrate * kp.q2_yawrate * kp.q1_yawrate;
  r.lu0 = __builtin_fma(2 * kp.w_acc, u0, __builtin_fma(sa, e1, -(sa * e2)));
  r.lu1 = __builtin_fma(2 * kp.w_yawrate, u1, __builtin_fma(sy, e3, -(sy * e4)));
  r.luu0 = __builtin_fma(ma, e2, ma * e1) + 2 * kp.w_acc;
  r.luu1 = __builtin_fma(my, e4, my * e3) + 2 * kp.w_yawrate;
}
// The six non-trivial Jacobian entries at (v_{t+1}, theta_{t+1}, a_t) (I/iLQR.cpp:102-106, I/Model.cpp:100-155)
__device__ __forceinline__ void ab_terms(const KParams& kp, double u0, double vn, double cn, double sn, Rec& r) {
#pragma clang fp contract(off)
  const double dt = kp.dt;
  const double adv = __builtin_fma(vn, dt, u0 * kp.half_dt2);
  r.al = dt * cn;            // A(2,0)
  r.be = dt * sn;            // A(2,1)
  r.ga = (-1) * sn * adv;    // A(3,0)
  r.de = cn * adv;           // A(3,1)
  r.p = kp.half_dt2 * cn;    // B(0,0)
  r.q = kp.half_dt2 * sn;    // B(0,1)
}

// The obstacle terms of one step (I/Constraints.cpp:177-187, I/Obstacle.cpp:39-112) added to the sums `a`, entries 0 … M-1 of `obs`:
// the loop of lin_step, on its own so that other wavefronts can evaluate a share of a step's entries (cilqr_solve.hip,
// cilqr_solve_split_kernel, cilqr_solve_share_kernel).  CULL, PAIRED, LANE_EXACT: see lin_step.
// SPLIT (the default): the entries with EVEN index are added up in one chain, in order, those with ODD index in another, and the
// odd sum joins the even one at the end — two independent accumulation chains for a lone wavefront, and an order of summation that two
// wavefronts can reproduce bit for bit, one taking the even entries, one the odd ones (each with SPLIT = false over its own
// entries: one chain, in order).
template <bool CULL, bool PAIRED, bool LANE_EXACT, bool SPLIT = true, typename ObsAt>
__device__ __forceinline__ void obstacle_loop(const ObsConsts& oc, int M, ObsAt obs, StepSums& a) {
  using Prep = ObsPrep;
  StepSums b{0.0, 0.0, 0.0, 0.0, 0.0};  // SPLIT: the odd entries' sums
  auto prep = [&](const ObsEntry& e, Prep& p) { obs_prep(oc, e, p); };
  // the wave-wide vote of CULL: false when the entry is negligible at every step of the wavefront
  auto wanted = [&](const Prep& p) { return __builtin_amdgcn_ballot_w64(obs_needed(p)) != 0; };
  auto finish = [&](const ObsEntry& e, const Prep& p, double w, bool odd) {
    if (CULL && LANE_EXACT) w = obs_needed(p) ? w : 0.0;
    obs_accumulate(SPLIT && odd ? b : a, obs_terms(oc, e, p), w);
  };
  if (CULL && PAIRED) {
    // Entries in PAIRS: the geometry of both first — two independent dependency chains that interleave (a lone wavefront issues a
    // dependent fp64 instruction every 8.5 ticks, an independent one every 5.2) — then one vote each, then the exponentials of
    // those that matter, in entry order (the sums are formed exactly as one by one).  The next pair's loads are issued between
    // the geometry and the votes.  `obs` returns false for an entry it has already established to be negligible
    // (SampledObstacles: a whole far obstacle at once); its stale registers still go through prep, which costs nothing that
    // matters and keeps the pair free of branches.
    ObsEntry e0{0, 0, 1, 0, 0, 0}, e1 = e0, e2 = e0, e3 = e0;
    double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0;
    bool v0 = false, v1 = false, v2 = false, v3 = false;
    if (M > 0) v0 = obs(0, e0, w0);
    if (M > 1) v1 = obs(1, e1, w1);
    for (int m = 0; m < M; m += 4) {
      {
        Prep pa, pb;
        const bool any = v0 || v1;  // wave-uniform
        if (any) { prep(e0, pa); prep(e1, pb); }
        v2 = m + 2 < M ? obs(m + 2, e2, w2) : false;
        v3 = m + 3 < M ? obs(m + 3, e3, w3) : false;
        if (any) {
          const bool na = v0 && wanted(pa), nb = v1 && wanted(pb);
          if (na) finish(e0, pa, w0, false);
          if (nb) finish(e1, pb, w1, true);
        }
      }
      if (m + 2 >= M) break;
      {
        Prep pa, pb;
        const bool any = v2 || v3;
        if (any) { prep(e2, pa); prep(e3, pb); }
        v0 = m + 4 < M ? obs(m + 4, e0, w0) : false;
        v1 = m + 5 < M ? obs(m + 5, e1, w1) : false;
        if (any) {
          const bool na = v2 && wanted(pa), nb = v3 && wanted(pb);
          if (na) finish(e2, pa, w2, false);
          if (nb) finish(e3, pb, w3, true);
        }
      }
    }
  } else {
    // two entries in flight, alternating registers (`ea`: the even entries, `eb`: the odd ones): the next entry's loads fly while this one computes
    auto add_entry = [&](const ObsEntry& e, double w, bool odd) {
      Prep p;
      prep(e, p);
      if (CULL && !wanted(p)) return;
      finish(e, p, w, odd);
    };
    ObsEntry ea, eb;
    double wa = 0.0, wb = 0.0;
    bool va = false, vb = false;
    if (M > 0) va = obs(0, ea, wa);
    int m = 0;
    for (; m + 1 < M; m += 2) {
      vb = obs(m + 1, eb, wb);
      if (va) add_entry(ea, wa, false);
      if (m + 2 < M) va = obs(m + 2, ea, wa);
      if (vb) add_entry(eb, wb, true);
    }
    if (m < M && va) add_entry(ea, wa, false);
  }
  if (SPLIT) {
    a.lx0 += b.lx0; a.lx1 += b.lx1; a.h00 += b.h00; a.h01 += b.h01; a.h11 += b.h11;
  }
}

// Constraints::get_state_cost / get_control_cost for one step (I/Constraints.cpp:145-227, 86-137; obstacles
// I/Obstacle.cpp:39-112) plus the six non-trivial Jacobian entries at (v_{t+1}, theta_{t+1}, a_t) (I/iLQR.cpp:102-106,
// I/Model.cpp:100-155).  (px,py,v,ct,st): state t with cos/sin of its heading; (vn,cn,sn): speed and cos/sin heading of
// state t+1; (cx,cy): closest path sample.  `obs(m, e, w)` supplies obstacle m at this step and its weight.
// Returns the stage cost of get_J.
// CULL (one wavefront per solve only: the lanes of a vote are the timesteps of ONE trajectory): an entry whose barrier
// exponent q2·c is below -64 on both ego circles for every step of the wavefront contributes less than e^-64 ≈ 1.6e-28 times
// O(10) factors to any sum — it is skipped after the 22 instructions that establish this, before its two exponentials.  The
// gradient and Hessian sums it would have been added to are O(1e-3 … 1e3): the omission is below 1e-26 absolute, i.e. far
// below one ulp of anything it feeds (measured: max|ΔU| against the oracle unchanged).  NaN exponents never vote to skip.
// LANE_EXACT (with CULL): a step at which the entry is negligible adds exactly nothing, whatever the other steps of the vote need
// (its weight is taken as 0) — the record of a step then depends on that step alone, not on which steps share its wavefront, so
// the two-wavefront kernel, whose votes span other sets of steps, produces the same bits (cilqr_solve.hip).
// The uncertainty-map term (I/Constraints.cpp:188-201) is NOT added here: the kernels add it to the stored record in a loop of
// its own (unc_cost_add — after the obstacle terms, i.e. in the reference's order of summation), which keeps its registers out of
// the obstacle loop's allocation (inlined here it cost the table-streaming configuration a third of its speed).
// PAIRED (with CULL): entries are taken two at a time with four entries' loads in flight — for obstacle tables streamed from
// global memory, where it is worth 12 % (config 3 materialised: 7.6 → 6.6 ms); where the entries come from LDS the extra live
// registers cost more than the overlap brings (config 2 +2.4 %, config 3 compact +2 %), so it is off there.
template <bool CULL = false, bool PAIRED = false, bool LANE_EXACT = false, bool SPLIT = !PAIRED, typename ObsAt>
__device__ __forceinline__ double lin_step(const KParams& kp, double px, double py, double v, double ct, double st, double u0,
                                           double u1, double vn, double cn, double sn, double cx, double cy, int M, ObsAt obs,
                                           Rec& r) {
  // --- tracking cost (I/Constraints.cpp:163-174)
  const double dx = px - cx, dy = py - cy, dv = v - kp.desired_speed;
  const double lx2 = (2 * kp.w_vel) * dv;
  const double J = stage_cost(kp, dx, dy, dv, u0, u1);

  // --- obstacles (I/Constraints.cpp:177-187, I/Obstacle.cpp:39-112): summed from zero, joined with the tracking terms below
  StepSums a{0.0, 0.0, 0.0, 0.0, 0.0};
  // (SPLIT: obstacle_loop; one chain where the entries are taken in pairs from a streamed table and in the kernels that deal a step's
  // entries to several wavefronts by obstacle — those instantiations have no registers left for a second set of sums)
  obstacle_loop<CULL, PAIRED, LANE_EXACT, SPLIT>(make_obs_consts(kp, px, py, ct, st), M, obs, a);

  // --- control cost (I/Constraints.cpp:110-131)
  double a1, a2, a3, a4;
  ctrl_args(kp, u0, u1, v, a1, a2, a3, a4);
  const double e1 = exp_fast(a1);
  const double e2 = exp_fast(a2);
  const double e3 = exp_fast(a3);
  const double e4 = exp_fast(a4);
  state_terms(kp, dx, dy, a, r.lx0, r.lx1, r.l00, r.l01, r.l11);
  r.lx2 = lx2;
  ctrl_terms(kp, u0, u1, e1, e2, e3, e4, r);

  // --- A/B entries
  ab_terms(kp, u0, vn, cn, sn, r);
  return J;
}

// ---- regularised Q_uu inverse ----------------------------------------------------------------------------------------
// V diag(1/(max(eig,0)+lamb)) V' of the symmetric 2×2 [[a,b],[b,d]] (I/iLQR.cpp:155-175).
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

// ---- one step of the backward recursion ----------------------------------------------------------------------------------
// Value function carried by the recursion: V_x and the upper triangle of the symmetric V_xx.
struct Value {
  double x0, x1, x2, x3;
  double v00, v01, v02, v03, v11, v12, v13, v22, v23, v33;
};
// :108-113: terminal value = stage N-1
__device__ __forceinline__ void value_terminal(Value& V, const Rec& c, double two_wvel) {
  V.x0 = c.lx0; V.x1 = c.lx1; V.x2 = c.lx2; V.x3 = 0.0;
  V.v00 = c.l00; V.v01 = c.l01; V.v02 = 0.0; V.v03 = 0.0; V.v11 = c.l11; V.v12 = 0.0; V.v13 = 0.0;
  V.v22 = two_wvel; V.v23 = 0.0; V.v33 = 0.0;
}
struct Gains {
  double g[KR];  // k0, k1, K00..K03, K10..K13
};

// One step of iLQR::backward_pass (I/iLQR.cpp:133-191).
//
// With fx = [[1,0,0,0],[0,1,0,0],[al,be,1,0],[ga,de,0,1]] and fu = [[p,q,dt,0],[0,0,0,dt]] (the reference's
// stored-transposed Jacobians, I/Model.cpp:100-155) the products of :149-153 reduce to one or two fused multiply-adds per
// entry.  V_xx, Q_xx and Q_uu are carried as symmetric matrices (the reference computes both triangles, which agree to
// rounding).
// FAST: branch-free positive-semi-definite form; `ok` reports whether this step's Q_uu passed the PSD test (false also
// for NaN) — callers hand solves with a failed step to the GENERAL kernel.  !FAST: branching form; `ok` false only for a
// non-finite Q_uu (the reference's EigenSolver cannot give a real decomposition there), V and gains then untouched.
template <bool FAST>
__device__ __forceinline__ void riccati_step(const Rec& c, Value& V, double dt, double two_wvel, double lamb, Gains& out, bool& ok) {
  const double al = c.al, be = c.be, ga = c.ga, de = c.de, p = c.p, q = c.q;
  const double x0 = V.x0, x1 = V.x1, x2 = V.x2, x3 = V.x3;
  const double v00 = V.v00, v01 = V.v01, v02 = V.v02, v03 = V.v03, v11 = V.v11, v12 = V.v12, v13 = V.v13;
  const double v22 = V.v22, v23 = V.v23, v33 = V.v33;

  // Q_x = l_x + fx V_x ; Q_u = l_u + fu V_x (:149-150)
  const double qx0 = c.lx0 + x0;
  const double qx1 = c.lx1 + x1;
  const double qx3 = fma(ga, x0, fma(de, x1, x3));
  double qx2, qu0;
  if (FAST) {
    // Row 0 of fu is (p, q, dt, 0) = (dt/2)·(row 2 of fx + e_2) — p = (dt/2)·al, q = (dt/2)·be by their definitions
    // (I/Model.cpp:139-155 against :100-127) — so every product with it follows from the matching row-2 product of fx with one
    // add and one multiply instead of three multiply-adds: here, and for E's first row and Q_uu(0,0) below (6 instructions per
    // step in all; equal to the direct products to rounding).
    const double s2 = fma(al, x0, fma(be, x1, x2));
    qx2 = s2 + c.lx2;
    qu0 = fma(0.5 * dt, s2 + x2, c.lu0);
  } else {
    qx2 = fma(al, x0, fma(be, x1, x2 + c.lx2));
    qu0 = fma(p, x0, fma(q, x1, fma(dt, x2, c.lu0)));
  }
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
  double e00, e01, e02, e03;
  if (FAST) {
    const double hd = 0.5 * dt;  // e0j = (dt/2)·(t2j + v2j)
    e00 = hd * (t20 + v02); e01 = hd * (t21 + v12); e02 = hd * (t22 + v22); e03 = hd * (t23 + v23);
  } else {
    e00 = fma(p, v00, fma(q, v01, dt * v02));
    e01 = fma(p, v01, fma(q, v11, dt * v12));
    e02 = fma(p, v02, fma(q, v12, dt * v22));
    e03 = fma(p, v03, fma(q, v13, dt * v23));
  }
  const double e10 = dt * v03, e11 = dt * v13, e12 = dt * v23, e13 = dt * v33;
  const double ux02 = fma(al, e00, fma(be, e01, e02));
  const double ux03 = fma(ga, e00, fma(de, e01, e03));
  const double ux12 = fma(al, e10, fma(be, e11, e12));
  const double ux13 = fma(ga, e10, fma(de, e11, e13));
  const double a = FAST ? fma(0.5 * dt, ux02 + e02, c.luu0) : fma(p, e00, fma(q, e01, fma(dt, e02, c.luu0)));
  const double b = dt * e03;
  const double d = fma(dt, e13, c.luu1);

  double i00, i01, i11;
  if (FAST) {
    const double bb = b * b;
    const double det0 = fma(a, d, -bb);
    ok = (det0 >= 0.0) & (a + d >= 0.0);
    quu_inverse_psd(a, b, d, lamb, bb, i00, i01, i11);
  } else {
    ok = quu_inverse_general(a, b, d, lamb, i00, i01, i11);
    if (!ok) return;
  }

  // k = -Qinv Q_u ; K = -Qinv Q_ux (:177-178)
  const double k0 = fma(-i00, qu0, -(i01 * qu1));
  const double k1 = fma(-i01, qu0, -(i11 * qu1));
  const double K00 = fma(-i00, e00, -(i01 * e10)), K01 = fma(-i00, e01, -(i01 * e11));
  const double K02 = fma(-i00, ux02, -(i01 * ux12)), K03 = fma(-i00, ux03, -(i01 * ux13));
  const double K10 = fma(-i01, e00, -(i11 * e10)), K11 = fma(-i01, e01, -(i11 * e11));
  const double K12 = fma(-i01, ux02, -(i11 * ux12)), K13 = fma(-i01, ux03, -(i11 * ux13));

  // G = K' Q_uu (unregularised) ; V_x = Q_x - G k ; V_xx = Q_xx - G K (:180-181)
  double g00, g01, g10, g11, g20, g21, g30, g31;
  if (FAST) {
    // In the positive-semi-definite case the regularised inverse is inv(Q_uu + lamb I) itself, so
    //   Q_uu K = -(Q_uu + lamb I - lamb I) inv(Q_uu + lamb I) Q_ux = -Q_ux - lamb K,   i.e.  G = -(Q_ux + lamb K)':
    // one fused multiply-add per entry instead of two products (8 instructions less on the serial chain of every step), and
    // without the cancellation of the direct product (Q_uu K ≈ -Q_ux for small lamb).  Agrees with the reference's product to
    // rounding; the GENERAL kernel, where clamped eigenvalues break the identity, keeps the product.
    g00 = fma(-lamb, K00, -e00); g01 = fma(-lamb, K10, -e10);
    g10 = fma(-lamb, K01, -e01); g11 = fma(-lamb, K11, -e11);
    g20 = fma(-lamb, K02, -ux02); g21 = fma(-lamb, K12, -ux12);
    g30 = fma(-lamb, K03, -ux03); g31 = fma(-lamb, K13, -ux13);
  } else {
    g00 = fma(K00, a, K10 * b); g01 = fma(K00, b, K10 * d);
    g10 = fma(K01, a, K11 * b); g11 = fma(K01, b, K11 * d);
    g20 = fma(K02, a, K12 * b); g21 = fma(K02, b, K12 * d);
    g30 = fma(K03, a, K13 * b); g31 = fma(K03, b, K13 * d);
  }
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

  out.g[0] = k0; out.g[1] = k1;
  out.g[2] = K00; out.g[3] = K01; out.g[4] = K02; out.g[5] = K03;
  out.g[6] = K10; out.g[7] = K11; out.g[8] = K12; out.g[9] = K13;
}

}  // namespace dev
}  // namespace cilqr
