// costmap_blur.hip — pose-uncertainty propagation ("blur") over the vehicle-frame costmap for gfx950 (MI355X).
//
// Reference: thrust_propagateUncertainty (M/src/arbitrary_transformation.cu:8-157) + its functors and nomal2
// (M/include/ARBIT.cuh:51-107) + the copy-through of M/src/local_costmap.cpp:483-496.  The reference runs it as three
// thrust::for_each launches, a serial host loop of 2×2 float eigen-solves, six device↔host vector copies and an OpenMP loop
// over grid_map::EllipseIterator.  Here it is ONE kernel, one lane per destination cell:
//   1. per-cell pose-uncertainty covariance (fp64): sigma_x_i, sigma_y_i, rho  → a, b, c
//   2. 2×2 eigen-decomposition in float, following Eigen::EigenSolver<Matrix2f> step for step with contraction off, so
//      that the confidence ellipse (half axes 2.4477·sqrt(eig), angle from ROW `major` of the un-normalised eigenvector
//      matrix — the reference's indexing, ARBIT.cuh:89) is bit-identical, NaN axes for slightly negative float eigenvalues
//      included
//   3. the EllipseIterator walk (G/grid_map_core/src/iterators/EllipseIterator.cpp:84-107): bounding box of the rotated
//      FULL lengths, clamped into the map (boundPositionToRange), inside test ((T d)²/semi² summed ≤ 1), and the
//      Gaussian-weighted average  Σ f_j·map_j / Σ f_j  with the bivariate normal density f of nomal2.
// Source reads are 4-byte gathers from a window of a few cells around the lane's own cell: neighbouring lanes (adjacent i)
// read overlapping windows, served by L1/L2.  The inside test uses reciprocal semi-axes and falls back to the reference's
// exact divisions only within 1e-9 of the boundary; the density uses hoisted reciprocals (results differ from the
// reference's association by ~1e-16 relative, i.e. at most one float32 ulp after the final cast, and rarely that).
// Cells the reference reaches only through an uninitialised Position (its submap can include the non-existent row/column
// `size` when the box is clamped onto the far map edge, EllipseIterator.cpp:86-88 with GridMapMath.cpp:122) are skipped.
#include <float.h>

#include "cilqr_internal.h"
#include "costmap_cells.hpp"

namespace cilqr {

namespace {

struct Ellipse {
  double half_major, half_minor, angle;
};

// Correctly rounded float square root.  v_sqrt_f32 is a 1-ulp instruction and hipcc emits it bare for sqrtf/__fsqrt_rn; the
// fp64 square root is correctly rounded, and rounding it to float is again correctly rounded (53 ≥ 2·24 + 2).
__device__ __forceinline__ float sqrt_f32_rn(float x) { return (float)sqrt((double)x); }

// Eigen::EigenSolver<Matrix2f> on [[a,b],[b,c]] → pseudoEigenvalueMatrix / pseudoEigenvectors → ellipse parameters
// (M/src/arbitrary_transformation.cu:60-83, ARBIT.cuh:82-99; Eigen 3.2.10 RealSchur.h:246-392, EigenSolver.h:370-600).
__device__ __forceinline__ Ellipse ellipse_from_cov(double a, double b, double c) {
#pragma clang fp contract(off)
  float t00 = (float)a, t10 = (float)b, t01 = (float)b, t11 = (float)c;
  float u00 = 1.f, u10 = 0.f, u01 = 0.f, u11 = 1.f;
  const float eps = FLT_EPSILON;
  const float norm = fabsf(t00) + fabsf(t10) + fabsf(t01) + fabsf(t11);
  bool complex_pair = false;
  if (norm != 0.f) {
    const float s = fabsf(t00) + fabsf(t11);
    if (fabsf(t10) <= eps * s) {
      t10 = 0.f;
    } else {
      const float pp = 0.5f * (t00 - t11);
      const float q = pp * pp + t10 * t01;
      if (q >= 0.f) {
        const float z = sqrt_f32_rn(fabsf(q));
        const float gp = (pp >= 0.f) ? pp + z : pp - z, gq = t10;
        float cc, sn;
        if (gq == 0.f) { cc = gp < 0.f ? -1.f : 1.f; sn = 0.f; }
        else if (gp == 0.f) { cc = 0.f; sn = gq < 0.f ? 1.f : -1.f; }
        else if (fabsf(gp) > fabsf(gq)) {
          const float tt = __fdiv_rn(gq, gp);
          float uu = sqrt_f32_rn(1.f + tt * tt);
          if (gp < 0.f) uu = -uu;
          cc = __fdiv_rn(1.f, uu); sn = -tt * cc;
        } else {
          const float tt = __fdiv_rn(gp, gq);
          float uu = sqrt_f32_rn(1.f + tt * tt);
          if (gq < 0.f) uu = -uu;
          sn = __fdiv_rn(-1.f, uu); cc = -tt * sn;
        }
        if (!(cc == 1.f && -sn == 0.f)) {
          float x0 = t00, y0 = t10, x1 = t01, y1 = t11;
          t00 = cc * x0 - sn * y0; t10 = sn * x0 + cc * y0;
          t01 = cc * x1 - sn * y1; t11 = sn * x1 + cc * y1;
          x0 = t00; y0 = t01; x1 = t10; y1 = t11;
          t00 = cc * x0 - sn * y0; t01 = sn * x0 + cc * y0;
          t10 = cc * x1 - sn * y1; t11 = sn * x1 + cc * y1;
          x0 = u00; y0 = u01; x1 = u10; y1 = u11;
          u00 = cc * x0 - sn * y0; u01 = sn * x0 + cc * y0;
          u10 = cc * x1 - sn * y1; u11 = sn * x1 + cc * y1;
        }
        t10 = 0.f;
      } else {
        complex_pair = true;
      }
    }
  }
  const float d0 = t00, d1 = t11;
  if (!complex_pair) {
    const float norm2 = fabsf(t00) + fabsf(t01) + fabsf(t10) + fabsf(t11);
    if (norm2 != 0.f) {
      t11 = 1.f;
      const float w = t00 - d1;
      const float r = t01 * t11;
      if (w != 0.f) t01 = __fdiv_rn(-r, w); else t01 = __fdiv_rn(-r, eps * norm2);
      const float tt = fabsf(t01);
      if ((eps * tt) * tt > 1.f) { t01 = __fdiv_rn(t01, tt); t11 = __fdiv_rn(t11, tt); }
      t00 = 1.f;
      const float n01 = u00 * t01 + u01 * t11, n11 = u10 * t01 + u11 * t11;
      u01 = n01; u11 = n11;
      u00 = u00 * t00; u10 = u10 * t00;
    }
  }
  const int major = (d0 > d1) ? 0 : 1;
  const float v_m0 = major == 0 ? u00 : u10, v_m1 = major == 0 ? u01 : u11;  // ROW `major` of V
  double ang = atan2((double)v_m1, (double)v_m0);
  if (ang < 0) ang += 6.28318530718;
  Ellipse e;
  e.angle = ang;
  e.half_major = 2.4477 * sqrt((double)(major == 0 ? d0 : d1));
  e.half_minor = 2.4477 * sqrt((double)(major == 0 ? d1 : d0));
  return e;
}

// Per-cell pose-uncertainty covariance and its confidence ellipse.
__device__ __forceinline__ Ellipse cell_ellipse(const BlurArgs& a, double Cx, double Cy, double& sxi, double& syi, double& rho) {
#pragma clang fp contract(off)
  const double s = a.sin_t, c = a.cos_t;
  const double u = (-s * Cx - c * Cy) * (-s * Cx - c * Cy);
  const double v = (c * Cx - s * Cy) * (c * Cx - s * Cy);
  const double t = s * c * (Cx * Cx - Cy * Cy) + Cx * Cy * (s * s - c * c);
  sxi = sqrt(a.sigma_x * a.sigma_x + a.sigma_theta * a.sigma_theta * u);
  syi = sqrt(a.sigma_y * a.sigma_y + a.sigma_theta * a.sigma_theta * v);
  rho = a.sigma_theta * a.sigma_theta * t / (sxi * syi);
  return ellipse_from_cov(sxi * sxi, rho * sxi * syi, syi * syi);
}

// boundPositionToRange, one axis (G/grid_map_core/src/GridMapMath.cpp:240-263)
__device__ __forceinline__ double bound_axis(double position, double map_len, double map_pos) {
#pragma clang fp contract(off)
  const double v2o = 0.5 * map_len;
  double shifted = position - map_pos + v2o;
  double epsilon = 10.0 * DBL_EPSILON;
  if (fabs(position) > 1.0) epsilon *= fabs(position);
  if (shifted <= 0) shifted = epsilon;
  else if (shifted >= map_len) shifted = map_len - epsilon;
  return shifted + map_pos - v2o;
}

// getIndexFromPosition, one axis (GridMapMath.cpp:129-142): trunc(-((p - len/2) - pos)/res)
__device__ __forceinline__ int index_axis(double p, double map_len, double map_pos, double res) {
#pragma clang fp contract(off)
  const double n = -(((p - 0.5 * map_len) - map_pos) / res);
  return (n > -2e9 && n < 2e9) ? (int)n : -1;
}

// EllipseIterator::isInside (G/grid_map_core/src/iterators/EllipseIterator.cpp:84-90) for cell (ii, jj), every operation
// separate and the divisions carried out: the form the reference's compiler emits.
__device__ __forceinline__ double inside_value_exact(double x_first, double y_first, double res, int ii, int jj, double Cx, double Cy,
                                                     double cosR, double sinR, double semi0, double semi1) {
#pragma clang fp contract(off)
  const double dx = (x_first + res * (double)(-ii)) - Cx, dy = (y_first + res * (double)(-jj)) - Cy;
  const double tx = cosR * dx + sinR * dy, ty = sinR * dx - cosR * dy;
  return tx * tx / semi0 + ty * ty / semi1;
}

// LPC lanes per destination cell.  One lane per cell fills the chip from ≈ 65 k cells; the node's own map has 15 000, which
// leaves three quarters of the SIMDs idle while each lane walks its ellipse alone — there four adjacent lanes share a cell
// (rows of the ellipse's box dealt round-robin, the three sums combined by two shuffles; summation order changes by that).
template <int LPC>
__global__ __launch_bounds__(256) void blur_kernel(BlurArgs a) {
#pragma clang fp contract(off)  // cell centres, ellipse box and row offsets as the reference's compiler forms them: no fma
  const long n = (long)a.g.rows * a.g.cols;
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long lin = tid / LPC;
  const int sub = (int)(tid % LPC);
  if (lin >= n) return;  // the LPC lanes of a cell leave together
  if (lin < a.index) {
    if (sub == 0) {
      a.out[lin] = __builtin_nanf("");  // never written by the reference (layer cleared by setGeometry)
      if (a.occ_out) a.occ_out[n - 1 - lin] = (int8_t)-1;
    }
    return;
  }
  const int rows = a.g.rows, cols = a.g.cols;
  const int ci = (int)(lin % rows), cj = (int)(lin / rows);
  const double res = a.g.res;
  const double x_first = (a.g.pos_x + (0.5 * a.g.len_x - 0.5 * res));  // centre of cell row 0
  const double y_first = (a.g.pos_y + (0.5 * a.g.len_y - 0.5 * res));
  const double Cx = x_first + res * (double)(-ci), Cy = y_first + res * (double)(-cj);

  // uncertainty_error_functor / abc_functor (ARBIT.cuh:59-79), contraction off: a, b, c feed a FLOAT eigen-solve whose
  // outcome (which eigenvalue is the major one, the orientation of a near-circular ellipse) can hinge on their last bit
  double sxi, syi, rho;
  const Ellipse el = cell_ellipse(a, Cx, Cy, sxi, syi, rho);

  double numerator = 0.0, denominator = 0.0;
  int count = 0;
  const double len0 = 2 * el.half_major, len1 = 2 * el.half_minor;
  if (len0 == len0 && len1 == len1) {  // NaN axes: the reference's iterator visits nothing
    double sinR, cosR;
    sincos(el.angle, &sinR, &cosR);
    const double semi0 = (0.5 * len0) * (0.5 * len0), semi1 = (0.5 * len1) * (0.5 * len1);
    const double ux = cosR * len0, uy = sinR * len0, vx = -(sinR * len1), vy = cosR * len1;
    const double bbx = sqrt(ux * ux + vx * vx), bby = sqrt(uy * uy + vy * vy);
    const int i0 = index_axis(bound_axis(Cx + bbx, a.g.len_x, a.g.pos_x), a.g.len_x, a.g.pos_x, res);
    const int j0 = index_axis(bound_axis(Cy + bby, a.g.len_y, a.g.pos_y), a.g.len_y, a.g.pos_y, res);
    int i1 = index_axis(bound_axis(Cx - bbx, a.g.len_x, a.g.pos_x), a.g.len_x, a.g.pos_x, res);
    int j1 = index_axis(bound_axis(Cy - bby, a.g.len_y, a.g.pos_y), a.g.len_y, a.g.pos_y, res);
    i1 = min(i1, rows - 1);
    j1 = min(j1, cols - 1);
    // hoisted pieces of the inside test and of nomal2
    const double r0 = 1.0 / semi0, r1 = 1.0 / semi1;
    const double omr = 1 - rho * rho;
    const double pref = 1.0 / (sqrt(omr) * (2 * 3.14159265358979323846 * sxi * syi));
    const double kexp = -1 / (2 * omr);
    const double ixx = 1.0 / (sxi * sxi), ixy = 2 * rho / (sxi * syi), iyy = 1.0 / (syi * syi);
    for (int ii = max(i0, 0) + sub; ii <= i1; ii += LPC) {
      const double dx = (x_first + res * (double)(-ii)) - Cx;
      for (int jj = max(j0, 0); jj <= j1; ++jj) {
        double dy, value;
        {
#pragma clang fp contract(fast)  // the hot test, fused: decides membership only when it is clear of the boundary
          dy = (y_first + res * (double)(-jj)) - Cy;
          const double tx = cosR * dx + sinR * dy, ty = sinR * dx - cosR * dy;
          value = tx * tx * r0 + ty * ty * r1;
        }
        // within 1e-9 of the boundary (fusing and the hoisted reciprocals move `value` by ≈1e-16): the reference's own form,
        // unfused, with its divisions (EllipseIterator.cpp:84-90)
        if (fabs(value - 1.0) < 1e-9) value = inside_value_exact(x_first, y_first, res, ii, jj, Cx, Cy, cosR, sinR, semi0, semi1);
        if (!(value <= 1)) continue;
        {
#pragma clang fp contract(fast)  // the weights do not decide membership (the float32 output stays within the stated 1 ulp)
          const double f = pref * exp(kexp * (dx * dx * ixx - ixy * dx * dy + dy * dy * iyy));
          numerator += f * (double)a.src[(size_t)jj * rows + ii];
          denominator += f;
        }
        ++count;
      }
    }
  }
#pragma unroll
  for (int o = LPC >> 1; o > 0; o >>= 1) {
    numerator += __shfl_xor(numerator, o, 64);
    denominator += __shfl_xor(denominator, o, 64);
    count += __shfl_xor(count, o, 64);
  }
  if (sub != 0) return;
  const float blurred = count == 0 ? a.src[lin] : (float)(numerator / denominator);  // local_costmap.cpp:489-496
  a.out[lin] = blurred;
  // fused GridMapRosConverter::toOccupancyGrid of this layer (M/src/local_costmap.cpp:298), cell order reversed
  if (a.occ_out) a.occ_out[n - 1 - lin] = layer_to_cell(blurred, a.occ_min, a.occ_den);
  if (a.count_out) a.count_out[lin] = count;
}

}  // namespace

namespace {
__global__ void blur_ellipse_kernel(int n, const double* abc, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Ellipse e = ellipse_from_cov(abc[3 * i], abc[3 * i + 1], abc[3 * i + 2]);
  out[3 * i] = e.half_major; out[3 * i + 1] = e.half_minor; out[3 * i + 2] = e.angle;
}
}  // namespace

hipError_t launch_blur_ellipse(int n, const double* abc, double* out, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(blur_ellipse_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, abc, out);
  return hipGetLastError();
}

hipError_t launch_blur(const BlurArgs& a, hipStream_t stream) {
  const long n = (long)a.g.rows * a.g.cols;
  if (n <= 0) return hipSuccess;
  // measured on MI355X: 150×100 cells 55 µs with one lane per cell, 20.7 / 16.4 / 17.4 µs with 4 / 8 / 16; 256² cells 101 µs
  // against 58 µs with 4; from about a million cells one lane per cell already fills the chip
  const int lpc = n <= 20000 ? 8 : n <= 300000 ? 4 : 1;
  if (lpc == 8) hipLaunchKernelGGL(blur_kernel<8>, dim3((unsigned)((8 * n + 255) / 256)), dim3(256), 0, stream, a);
  else if (lpc == 4) hipLaunchKernelGGL(blur_kernel<4>, dim3((unsigned)((4 * n + 255) / 256)), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(blur_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace cilqr
