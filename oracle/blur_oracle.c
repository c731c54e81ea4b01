/*
 * blur_oracle.c — CPU restatement of the reference's uncertainty propagation ("blur") over the vehicle-frame costmap.
 * TEST INFRASTRUCTURE ONLY (see cilqr_oracle.h).  Pinned by oracle/_ref/libref_gridmap.so::ref_blur, which runs the same
 * recipe over the reference's own grid_map_core (EllipseIterator) and vendored Eigen (EigenSolver<Matrix2f>).
 *
 * Reference: M/src/arbitrary_transformation.cu:8-157 (thrust_propagateUncertainty), M/include/ARBIT.cuh:51-107 (functors,
 * nomal2), M/src/local_costmap.cpp:474-498 (copy-through when the ellipse is empty), G/grid_map_core/src/iterators/
 * EllipseIterator.cpp:18-109, SubmapIterator (GridMapMath.cpp:459-488), boundPositionToRange (GridMapMath.cpp:240-263).
 * M/ = CILQR/src/map_engine/, G/ = CILQR/src/grid_map/.
 */
#include <float.h>
#include <math.h>

#include "cilqr_oracle.h"
#ifdef _OPENMP
#include <omp.h>
#endif

/* Eigen::EigenSolver<Matrix2f> on the symmetric covariance [[a,b],[b,c]] (cast to float), then
 * pseudoEigenvalueMatrix() / pseudoEigenvectors() (the UN-normalised m_eivec), as used at arbitrary_transformation.cu:60-83
 * and ARBIT.cuh:82-99.  Same algorithm as oracle_quu_inverse (RealSchur.h:246-392, EigenSolver.h:370-600), in float. */
void oracle_blur_ellipse(double a, double b, double c, double* half_major, double* half_minor, double* angle) {
  float t00 = (float)a, t10 = (float)b, t01 = (float)b, t11 = (float)c;
  float u00 = 1, u10 = 0, u01 = 0, u11 = 1;
  const float eps = FLT_EPSILON;
  float norm = fabsf(t00) + fabsf(t10) + fabsf(t01) + fabsf(t11);
  int complex_pair = 0;
  if (norm != 0) {
    float s = fabsf(t00) + fabsf(t11);
    if (fabsf(t10) <= eps * s) {
      t10 = 0;
    } else {
      float pp = 0.5f * (t00 - t11);
      float q = pp * pp + t10 * t01;
      if (q >= 0) {
        float z = sqrtf(fabsf(q));
        float gp = (pp >= 0) ? pp + z : pp - z, gq = t10, cc, sn;
        if (gq == 0) { cc = gp < 0 ? -1.f : 1.f; sn = 0; }
        else if (gp == 0) { cc = 0; sn = gq < 0 ? 1.f : -1.f; }
        else if (fabsf(gp) > fabsf(gq)) {
          float tt = gq / gp, uu = sqrtf(1.f + tt * tt);
          if (gp < 0) uu = -uu;
          cc = 1.f / uu; sn = -tt * cc;
        } else {
          float tt = gp / gq, uu = sqrtf(1.f + tt * tt);
          if (gq < 0) uu = -uu;
          sn = -1.f / uu; cc = -tt * sn;
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
        t10 = 0;
      } else {
        complex_pair = 1; /* cannot happen for b*b >= 0; kept for NaN inputs */
      }
    }
  }
  float d0 = t00, d1 = t11; /* pseudoEigenvalueMatrix diagonal */
  if (!complex_pair) {
    float norm2 = fabsf(t00) + fabsf(t01) + fabsf(t10) + fabsf(t11);
    if (norm2 != 0.0f) {
      t11 = 1.0f;
      {
        float w = t00 - d1;
        float r = t01 * t11;
        if (w != 0.0f) t01 = -r / w; else t01 = -r / (eps * norm2);
        float tt = fabsf(t01);
        if ((eps * tt) * tt > 1) { t01 /= tt; t11 /= tt; }
      }
      t00 = 1.0f;
      float n01 = u00 * t01 + u01 * t11, n11 = u10 * t01 + u11 * t11;
      u01 = n01; u11 = n11;
      u00 = u00 * t00; u10 = u10 * t00;
    }
  }
  /* arbitrary_transformation.cu:73-82 */
  int major = (d0 > d1) ? 0 : 1, minor = 1 - major;
  /* ellipse_params_functor: angle from ROW `major` of V (ARBIT.cuh:89) */
  float v_m0 = major == 0 ? u00 : u10, v_m1 = major == 0 ? u01 : u11;
  double ang = atan2((double)v_m1, (double)v_m0);
  if (ang < 0) ang += 6.28318530718;
  const double chisquare_val = 2.4477;
  *angle = ang;
  *half_major = chisquare_val * sqrt((double)(major == 0 ? d0 : d1));
  *half_minor = chisquare_val * sqrt((double)(minor == 0 ? d0 : d1));
}

/* boundPositionToRange, one axis (GridMapMath.cpp:240-263) */
static double bound_axis(double position, double map_len, double map_pos) {
  double v2o = 0.5 * map_len;
  double shifted = position - map_pos + v2o;
  double epsilon = 10.0 * DBL_EPSILON;
  if (fabs(position) > 1.0) epsilon *= fabs(position);
  if (shifted <= 0) shifted = epsilon;
  else if (shifted >= map_len) shifted = map_len - epsilon;
  return shifted + map_pos - v2o;
}

long oracle_blur(const float* src, const cilqr_map_geom* g, int index, double sin_t, double cos_t, double sigma_x,
                 double sigma_y, double sigma_theta, float* out, int* count_out, int threads) {
  const long n = (long)g->rows * g->cols;
  if (threads < 1) threads = 1;
  long empty = 0;
  for (long lin = 0; lin < index && lin < n; lin++) out[lin] = NAN; /* layer was cleared by setGeometry; never written */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads) reduction(+ : empty)
#endif
  for (long lin = index; lin < n; lin++) {
    const int ci = (int)(lin % g->rows), cj = (int)(lin / g->rows);
    double Cx, Cy;
    oracle_map_get_position(g, ci, cj, &Cx, &Cy);
    /* uncertainty_error_functor, ARBIT.cuh:59-68 */
    const double u = (-sin_t * Cx - cos_t * Cy) * (-sin_t * Cx - cos_t * Cy);
    const double v = (cos_t * Cx - sin_t * Cy) * (cos_t * Cx - sin_t * Cy);
    const double t = sin_t * cos_t * (Cx * Cx - Cy * Cy) + Cx * Cy * (sin_t * sin_t - cos_t * cos_t);
    const double sxi = sqrt(sigma_x * sigma_x + sigma_theta * sigma_theta * u);
    const double syi = sqrt(sigma_y * sigma_y + sigma_theta * sigma_theta * v);
    const double rho = sigma_theta * sigma_theta * t / (sxi * syi);
    /* abc_functor :74-79 */
    const double a = sxi * sxi, b = rho * sxi * syi, c = syi * syi;
    double hmaj, hmin, angle;
    oracle_blur_ellipse(a, b, c, &hmaj, &hmin, &angle);

    /* EllipseIterator(map, position, Length(2*hmaj, 2*hmin), angle) */
    const double len0 = 2 * hmaj, len1 = 2 * hmin;
    const double semi0 = (0.5 * len0) * (0.5 * len0), semi1 = (0.5 * len1) * (0.5 * len1);
    const double sinR = sin(angle), cosR = cos(angle);
    double numerator = 0, denominator = 0;
    int count = 0;
    if (len0 == len0 && len1 == len1) { /* NaN axes: the reference's iterator visits nothing */
      /* findSubmapParameters :92-107 */
      const double ux = cosR * len0 - sinR * 0.0, uy = sinR * len0 + cosR * 0.0;
      const double vx = cosR * 0.0 - sinR * len1, vy = sinR * 0.0 + cosR * len1;
      const double bbx = sqrt(ux * ux + vx * vx), bby = sqrt(uy * uy + vy * vy);
      const double tlx = bound_axis(Cx + bbx, g->len_x, g->pos_x), tly = bound_axis(Cy + bby, g->len_y, g->pos_y);
      const double brx = bound_axis(Cx - bbx, g->len_x, g->pos_x), bry = bound_axis(Cy - bby, g->len_y, g->pos_y);
      int i0, j0, i1, j1;
      oracle_map_get_index(g, tlx, tly, &i0, &j0);
      oracle_map_get_index(g, brx, bry, &i1, &j1);
      const int ni = i1 - i0 + 1, nj = j1 - j0 + 1;
      /* SubmapIterator order: column index fastest (GridMapMath.cpp:467-475); cells outside the map cannot occur after
       * the bounding above */
      for (int di = 0; di < ni; di++)
        for (int dj = 0; dj < nj; dj++) {
          const int ii = i0 + di, jj = j0 + dj;
          double x, y;
          if (!oracle_map_get_position(g, ii, jj, &x, &y)) continue;
          /* isInside :84-90: transform [[cos, sin],[sin, -cos]] */
          const double dx = x - Cx, dy = y - Cy;
          const double tx = cosR * dx + sinR * dy, ty = sinR * dx + (-cosR) * dy;
          const double value = tx * tx / semi0 + ty * ty / semi1;
          if (!(value <= 1)) continue;
          /* nomal2, ARBIT.cuh:103-107 */
          const double f = 1.0 / (sqrt(1 - rho * rho) * (2 * M_PI * sxi * syi)) *
                           exp((-1 / (2 * (1 - rho * rho))) *
                               ((x - Cx) * (x - Cx) / (sxi * sxi) - 2 * rho * (x - Cx) * (y - Cy) / (sxi * syi) +
                                (y - Cy) * (y - Cy) / (syi * syi)));
          numerator += f * (double)src[(size_t)jj * g->rows + ii];
          denominator += f;
          count++;
        }
    }
    if (count == 0) { out[lin] = src[lin]; empty++; } /* local_costmap.cpp:489-493 */
    else out[lin] = (float)(numerator / denominator);
    if (count_out) count_out[lin] = count;
  }
  return empty;
}
