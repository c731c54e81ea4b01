/*
 * uncertainty_oracle.c — CPU statement of the costmap-lookup uncertainty cost.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the reference's class Uncertainty (get_uncertainty_cost, called at I/Constraints.cpp:193) is absent from
 * the reference repository, it holds no fixture for it, and nothing can be run to produce one.  The arithmetic is
 * the one include/cilqr.h defines at cilqr_set_uncertainty_map, written here independently of the HIP code, in plain C with
 * glibc libm and no contraction.  What IS pinned to the reference: the bilinear lookup, against the reference's own
 * GridMap::atPositionLinearInterpolated built in place (oracle/_ref, tests/golden/ref_gridmap_linear.json), and the call site
 * (weight w_uncertainty, l_x / l_xx only, not J: I/Constraints.cpp:188-201, :553-557).
 */
#include <math.h>
#include <stddef.h>

#include "cilqr_oracle.h"

/* Bilinear interpolation of a column-major float32 layer over the four cell centres around (qx, qy), evaluated in double
 * (the arithmetic of G/grid_map_core/src/GridMap.cpp:770-837 without its final rounding to float), with the interpolant's
 * gradient.  Cell (i, j) has its centre at first - res·(i, j) (GridMapMath.cpp:114-127).  Returns 0 when the four cells are not
 * all inside the map and finite. */
int oracle_layer_bilinear(const float* layer, const cilqr_map_geom* g, double qx, double qy, double* value, double* d_dx,
                          double* d_dy) {
  const double x_first = g->pos_x + (0.5 * g->len_x - 0.5 * g->res);
  const double y_first = g->pos_y + (0.5 * g->len_y - 0.5 * g->res);
  const double inv_res = 1.0 / g->res;
  const double fi = (x_first - qx) * inv_res, fj = (y_first - qy) * inv_res;
  if (!(fi >= 0.0) || !(fj >= 0.0) || !(fi < (double)(g->rows - 1)) || !(fj < (double)(g->cols - 1))) return 0;
  const int i0 = (int)fi, j0 = (int)fj;
  const double ti = fi - (double)i0, tj = fj - (double)j0;
  const double f00 = layer[(size_t)j0 * g->rows + i0], f10 = layer[(size_t)j0 * g->rows + i0 + 1];
  const double f01 = layer[(size_t)(j0 + 1) * g->rows + i0], f11 = layer[(size_t)(j0 + 1) * g->rows + i0 + 1];
  if (!isfinite(f00) || !isfinite(f10) || !isfinite(f01) || !isfinite(f11)) return 0;
  const double a0 = f00 + ti * (f10 - f00), a1 = f01 + ti * (f11 - f01); /* along i at j0, j0+1 */
  *value = a0 + tj * (a1 - a0);
  const double di = (f10 - f00) + tj * ((f11 - f01) - (f10 - f00)); /* d/d fi */
  const double dj = a1 - a0;                                         /* d/d fj */
  *d_dx = -di * inv_res; /* fi falls as x grows */
  *d_dy = -dj * inv_res;
  return 1;
}

/* The map cost at one state for solve b of a batch (host pointers in *m): cost, vx (4), mx (4×4 column-major). */
void oracle_uncertainty_cost(const cilqr_params* p, const cilqr_uncertainty_map* m, int b, const double* state, double* cost,
                             double* vx4, double* mx16) {
  const float* layer = m->layer + (size_t)b * (size_t)m->layer_stride;
  const double px = m->poses ? m->poses[3 * (size_t)b] : m->pose_x, py = m->poses ? m->poses[3 * (size_t)b + 1] : m->pose_y;
  const double pth = m->poses ? m->poses[3 * (size_t)b + 2] : m->pose_theta;
  const double cp = cos(pth), sp = sin(pth);
  const double ct = cos(state[3]), st = sin(state[3]);
  const int nl = m->probes_l, nw = m->probes_w;
  const double la0 = nl > 1 ? -0.5 * p->safe_length : 0.0, la_step = nl > 1 ? p->safe_length / (double)(nl - 1) : 0.0;
  const double wb0 = nw > 1 ? -0.5 * p->safe_width : 0.0, wb_step = nw > 1 ? p->safe_width / (double)(nw - 1) : 0.0;
  double x = 0.0, gx = 0.0, gy = 0.0, hxx = 0.0, hxy = 0.0, hyy = 0.0;
  for (int k = 0; k < nl; k++) {
    const double a = la0 + (double)k * la_step;
    for (int l = 0; l < nw; l++) {
      const double bb = wb0 + (double)l * wb_step;
      const double Px = state[0] + (a * ct - bb * st), Py = state[1] + (a * st + bb * ct);
      const double dx = Px - px, dy = Py - py;
      const double qx = cp * dx + sp * dy, qy = cp * dy - sp * dx;
      double o, ox, oy;
      if (!oracle_layer_bilinear(layer, &m->geom, qx, qy, &o, &ox, &oy)) continue;
      const double c = o * 0.01 - 1.0;
      const double e = p->q1_uncertainty * exp(p->q2_uncertainty * c);
      const double cqx = ox * 0.01, cqy = oy * 0.01;                   /* grad of c in the vehicle frame */
      const double cX = cp * cqx - sp * cqy, cY = sp * cqx + cp * cqy; /* … in the planning frame */
      const double sv = p->q2_uncertainty * e, sm = p->q2_uncertainty * p->q2_uncertainty * e;
      x += e;
      gx += sv * cX;
      gy += sv * cY;
      hxx += (sm * cX) * cX;
      hxy += (sm * cX) * cY;
      hyy += (sm * cY) * cY;
    }
  }
  const double inv = 1.0 / (double)(nl * nw);
  *cost = x * inv;
  for (int i = 0; i < 4; i++) vx4[i] = 0.0;
  for (int i = 0; i < 16; i++) mx16[i] = 0.0;
  vx4[0] = gx * inv;
  vx4[1] = gy * inv;
  mx16[0] = hxx * inv;
  mx16[1] = hxy * inv;
  mx16[4] = hxy * inv;
  mx16[5] = hyy * inv;
}
