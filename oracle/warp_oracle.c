/*
 * warp_oracle.c — CPU restatement of the reference costmap warp.  TEST INFRASTRUCTURE ONLY
 * (see cilqr_oracle.h).  Pinned by oracle/_ref (the reference's grid_map_core compiled as-is) and by the
 * known answers of G/grid_map_core/test/{GridMapMathTest,GridMapTest}.cpp.
 *
 * M/ = CILQR/src/map_engine/, G/ = CILQR/src/grid_map/.
 */
#include "cilqr_oracle.h"

#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* GridMap::setGeometry, G/grid_map_core/src/GridMap.cpp:45-62 */
void oracle_map_geom_set(cilqr_map_geom* g, double len_x, double len_y, double res, double pos_x, double pos_y) {
  g->rows = (int)round(len_x / res);
  g->cols = (int)round(len_y / res);
  g->res = res;
  g->len_x = (double)g->rows * res;
  g->len_y = (double)g->cols * res;
  g->pos_x = pos_x;
  g->pos_y = pos_y;
}

/* getPositionFromIndex, G/grid_map_core/src/GridMapMath.cpp:114-127 (start index at default position) */
int oracle_map_get_position(const cilqr_map_geom* g, int i, int j, double* px, double* py) {
  if (!(i >= 0 && j >= 0 && i < g->rows && j < g->cols)) return 0;
  double off_x = 0.5 * g->len_x - 0.5 * g->res; /* getVectorToFirstCell :43-53 */
  double off_y = 0.5 * g->len_y - 0.5 * g->res;
  *px = (g->pos_x + off_x) + g->res * (double)(-i);
  *py = (g->pos_y + off_y) + g->res * (double)(-j);
  return 1;
}

/* getIndexFromPosition, GridMapMath.cpp:129-142; checkIfPositionWithinMap :144-158 */
int oracle_map_get_index(const cilqr_map_geom* g, double px, double py, int* i, int* j) {
  double off_x = 0.5 * g->len_x, off_y = 0.5 * g->len_y;
  double vx = ((px - off_x) - g->pos_x) / g->res;
  double vy = ((py - off_y) - g->pos_y) / g->res;
  /* transformMapFrameToBufferOrder(const Vector&): Index{-v[0], -v[1]} — double→int truncation (:70-72) */
  double nx = -vx, ny = -vy;
  int ii, jj;
  /* keep the conversion defined for wild positions; such positions fail the range test below anyway */
  if (!(nx > -2e9 && nx < 2e9)) ii = -1; else ii = (int)nx;
  if (!(ny > -2e9 && ny < 2e9)) jj = -1; else jj = (int)ny;
  *i = ii;
  *j = jj;
  double tx = -1.0 * ((px - g->pos_x) - off_x);
  double ty = -1.0 * ((py - g->pos_y) - off_y);
  int within = (tx >= 0.0 && ty >= 0.0 && tx < g->len_x && ty < g->len_y);
  int inrange = (ii >= 0 && jj >= 0 && ii < g->rows && jj < g->cols);
  return within && inrange;
}

/* LocalCostmap::odomCallback warp loop, M/src/local_costmap.cpp:242-264.  GridMapIterator visits linear
 * indices 0..rows*cols-1 with index = (lin % rows, lin / rows) (GridMapMath.cpp:514-518). */
long oracle_warp_costmap(const float* src, const cilqr_map_geom* sg, float* dst, const cilqr_map_geom* dg,
                         double vx, double vy, double vtheta, const float* bbox, int threads) {
  const double s = sin(vtheta), c = cos(vtheta); /* :201-202 */
  const long n = (long)dg->rows * dg->cols;
  long oob = 0;
  if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads) reduction(+ : oob)
#endif
  for (long lin = 0; lin < n; lin++) {
    int i = (int)(lin % dg->rows), j = (int)(lin / dg->rows);
    double Cx, Cy;
    oracle_map_get_position(dg, i, j, &Cx, &Cy);
    double x_og = (Cx * c - Cy * s) + vx; /* :249-250 */
    double y_og = (Cx * s + Cy * c) + vy;
    int si, sj;
    float v;
    if (oracle_map_get_index(sg, x_og, y_og, &si, &sj)) {
      v = src[(size_t)sj * sg->rows + si]; /* column-major Eigen::MatrixXf */
    } else {
      v = NAN; /* reference: atPosition throws std::out_of_range (GridMap.cpp:160-166) */
      oob++;
    }
    if (bbox && bbox[lin] > 90) v = bbox[lin]; /* :260-263 */
    dst[lin] = v;
  }
  return oob;
}
