/*
 * cilqr_oracle.c — CPU restatement of the reference CILQR solver path.  TEST INFRASTRUCTURE ONLY
 * (see cilqr_oracle.h for who may use it and how it is pinned).
 *
 * Plain C99, fp64, dense small-matrix loops written in the reference's own evaluation order
 * (including its multiplications by structural zeros/ones), glibc libm for pow/exp/sin/cos/tan.
 * Every function cites the reference lines it follows; I/ = CILQR/src/ilqr/include/ilqr/.
 */
#include "cilqr_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * Parameters — I/Parameters.cpp:3-75, I/iLQR.cpp:17-18
 * ---------------------------------------------------------------------------------------------- */
void oracle_params_default(cilqr_params* p) {
  memset(p, 0, sizeof(*p));
  p->num_of_local_wpts = 20;
  p->poly_order = 5;
  p->desired_speed = 5.0;
  p->timestep = 0.1;
  p->horizon = 40;
  p->tolerance = 1e-4;
  p->max_iterations = 20;
  p->num_states = 4;
  p->num_ctrls = 2;
  p->w_acc = 1.0;
  p->w_yawrate = 4.0;
  p->w_pos = 0.65;
  p->w_vel = 3.0;
  p->w_obstacle = 1.0;
  p->w_uncertainty = 1.0;
  p->q1_acc = 1.00;
  p->q2_acc = 1.00;
  p->q1_yawrate = 1.0;
  p->q2_yawrate = 1.0;
  p->q1_front = 2.75;
  p->q2_front = 2.75;
  p->q1_rear = 2.5;
  p->q2_rear = 2.5;
  p->q1_uncertainty = 2.5;
  p->q2_uncertainty = 2.5;
  p->acc_max = 2.0;
  p->acc_min = -5.5;
  p->steer_angle_min = -0.75;
  p->steer_angle_max = 0.75;
  p->wheelbase = 2.94;
  p->speed_max = 30.0;
  p->steer_control_max = 1.0;
  p->steer_control_min = -1.0;
  p->throttle_control_max = 1.0;
  p->throttle_control_min = -1.0;
  p->t_safe = 0.1;
  p->s_safe_a = 0;
  p->s_safe_b = 0;
  p->ego_rad = 1.35;
  p->ego_front = 1.47 + 0.925;
  p->ego_rear = 1.47 + 0.925;
  p->length = 4.79;
  p->width = 2.16;
  p->safe_length = 0.0;
  p->safe_width = 0.0;
  p->lamb_factor = 10;
  p->lamb_max = 10000;
}

/* I/iLQR.cpp:9-15 */
void oracle_default_control_seq(int N, double* U) {
  int num_zeros = N / 2;
  for (int i = 0; i < N; i++) {
    U[2 * i + 0] = 1.0 * 0.5;
    U[2 * i + 1] = (i < num_zeros) ? 0.0 : 1.0 * 0.1;
  }
}

/* ------------------------------------------------------------------------------------------------
 * Model — I/Model.cpp:17-30
 * ---------------------------------------------------------------------------------------------- */
void oracle_forward_simulate(const cilqr_params* p, const double* state, const double* control, double* next) {
  double c0 = control[0], c1 = control[1];
  c0 = fmax(fmin(c0, p->acc_max), p->acc_min);
  c1 = fmax(fmin(c1, state[2] * tan(p->steer_angle_max) / p->wheelbase),
            state[2] * tan(p->steer_angle_min) / p->wheelbase);
  const double dt = p->timestep;
  double n0 = state[0] + cos(state[3]) * (state[2] * dt + c0 * dt * dt / 2.0);
  double n1 = state[1] + sin(state[3]) * (state[2] * dt + c0 * dt * dt / 2.0);
  double n2 = fmin(fmax(state[2] + c0 * dt, 0.0), p->speed_max);
  double n3 = state[3] + c1 * dt;
  next[0] = n0; next[1] = n1; next[2] = n2; next[3] = n3;
}

/* I/iLQR.cpp:51-62 */
void oracle_nominal_trajectory(const cilqr_params* p, int N, const double* x0, const double* U, double* X) {
  memcpy(X, x0, 4 * sizeof(double));
  for (int i = 0; i < N; i++) oracle_forward_simulate(p, X + 4 * i, U + 2 * i, X + 4 * (i + 1));
}

/* I/Model.cpp:100-127 (A, stored transposed) and :139-155 (B, stored transposed).
 * Tensor (r,c,t) → r + R*(c + C*t). */
void oracle_AB(const cilqr_params* p, int N, const double* vel, const double* theta, const double* acc,
               double* A, double* B) {
  const double dt = p->timestep;
  for (int i = 0; i < N; i++) {
    double* a = A + 16 * i;
    double* b = B + 8 * i;
#define A_(r, c) a[(r) + 4 * (c)]
#define B_(r, c) b[(r) + 2 * (c)]
    A_(0, 0) = 1.0; A_(1, 0) = 0.0;
    A_(2, 0) = dt * cos(theta[i]);
    A_(3, 0) = (-1) * sin(theta[i]) * (vel[i] * dt + 0.5 * acc[i] * dt * dt);
    A_(0, 1) = 0.0; A_(1, 1) = 1.0;
    A_(2, 1) = dt * sin(theta[i]);
    A_(3, 1) = cos(theta[i]) * (vel[i] * dt + 0.5 * acc[i] * dt * dt);
    A_(0, 2) = 0.0; A_(1, 2) = 0.0; A_(2, 2) = 1.0; A_(3, 2) = 0.0;
    A_(0, 3) = 0.0; A_(1, 3) = 0.0; A_(2, 3) = 0.0; A_(3, 3) = 1.0;
    B_(0, 0) = dt * dt * cos(theta[i]) / 2.0; B_(1, 0) = 0.0;
    B_(0, 1) = dt * dt * sin(theta[i]) / 2.0; B_(1, 1) = 0.0;
    B_(0, 2) = dt; B_(1, 2) = 0.0;
    B_(0, 3) = 0.0; B_(1, 3) = dt;
#undef A_
#undef B_
  }
}

/* ------------------------------------------------------------------------------------------------
 * Constraints — I/Constraints.cpp
 * ---------------------------------------------------------------------------------------------- */
/* :24-59.  Only x_local_plan(0) and x_local_plan(last) are read (:32-33). */
void oracle_find_closest_point(const cilqr_params* p, const double* state, const double* coeffs,
                               double xplan_first, double xplan_last, double* out_xy) {
  const int S = p->num_of_local_wpts * 10;
  double* new_x = (double*)malloc(sizeof(double) * 2 * (size_t)S);
  double* new_y = new_x + S;
  double dx = (xplan_last - xplan_first) / (p->num_of_local_wpts * 10);
  double start_x = xplan_first;
  for (int i = 0; i < S; i++) {
    new_x[i] = start_x + dx * i;
    new_y[i] = 0.0;
    for (int j = 0; j < p->poly_order + 1; j++) new_y[i] += coeffs[j] * pow(new_x[i], (double)j);
  }
  double min_distance = (new_x[0] - state[0]) * (new_x[0] - state[0]) + (new_y[0] - state[1]) * (new_y[0] - state[1]);
  int min_index = 0;
  for (int i = 0; i < S; i++) {
    double t = (new_x[i] - state[0]) * (new_x[i] - state[0]) + (new_y[i] - state[1]) * (new_y[i] - state[1]);
    if (t < min_distance) { min_distance = t; min_index = i; }
  }
  out_xy[0] = new_x[min_index];
  out_xy[1] = new_y[min_index];
  free(new_x);
}

/* Same search over samples built once (the samples depend only on coeffs / x_local_plan, :28-42). */
static void build_samples(const cilqr_params* p, const double* coeffs, double xf, double xl, double* sx, double* sy) {
  const int S = p->num_of_local_wpts * 10;
  double dx = (xl - xf) / (p->num_of_local_wpts * 10);
  for (int i = 0; i < S; i++) {
    sx[i] = xf + dx * i;
    sy[i] = 0.0;
    for (int j = 0; j < p->poly_order + 1; j++) sy[i] += coeffs[j] * pow(sx[i], (double)j);
  }
}
static void closest_in_samples(int S, const double* sx, const double* sy, const double* state, double* out_xy) {
  double md = (sx[0] - state[0]) * (sx[0] - state[0]) + (sy[0] - state[1]) * (sy[0] - state[1]);
  int mi = 0;
  for (int i = 0; i < S; i++) {
    double t = (sx[i] - state[0]) * (sx[i] - state[0]) + (sy[i] - state[1]) * (sy[i] - state[1]);
    if (t < md) { md = t; mi = i; }
  }
  out_xy[0] = sx[mi];
  out_xy[1] = sy[mi];
}

/* 4×4 column-major helpers */
#define M4(m, r, c) (m)[(r) + 4 * (c)]
static void mat4_mul(const double* a, const double* b, double* out) { /* out = a*b */
  double t[16];
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) {
      double s = 0.0;
      for (int k = 0; k < 4; k++) s += M4(a, r, k) * M4(b, k, c);
      M4(t, r, c) = s;
    }
  memcpy(out, t, sizeof(t));
}
static void mat4_vec(const double* a, const double* v, double* out) {
  double t[4];
  for (int r = 0; r < 4; r++) {
    double s = 0.0;
    for (int k = 0; k < 4; k++) s += M4(a, r, k) * v[k];
    t[r] = s;
  }
  memcpy(out, t, sizeof(t));
}

/* Obstacle::barrier_function, I/Obstacle.cpp:21-32 */
static void obstacle_barrier(double q1, double q2, double c, const double* c_dot, double* vx, double* mx) {
  double e = exp(q2 * c);
  double sv = q2 * q1 * e;
  for (int i = 0; i < 4; i++) vx[i] = sv * c_dot[i];
  double sm = q2 * q2 * q1 * e;
  for (int j = 0; j < 4; j++)
    for (int i = 0; i < 4; i++) M4(mx, i, j) = (sm * c_dot[i]) * c_dot[j];
}

/* Obstacle::get_obstalce_cost, I/Obstacle.cpp:39-112 */
void oracle_obstacle_cost(const cilqr_params* p, const double* pose, const double* dim,
                          const double* ego_state, double* vx4, double* mx16) {
  double a = dim[0] / 2.0 + fabs(pose[2] * cos(pose[3])) * p->t_safe + p->s_safe_a + p->ego_rad;
  double b = dim[1] / 2.0 + fabs(pose[2] * sin(pose[3])) * p->t_safe + p->s_safe_b + p->ego_rad + 1;
  double P1[16] = {0};
  M4(P1, 0, 0) = 1.0 / a / a;
  M4(P1, 1, 1) = 1.0 / b / b;
  M4(P1, 2, 2) = 0.0;
  M4(P1, 3, 3) = 0.0;
  double obs_theta = pose[3];
  double vehicle_theta = ego_state[3];
  double tf[16] = {0}, tfr[16] = {0};
  M4(tf, 0, 0) = cos(obs_theta);
  M4(tf, 0, 1) = sin(obs_theta);
  M4(tf, 1, 0) = -sin(obs_theta);
  M4(tf, 1, 1) = cos(obs_theta);
  M4(tfr, 0, 0) = cos(-obs_theta);
  M4(tfr, 0, 1) = sin(-obs_theta);
  M4(tfr, 1, 0) = -sin(-obs_theta);
  M4(tfr, 1, 1) = cos(-obs_theta);
  /* (-2) * tf_martix_reverse * P1, evaluated left to right (:82,101) */
  double m2tfr[16], T1[16];
  for (int i = 0; i < 16; i++) m2tfr[i] = (-2) * tfr[i];
  mat4_mul(m2tfr, P1, T1);

  double vxf[4], mxf[16], vxr[4], mxr[16];
  for (int side = 0; side < 2; side++) {
    double ego[4] = {ego_state[0], ego_state[1], ego_state[2], ego_state[3]};
    if (side == 0) { /* :65-69 */
      ego[0] = ego_state[0] + cos(vehicle_theta) * p->ego_front;
      ego[1] = ego_state[1] + sin(vehicle_theta) * p->ego_front;
    } else { /* :86-90 */
      ego[0] = ego_state[0] - cos(vehicle_theta) * p->ego_rear;
      ego[1] = ego_state[1] - sin(vehicle_theta) * p->ego_rear;
    }
    double d4[4], diff[4];
    for (int i = 0; i < 4; i++) d4[i] = ego[i] - pose[i];
    mat4_vec(tf, d4, diff);
    /* temp = diff' * P1 * diff (:78,99) */
    double row[4];
    for (int j = 0; j < 4; j++) {
      double s = 0.0;
      for (int i = 0; i < 4; i++) s += diff[i] * M4(P1, i, j);
      row[j] = s;
    }
    double temp = 0.0;
    for (int j = 0; j < 4; j++) temp += row[j] * diff[j];
    double c = 1 - temp;
    double c_dot[4];
    mat4_vec(T1, diff, c_dot);
    if (side == 0) obstacle_barrier(p->q1_front, p->q2_front, c, c_dot, vxf, mxf);
    else obstacle_barrier(p->q1_rear, p->q2_rear, c, c_dot, vxr, mxr);
  }
  for (int i = 0; i < 4; i++) vx4[i] = vxf[i] + vxr[i];
  for (int i = 0; i < 16; i++) mx16[i] = mxf[i] + mxr[i];
}

/* Constraints::get_state_cost, I/Constraints.cpp:145-227 with pre-built samples */
static void state_cost_samples(const cilqr_params* p, int N, const double* X, int S, const double* sx,
                               const double* sy, int M, const double* obs_pose, const double* obs_dim,
                               const double* obs_weight, const cilqr_uncertainty_map* um, int ub, double* l_x, double* l_xx) {
  double Q[16] = {0};
  M4(Q, 0, 0) = p->w_pos;
  M4(Q, 1, 1) = p->w_pos;
  M4(Q, 2, 2) = p->w_vel;
  for (int i = 0; i < N; i++) {
    const double* st = X + 4 * i;
    double cp[2];
    closest_in_samples(S, sx, sy, st, cp);
    double temp[4] = {st[0] - cp[0], st[1] - cp[1], st[2] - p->desired_speed, 0};
    double Q2[16];
    for (int k = 0; k < 16; k++) Q2[k] = 2 * Q[k];
    double lxi[4], lxxi[16];
    mat4_vec(Q2, temp, lxi);                             /* :170 */
    for (int k = 0; k < 16; k++) lxxi[k] = Q[k] * 2;     /* :174 */
    for (int j = 0; j < M; j++) {                        /* :177-187 */
      double vx[4], mx[16];
      double w = obs_weight ? obs_weight[j] : p->w_obstacle;
      oracle_obstacle_cost(p, obs_pose + ((size_t)j * N + i) * 4, obs_dim + ((size_t)j * N + i) * 2, st, vx, mx);
      for (int k = 0; k < 4; k++) lxi[k] = lxi[k] + vx[k] * w;
      for (int k = 0; k < 16; k++) lxxi[k] = lxxi[k] + mx[k] * w;
    }
    if (um) { /* :188-201; the cost itself is uncertainty_oracle.c (source absent from the reference) */
      double ux, uvx[4], umx[16];
      oracle_uncertainty_cost(p, um, ub, st, &ux, uvx, umx);
      for (int k = 0; k < 4; k++) lxi[k] += uvx[k] * p->w_uncertainty;
      for (int k = 0; k < 16; k++) lxxi[k] += umx[k] * p->w_uncertainty;
    }
    memcpy(l_x + 4 * i, lxi, sizeof(lxi));
    memcpy(l_xx + 16 * i, lxxi, sizeof(lxxi));
  }
}

void oracle_state_cost(const cilqr_params* p, int N, const double* X, const double* coeffs,
                       double xplan_first, double xplan_last, int M, const double* obs_pose,
                       const double* obs_dim, const double* obs_weight, double* l_x, double* l_xx) {
  const int S = p->num_of_local_wpts * 10;
  double* sx = (double*)malloc(sizeof(double) * 2 * (size_t)S);
  build_samples(p, coeffs, xplan_first, xplan_last, sx, sx + S);
  state_cost_samples(p, N, X, S, sx, sx + S, M, obs_pose, obs_dim, obs_weight, NULL, 0, l_x, l_xx);
  free(sx);
}

/* Constraints::barrier_function, I/Constraints.cpp:67-78 (2-vector form) */
static void ctrl_barrier(double q1, double q2, double c, const double* c_dot, double* vx, double* mx) {
  double e = exp(q2 * c);
  double sv = q2 * q1 * e;
  vx[0] = sv * c_dot[0];
  vx[1] = sv * c_dot[1];
  double sm = q2 * q2 * q1 * e;
  for (int j = 0; j < 2; j++)
    for (int i = 0; i < 2; i++) mx[i + 2 * j] = (sm * c_dot[i]) * c_dot[j];
}

/* Constraints::get_control_cost, I/Constraints.cpp:86-137 */
void oracle_control_cost(const cilqr_params* p, int N, const double* X, const double* U, double* l_u, double* l_uu) {
  const double P1[2] = {1, 0}, P2[2] = {0, 1}, nP1[2] = {-1, -0.0}, nP2[2] = {-0.0, -1};
  double R[4] = {p->w_acc, 0, 0, p->w_yawrate};
  for (int i = 0; i < N; i++) {
    const double* u = U + 2 * i;
    double v1[2], m1[4], v2[2], m2[4], v3[2], m3[4], v4[2], m4[4];
    double t = u[0] * P1[0] + u[1] * P1[1];
    double c = t - p->acc_max;
    ctrl_barrier(p->q1_acc, p->q2_acc, c, P1, v1, m1);
    c = p->acc_min - t;
    ctrl_barrier(p->q1_acc, p->q2_acc, c, nP1, v2, m2);
    t = u[0] * P2[0] + u[1] * P2[1];
    c = t - X[4 * i + 2] * tan(p->steer_angle_max) / p->wheelbase;
    ctrl_barrier(p->q1_yawrate, p->q2_yawrate, c, P2, v3, m3);
    c = X[4 * i + 2] * tan(p->steer_angle_min) / p->wheelbase - t;
    ctrl_barrier(p->q1_yawrate, p->q2_yawrate, c, nP2, v4, m4);
    for (int r = 0; r < 2; r++) {
      double ru = (2 * R[r + 0]) * u[0] + (2 * R[r + 2]) * u[1];
      l_u[2 * i + r] = v1[r] + v2[r] + v3[r] + v4[r] + ru;
    }
    for (int k = 0; k < 4; k++) l_uu[4 * i + k] = m1[k] + m2[k] + m3[k] + m4[k] + 2 * R[k];
  }
}

/* Constraints::get_J, I/Constraints.cpp:534-561 */
static double get_J_samples(const cilqr_params* p, int N, const double* X, const double* U, int S,
                            const double* sx, const double* sy) {
  double J = 0;
  double Q[4] = {p->w_pos, p->w_pos, p->w_vel, 0.0};
  for (int i = 0; i < N; i++) {
    const double* st = X + 4 * i;
    double cp[2];
    closest_in_samples(S, sx, sy, st, cp);
    double d[4] = {st[0] - cp[0], st[1] - cp[1], st[2] - p->desired_speed, st[3]};
    double x_cost = 0.0;
    for (int j = 0; j < 4; j++) x_cost += (d[j] * Q[j]) * d[j];
    const double* u = U + 2 * i;
    double u_cost = (u[0] * p->w_acc) * u[0] + (u[1] * p->w_yawrate) * u[1];
    J += x_cost + u_cost;
  }
  return J;
}

double oracle_get_J(const cilqr_params* p, int N, const double* X, const double* U, const double* coeffs,
                    double xplan_first, double xplan_last) {
  const int S = p->num_of_local_wpts * 10;
  double* sx = (double*)malloc(sizeof(double) * 2 * (size_t)S);
  build_samples(p, coeffs, xplan_first, xplan_last, sx, sx + S);
  double J = get_J_samples(p, N, X, U, S, sx, sx + S);
  free(sx);
  return J;
}

/* ------------------------------------------------------------------------------------------------
 * Q_uu regularised inverse — I/iLQR.cpp:155-175, with the arithmetic of Eigen::EigenSolver<MatrixXd>
 * on a real 2×2 (Eigen 3.2.10 vendored in the reference: M/include/map_engine/Eigen/src/Eigenvalues/
 * RealSchur.h:246-392, EigenSolver.h:370-600, Jacobi/Jacobi.h:214-250,300-330).  Validated against that
 * code by oracle/_ref (tests/test_oracle_ref.py).
 * ---------------------------------------------------------------------------------------------- */
int oracle_quu_inverse(const double* Quu, double lamb, double* Qinv, double* eval2, double* evec4) {
  /* column-major 2×2: T(r,c) = t[r + 2c] */
  double t00 = Quu[0], t10 = Quu[1], t01 = Quu[2], t11 = Quu[3];
  double u00 = 1, u10 = 0, u01 = 0, u11 = 1;
  const double eps = DBL_EPSILON;
  double norm = fabs(t00) + fabs(t10) + fabs(t01) + fabs(t11); /* RealSchur::computeNormOfT */
  if (!(norm == norm)) return -1;                               /* NaN: no real decomposition */
  if (norm != 0) {
    double s = fabs(t00) + fabs(t11);
    if (fabs(t10) <= eps * s) {
      t10 = 0; /* two single roots (findSmallSubdiagEntry) */
    } else {   /* splitOffTwoRows */
      double pp = 0.5 * (t00 - t11);
      double q = pp * pp + t10 * t01;
      if (q >= 0) {
        double z = sqrt(fabs(q));
        double gp = (pp >= 0) ? pp + z : pp - z, gq = t10, c, sn;
        if (gq == 0) { c = gp < 0 ? -1 : 1; sn = 0; }
        else if (gp == 0) { c = 0; sn = gq < 0 ? 1 : -1; }
        else if (fabs(gp) > fabs(gq)) {
          double tt = gq / gp, uu = sqrt(1 + tt * tt);
          if (gp < 0) uu = -uu;
          c = 1 / uu; sn = -tt * c;
        } else {
          double tt = gp / gq, uu = sqrt(1 + tt * tt);
          if (gq < 0) uu = -uu;
          sn = -1 / uu; c = -tt * sn;
        }
        /* T.applyOnTheLeft(0,1,rot.adjoint()): x' = c x - s y, y' = s x + c y on rows */
        if (!(c == 1 && -sn == 0)) {
          double x0 = t00, y0 = t10, x1 = t01, y1 = t11;
          t00 = c * x0 - sn * y0; t10 = sn * x0 + c * y0;
          t01 = c * x1 - sn * y1; t11 = sn * x1 + c * y1;
          /* T.applyOnTheRight(0,1,rot): cols with rot.transpose() = (c,-s) */
          x0 = t00; y0 = t01; x1 = t10; y1 = t11;
          t00 = c * x0 - sn * y0; t01 = sn * x0 + c * y0;
          t10 = c * x1 - sn * y1; t11 = sn * x1 + c * y1;
          /* U.applyOnTheRight(0,1,rot) */
          x0 = u00; y0 = u01; x1 = u10; y1 = u11;
          u00 = c * x0 - sn * y0; u01 = sn * x0 + c * y0;
          u10 = c * x1 - sn * y1; u11 = sn * x1 + c * y1;
        }
        t10 = 0;
      } else {
        return -1; /* complex pair: the reference would take real parts of complex vectors */
      }
    }
  }
  double e0 = t00, e1 = t11;
  /* doComputeEigenvectors */
  double norm2 = fabs(t00) + fabs(t01) + fabs(t10) + fabs(t11);
  if (norm2 != 0.0) {
    /* n = 1 */
    t11 = 1.0;
    {
      double w = t00 - e1;
      double r = t01 * t11;
      if (w != 0.0) t01 = -r / w; else t01 = -r / (eps * norm2);
      double tt = fabs(t01);
      if ((eps * tt) * tt > 1) { t01 /= tt; t11 /= tt; }
    }
    /* n = 0 */
    t00 = 1.0;
    /* back transformation */
    double n01 = u00 * t01 + u01 * t11, n11 = u10 * t01 + u11 * t11;
    u01 = n01; u11 = n11;
    u00 = u00 * t00; u10 = u10 * t00;
  }
  /* eigenvectors(): normalise columns */
  double n0 = sqrt(u00 * u00 + u10 * u10), n1 = sqrt(u01 * u01 + u11 * u11);
  u00 /= n0; u10 /= n0; u01 /= n1; u11 /= n1;
  if (eval2) { eval2[0] = e0; eval2[1] = e1; }
  if (evec4) { evec4[0] = u00; evec4[1] = u10; evec4[2] = u01; evec4[3] = u11; }
  /* :167-169 */
  double d0 = 1.0 / (fmax(e0, 0.0) + lamb);
  double d1 = 1.0 / (fmax(e1, 0.0) + lamb);
  /* Q_uu_inv = V * (D * V') (:175) */
  double dv00 = d0 * u00, dv01 = d0 * u10; /* (D V')(0,0), (0,1) */
  double dv10 = d1 * u01, dv11 = d1 * u11; /* (D V')(1,0), (1,1) */
  Qinv[0] = u00 * dv00 + u01 * dv10;
  Qinv[1] = u10 * dv00 + u11 * dv10;
  Qinv[2] = u00 * dv01 + u01 * dv11;
  Qinv[3] = u10 * dv01 + u11 * dv11;
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * iLQR::backward_pass — I/iLQR.cpp:91-195
 * ---------------------------------------------------------------------------------------------- */
static int backward_pass_samples(const cilqr_params* p, int N, const double* X, const double* U, int S,
                                 const double* sx, const double* sy, int M, const double* obs_pose,
                                 const double* obs_dim, const double* obs_weight, const cilqr_uncertainty_map* um, int ub,
                                 double lamb, double* k, double* K) {
  double* buf = (double*)malloc(sizeof(double) * (size_t)N * (4 + 16 + 2 + 4 + 16 + 8 + 3));
  double* l_x = buf;
  double* l_xx = l_x + 4 * N;
  double* l_u = l_xx + 16 * N;
  double* l_uu = l_u + 2 * N;
  double* A = l_uu + 4 * N;
  double* Bm = A + 16 * N;
  double* vel = Bm + 8 * N;
  double* th = vel + N;
  double* acc = th + N;
  state_cost_samples(p, N, X, S, sx, sy, M, obs_pose, obs_dim, obs_weight, um, ub, l_x, l_xx);
  oracle_control_cost(p, N, X, U, l_u, l_uu);
  for (int i = 0; i < N; i++) { /* :102-106: v, theta of X[:,1..N]; a = U.row(0) */
    vel[i] = X[4 * (i + 1) + 2];
    th[i] = X[4 * (i + 1) + 3];
    acc[i] = U[2 * i];
  }
  oracle_AB(p, N, vel, th, acc, A, Bm);

  double V_x[4], V_xx[16];
  memcpy(V_x, l_x + 4 * (N - 1), sizeof(V_x));       /* :108 */
  memcpy(V_xx, l_xx + 16 * (N - 1), sizeof(V_xx));   /* :110-113 */
  memset(k, 0, sizeof(double) * 2 * N);
  memset(K, 0, sizeof(double) * 8 * N);
  int ok = 1;
  for (int j = N - 1; j >= 0; j--) {
    const double* fx = A + 16 * j;  /* 4×4, = Aᵀ */
    const double* fu = Bm + 8 * j;  /* 2×4, = Bᵀ */
    const double* lxx = l_xx + 16 * j;
    const double* luu = l_uu + 4 * j;
    double Q_x[4], Q_u[2], Q_xx[16], Q_ux[8], Q_uu[4];
    /* :149 */
    for (int r = 0; r < 4; r++) {
      double s = 0.0;
      for (int c = 0; c < 4; c++) s += M4(fx, r, c) * V_x[c];
      Q_x[r] = l_x[4 * j + r] + s;
    }
    /* :150 */
    for (int r = 0; r < 2; r++) {
      double s = 0.0;
      for (int c = 0; c < 4; c++) s += fu[r + 2 * c] * V_x[c];
      Q_u[r] = l_u[2 * j + r] + s;
    }
    /* :151  (fx*V_xx)*fx' */
    double T[16];
    mat4_mul(fx, V_xx, T);
    for (int c = 0; c < 4; c++)
      for (int r = 0; r < 4; r++) {
        double s = 0.0;
        for (int m = 0; m < 4; m++) s += M4(T, r, m) * M4(fx, c, m);
        M4(Q_xx, r, c) = M4(lxx, r, c) + s;
      }
    /* :152-153  E = fu*V_xx (2×4) */
    double E[8];
    for (int c = 0; c < 4; c++)
      for (int r = 0; r < 2; r++) {
        double s = 0.0;
        for (int m = 0; m < 4; m++) s += fu[r + 2 * m] * M4(V_xx, m, c);
        E[r + 2 * c] = s;
      }
    for (int c = 0; c < 4; c++)
      for (int r = 0; r < 2; r++) {
        double s = 0.0;
        for (int m = 0; m < 4; m++) s += E[r + 2 * m] * M4(fx, c, m);
        Q_ux[r + 2 * c] = 0.0 + s; /* l_ux ≡ 0, I/Constraints.cpp:501-506 */
      }
    for (int c = 0; c < 2; c++)
      for (int r = 0; r < 2; r++) {
        double s = 0.0;
        for (int m = 0; m < 4; m++) s += E[r + 2 * m] * fu[c + 2 * m];
        Q_uu[r + 2 * c] = luu[r + 2 * c] + s;
      }
    /* :155-175 */
    double Qinv[4];
    if (oracle_quu_inverse(Q_uu, lamb, Qinv, NULL, NULL) != 0) { ok = 0; break; }
    /* :177-178 */
    double kj[2], Kj[8];
    for (int r = 0; r < 2; r++) kj[r] = ((-1) * Qinv[r]) * Q_u[0] + ((-1) * Qinv[r + 2]) * Q_u[1];
    for (int c = 0; c < 4; c++)
      for (int r = 0; r < 2; r++)
        Kj[r + 2 * c] = ((-1) * Qinv[r]) * Q_ux[0 + 2 * c] + ((-1) * Qinv[r + 2]) * Q_ux[1 + 2 * c];
    /* :180-181  G = K' * Q_uu (4×2) */
    double G[8];
    for (int c = 0; c < 2; c++)
      for (int r = 0; r < 4; r++) G[r + 4 * c] = Kj[0 + 2 * r] * Q_uu[0 + 2 * c] + Kj[1 + 2 * r] * Q_uu[1 + 2 * c];
    for (int r = 0; r < 4; r++) V_x[r] = Q_x[r] - (G[r] * kj[0] + G[r + 4] * kj[1]);
    for (int c = 0; c < 4; c++)
      for (int r = 0; r < 4; r++)
        M4(V_xx, r, c) = M4(Q_xx, r, c) - (G[r] * Kj[0 + 2 * c] + G[r + 4] * Kj[1 + 2 * c]);
    k[2 * j] = kj[0];
    k[2 * j + 1] = kj[1];
    memcpy(K + 8 * j, Kj, sizeof(Kj));
  }
  free(buf);
  return ok;
}

int oracle_backward_pass(const cilqr_params* p, int N, const double* X, const double* U, const double* coeffs,
                         double xplan_first, double xplan_last, int M, const double* obs_pose,
                         const double* obs_dim, const double* obs_weight, double lamb, double* k, double* K) {
  const int S = p->num_of_local_wpts * 10;
  double* sx = (double*)malloc(sizeof(double) * 2 * (size_t)S);
  build_samples(p, coeffs, xplan_first, xplan_last, sx, sx + S);
  int ok = backward_pass_samples(p, N, X, U, S, sx, sx + S, M, obs_pose, obs_dim, obs_weight, NULL, 0, lamb, k, K);
  free(sx);
  return ok;
}

/* iLQR::forward_pass — I/iLQR.cpp:68-86 */
void oracle_forward_pass(const cilqr_params* p, int N, const double* X, const double* U, const double* k,
                         const double* K, double* X_new, double* U_new) {
  memcpy(X_new, X, 4 * sizeof(double));
  for (int i = 0; i < N; i++) {
    double d[4];
    for (int r = 0; r < 4; r++) d[r] = X_new[4 * i + r] - X[4 * i + r];
    for (int r = 0; r < 2; r++) {
      double s = 0.0;
      for (int c = 0; c < 4; c++) s += K[8 * i + r + 2 * c] * d[c];
      U_new[2 * i + r] = U[2 * i + r] + k[2 * i + r] + s;
    }
    oracle_forward_simulate(p, X_new + 4 * i, U_new + 2 * i, X_new + 4 * (i + 1));
  }
}

/* iLQR::get_optimal_control_seq — I/iLQR.cpp:201-245 */
int oracle_solve(const cilqr_params* p, int N, int M, const double* x0, double* U, const double* coeffs,
                 double xplan_first, double xplan_last, const double* obs_pose, const double* obs_dim,
                 const double* obs_weight, double* X_out, double* J_out, int* status_out, double* trace) {
  return oracle_solve_unc(p, N, M, x0, U, coeffs, xplan_first, xplan_last, obs_pose, obs_dim, obs_weight, NULL, 0, X_out, J_out,
                          status_out, trace);
}

/* The same loop with the uncertainty map set (iLQR::set_uncertainty_map, I/iLQR.cpp:28-31): only get_state_cost changes. */
int oracle_solve_unc(const cilqr_params* p, int N, int M, const double* x0, double* U, const double* coeffs,
                     double xplan_first, double xplan_last, const double* obs_pose, const double* obs_dim,
                     const double* obs_weight, const cilqr_uncertainty_map* um, int ub, double* X_out, double* J_out,
                     int* status_out, double* trace) {
  const int S = p->num_of_local_wpts * 10;
  size_t nd = (size_t)2 * S + 4 * (N + 1) * 2 + 2 * N + 2 * N + 8 * N;
  double* buf = (double*)malloc(sizeof(double) * nd);
  double* sx = buf;
  double* sy = sx + S;
  double* X = sy + S;
  double* X_new = X + 4 * (N + 1);
  double* U_new = X_new + 4 * (N + 1);
  double* k = U_new + 2 * N;
  double* K = k + 2 * N;
  build_samples(p, coeffs, xplan_first, xplan_last, sx, sy);

  oracle_nominal_trajectory(p, N, x0, U, X);  /* :203 */
  double J_old = DBL_MAX;                      /* :204 */
  double lamb = 1;
  double J_new = 0;
  int iteration_times = 0;
  int status = CILQR_EXIT_MAX_ITER;
  for (int i = 0; i < p->max_iterations; i++) {
    iteration_times++;
    int ok = backward_pass_samples(p, N, X, U, S, sx, sy, M, obs_pose, obs_dim, obs_weight, um, ub, lamb, k, K);
    if (!ok) { status = CILQR_EXIT_NUMERIC; break; }
    oracle_forward_pass(p, N, X, U, k, K, X_new, U_new);
    J_new = get_J_samples(p, N, X, U, S, sx, sy); /* :217 — on the current (old) X, U */
    int accepted = 0;
    int stop = 0;
    if (J_new < J_old) {
      memcpy(X, X_new, sizeof(double) * 4 * (N + 1));
      memcpy(U, U_new, sizeof(double) * 2 * N);
      accepted = 1;
      lamb = lamb / p->lamb_factor;
      if (fabs(J_new - J_old) < p->tolerance) { status = CILQR_EXIT_TOLERANCE; stop = 1; }
    } else {
      lamb = lamb * p->lamb_factor;
      if (lamb > p->lamb_max) { status = CILQR_EXIT_LAMBDA_MAX; stop = 1; }
    }
    if (trace) { trace[3 * i] = J_new; trace[3 * i + 1] = lamb; trace[3 * i + 2] = accepted; }
    if (stop) break;
    J_old = J_new; /* :238 */
  }
  memcpy(X_out, X, sizeof(double) * 4 * (N + 1)); /* :243-244; U already in place */
  if (J_out) *J_out = get_J_samples(p, N, X, U, S, sx, sy);
  if (status_out) *status_out = status;
  free(buf);
  return iteration_times;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

int oracle_solve_batch(const cilqr_params* p, int B, int N, int M, const double* x0, double* U,
                       const double* poly, const double* xplan_fl, const double* obs_pose,
                       const double* obs_dim, const double* obs_weight, double* X_out, double* J_out,
                       int* iters_out, int* status_out, int threads) {
  return oracle_solve_batch_unc(p, B, N, M, x0, U, poly, xplan_fl, obs_pose, obs_dim, obs_weight, NULL, X_out, J_out, iters_out,
                                status_out, threads);
}

int oracle_solve_batch_unc(const cilqr_params* p, int B, int N, int M, const double* x0, double* U,
                           const double* poly, const double* xplan_fl, const double* obs_pose,
                           const double* obs_dim, const double* obs_weight, const cilqr_uncertainty_map* um, double* X_out,
                           double* J_out, int* iters_out, int* status_out, int threads) {
  if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic) num_threads(threads)
#endif
  for (int b = 0; b < B; b++) {
    int st = 0;
    double J = 0;
    int it = oracle_solve_unc(p, N, M, x0 + 4 * (size_t)b, U + 2 * (size_t)N * b, poly + 6 * (size_t)b,
                              xplan_fl[2 * b], xplan_fl[2 * b + 1],
                              M ? obs_pose + (size_t)b * M * N * 4 : NULL, M ? obs_dim + (size_t)b * M * N * 2 : NULL,
                              obs_weight ? obs_weight + (size_t)b * M : NULL, um, b, X_out + 4 * (size_t)(N + 1) * b, &J, &st, NULL);
    if (J_out) J_out[b] = J;
    if (iters_out) iters_out[b] = it;
    if (status_out) status_out[b] = st;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * LocalPlanner — I/LocalPlanner.cpp
 * ---------------------------------------------------------------------------------------------- */
/* polyfit (:101-117) = Vandermonde + Eigen::ColPivHouseholderQR::solve (Eigen 3.2.10:
 * QR/ColPivHouseholderQR.h:425-545, Householder/Householder.h makeHouseholder/applyHouseholderOnTheLeft). */
void oracle_polyfit(const double* x, const double* y, int n, int degree, double* coeffs) {
  const int rows = n, cols = degree + 1;
  const int size = rows < cols ? rows : cols;
  double* qr = (double*)malloc(sizeof(double) * ((size_t)rows * cols + 3 * (size_t)cols + rows));
  double* hco = qr + (size_t)rows * cols;
  double* nrm = hco + cols;
  double* tmp = nrm + cols;
  double* c = tmp + cols;
  int* perm = (int*)malloc(sizeof(int) * 2 * (size_t)cols);
  int* trans = perm + cols;
#define QR(r, cc) qr[(r) + (size_t)rows * (cc)]
  for (int i = 0; i < rows; i++)
    for (int j = 0; j < cols; j++) QR(i, j) = pow(x[i], (double)j);
  double maxn = 0;
  for (int k = 0; k < cols; k++) {
    double s = 0;
    for (int i = 0; i < rows; i++) s += QR(i, k) * QR(i, k);
    nrm[k] = s;
    if (k == 0 || s > maxn) maxn = s;
  }
  double threshold_helper = maxn * (DBL_EPSILON * DBL_EPSILON) / (double)rows;
  int nonzero_pivots = size;
  for (int k = 0; k < size; k++) {
    int big = k;
    double bn = nrm[k];
    for (int j = k + 1; j < cols; j++)
      if (nrm[j] > bn) { bn = nrm[j]; big = j; }
    bn = 0;
    for (int i = k; i < rows; i++) bn += QR(i, big) * QR(i, big);
    nrm[big] = bn;
    if (nonzero_pivots == size && bn < threshold_helper * (double)(rows - k)) nonzero_pivots = k;
    trans[k] = big;
    if (k != big) {
      for (int i = 0; i < rows; i++) { double t = QR(i, k); QR(i, k) = QR(i, big); QR(i, big) = t; }
      double t = nrm[k]; nrm[k] = nrm[big]; nrm[big] = t;
    }
    /* makeHouseholderInPlace on col k, rows k.. */
    double tail2 = 0;
    for (int i = k + 1; i < rows; i++) tail2 += QR(i, k) * QR(i, k);
    double c0 = QR(k, k), beta, tau;
    if (tail2 == 0) {
      tau = 0; beta = c0;
      for (int i = k + 1; i < rows; i++) QR(i, k) = 0;
    } else {
      beta = sqrt(c0 * c0 + tail2);
      if (c0 >= 0) beta = -beta;
      for (int i = k + 1; i < rows; i++) QR(i, k) = QR(i, k) / (c0 - beta);
      tau = (beta - c0) / beta;
    }
    hco[k] = tau;
    QR(k, k) = beta;
    /* applyHouseholderOnTheLeft to bottomRightCorner(rows-k, cols-k-1) */
    if (rows - k == 1) {
      for (int j = k + 1; j < cols; j++) QR(k, j) *= (1 - tau);
    } else {
      for (int j = k + 1; j < cols; j++) {
        double t = 0;
        for (int i = k + 1; i < rows; i++) t += QR(i, k) * QR(i, j);
        t += QR(k, j);
        QR(k, j) -= tau * t;
        for (int i = k + 1; i < rows; i++) QR(i, j) -= tau * QR(i, k) * t;
      }
    }
    for (int j = k + 1; j < cols; j++) nrm[j] -= QR(k, j) * QR(k, j);
  }
  for (int j = 0; j < cols; j++) perm[j] = j;
  for (int k = 0; k < size; k++) { int t = perm[k]; perm[k] = perm[trans[k]]; perm[trans[k]] = t; }
  /* solve */
  for (int j = 0; j < cols; j++) coeffs[j] = 0;
  if (nonzero_pivots > 0) {
    for (int i = 0; i < rows; i++) c[i] = y[i];
    for (int k = 0; k < nonzero_pivots; k++) { /* c = H_k c */
      if (rows - k == 1) { c[k] *= (1 - hco[k]); continue; }
      double t = 0;
      for (int i = k + 1; i < rows; i++) t += QR(i, k) * c[i];
      t += c[k];
      c[k] -= hco[k] * t;
      for (int i = k + 1; i < rows; i++) c[i] -= hco[k] * QR(i, k) * t;
    }
    for (int i = nonzero_pivots - 1; i >= 0; i--) { /* upper-triangular back substitution */
      double s = c[i];
      for (int j = i + 1; j < nonzero_pivots; j++) s -= QR(i, j) * c[j];
      c[i] = s / QR(i, i);
    }
    for (int i = 0; i < nonzero_pivots; i++) coeffs[perm[i]] = c[i];
  }
#undef QR
  free(perm);
  free(qr);
}

/* closest_point_index (:25-41), get_local_wpts (:47-60), get_local_plan (:66-85), get_local_plan_coeffs (:90-96) */
int oracle_local_plan(const cilqr_params* p, const double* path, int P, const double* ego_state,
                      double* coeffs, double* ref_traj) {
  double md = pow(ego_state[0] - path[0], 2) + pow(ego_state[1] - path[1], 2);
  int mi = 0;
  for (int i = 0; i < P; i++) {
    double t = pow(ego_state[0] - path[2 * i], 2) + pow(ego_state[1] - path[2 * i + 1], 2);
    if (t < md) { md = t; mi = i; }
  }
  int n = (P - mi) < p->num_of_local_wpts ? (P - mi) : p->num_of_local_wpts;
  if (n < 1) return 0;  /* P >= 1 and num_of_local_wpts >= 1 are the caller's contract */
  double* xs = (double*)calloc(2 * (size_t)n, sizeof(double));
  double* ys = xs + n;
  for (int i = 0; i < n; i++) { xs[i] = path[2 * (mi + i)]; ys[i] = path[2 * (mi + i) + 1]; }
  oracle_polyfit(xs, ys, n, p->poly_order, coeffs);
  for (int i = 0; i < n; i++) {
    double ny = 0.0;
    for (int j = 0; j < p->poly_order + 1; j++) ny += coeffs[j] * pow(xs[i], (double)j);
    ref_traj[2 * i] = xs[i];
    ref_traj[2 * i + 1] = ny;
  }
  free(xs);
  return n;
}
