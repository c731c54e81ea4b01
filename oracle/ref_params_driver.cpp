// ref_params_driver.cpp — TEST INFRASTRUCTURE ONLY.
//
// Copies the defaults set by the REFERENCE's Parameters::Parameters() (I/Parameters.cpp:3-75, compiled from
// where it lies, oracle/Makefile target _ref/libref_params.so) into the POD mirror of include/cilqr.h, so
// tests can check oracle_params_default and cilqr_params_default field by field.
#include <cstring>

#include "Parameters.h"
#include "cilqr.h"

extern "C" void ref_params_default(cilqr_params* o) {
  Parameters p;
  std::memset(o, 0, sizeof(*o));
  o->num_of_local_wpts = p.num_of_local_wpts;
  o->poly_order = p.poly_order;
  o->horizon = p.horizon;
  o->max_iterations = p.max_iterations;
  o->num_states = p.num_states;
  o->num_ctrls = p.num_ctrls;
  o->desired_speed = p.desired_speed;
  o->timestep = p.timestep;
  o->tolerance = p.tolerance;
  o->w_acc = p.w_acc;
  o->w_yawrate = p.w_yawrate;
  o->w_pos = p.w_pos;
  o->w_vel = p.w_vel;
  o->w_obstacle = p.w_obstacle;
  o->w_uncertainty = p.w_uncertainty;
  o->q1_acc = p.q1_acc;
  o->q2_acc = p.q2_acc;
  o->q1_yawrate = p.q1_yawrate;
  o->q2_yawrate = p.q2_yawrate;
  o->q1_front = p.q1_front;
  o->q2_front = p.q2_front;
  o->q1_rear = p.q1_rear;
  o->q2_rear = p.q2_rear;
  o->q1_uncertainty = p.q1_uncertainty;
  o->q2_uncertainty = p.q2_uncertainty;
  o->acc_max = p.acc_max;
  o->acc_min = p.acc_min;
  o->steer_angle_min = p.steer_angle_min;
  o->steer_angle_max = p.steer_angle_max;
  o->wheelbase = p.wheelbase;
  o->speed_max = p.speed_max;
  o->steer_control_max = p.steer_control_max;
  o->steer_control_min = p.steer_control_min;
  o->throttle_control_max = p.throttle_control_max;
  o->throttle_control_min = p.throttle_control_min;
  o->t_safe = p.t_safe;
  o->s_safe_a = p.s_safe_a;
  o->s_safe_b = p.s_safe_b;
  o->ego_rad = p.ego_rad;
  o->ego_front = p.ego_front;
  o->ego_rear = p.ego_rear;
  o->length = p.length;
  o->width = p.width;
  o->safe_length = p.safe_length;
  o->safe_width = p.safe_width;
  // lamb_factor / lamb_max live in iLQR::iLQR (I/iLQR.cpp:17-18), whose translation unit needs headers this
  // image lacks; they are left 0 here and checked against the literal values in the test instead.
}
