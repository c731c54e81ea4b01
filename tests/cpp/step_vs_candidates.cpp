// step_vs_candidates.cpp — iLQR::run_step (host pre-step) against iLQR::run_candidates with one candidate (device pre-step)
// from identical planner states, over a sweep of ego poses along a curved path.  Prints the largest |ΔU| and |ΔX| seen.
#include <cmath>
#include <cstdio>
#include <vector>

#include "ilqr_adapter.h"

using namespace cilqr_host;

int main() {
  const int N = 50, M = 3;
  Parameters params = default_parameters();
  params.horizon = N;
  Matrix path(2, 200);
  for (int i = 0; i < 200; ++i) { path(0, i) = 37.25 + 1.03 * i; path(1, i) = 1.2 * std::sin(0.05 * path(0, i)); }
  std::vector<Obstacle> obstacles;
  for (int o = 0; o < M; ++o) {
    Matrix dim(2, N), pose(4, N);
    for (int t = 0; t < N; ++t) {
      dim(0, t) = 4.79; dim(1, t) = 2.16;
      pose(0, t) = 60 + 14 * o; pose(1, t) = (o % 2) ? -1.5 : 2.0; pose(2, t) = 0; pose(3, t) = 0.1 * o;
    }
    obstacles.emplace_back(params, dim, pose);
  }
  double worst = 0.0;
  int checked = 0;
  for (int k = 0; k < 24; ++k) {
    const double x = 40.0 + 5.7 * k;
    const double ego[4] = {x, 1.2 * std::sin(0.05 * x) + 0.3 * std::cos(1.7 * k), 2.0 + 0.2 * k, 0.06 * std::cos(0.05 * x) + 0.02 * std::sin(2.3 * k)};
    iLQR a(params, 0, 8, 1), b(params, 0, 8, 1);  // fresh planners: the same default warm start
    a.set_global_plan(path); b.set_global_plan(path);
    a.set_Obstacle(obstacles); b.set_Obstacle(obstacles);
    a.run_step(ego);
    const int best = b.run_candidates(std::vector<double>(ego, ego + 4));
    if (best != 0 || a.last_iterations != b.last_iterations || a.last_exit != b.last_exit) {
      printf("tick %d: iterations %d vs %d, exit %d vs %d\n", k, a.last_iterations, b.last_iterations, a.last_exit, b.last_exit);
      return 1;
    }
    for (size_t i = 0; i < a.U_result.a.size(); ++i) worst = std::fmax(worst, std::fabs(a.U_result.a[i] - b.U_result.a[i]));
    for (size_t i = 0; i < a.X_result.a.size(); ++i) worst = std::fmax(worst, std::fabs(a.X_result.a[i] - b.X_result.a[i]));
    if (a.ref_traj_result.cols != b.ref_traj_result.cols) { printf("tick %d: slice length differs\n", k); return 1; }
    ++checked;
  }
  printf("checked %d worst %.3e\n", checked, worst);
  return 0;
}
