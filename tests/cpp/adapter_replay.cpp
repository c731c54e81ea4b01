// adapter_replay.cpp — drives cilqr_host::iLQR exactly as the reference node drives its iLQR object
// (set_global_plan → set_Obstacle → clear_uncertainty_map → run_step, I/ilqr_uncertainty_node.cpp:113-119) on the
// known-answer scene of SURVEY.md §8(c), for two consecutive ticks, and prints the results as JSON.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ilqr_adapter.h"

using namespace cilqr_host;

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 50, M = argc > 2 ? atoi(argv[2]) : 4;
  Parameters params = default_parameters();
  params.horizon = N;
  iLQR planner(params, 0, 64, 8);
  Matrix path(2, 200);
  for (int i = 0; i < 200; ++i) { path(0, i) = i; path(1, i) = 0.5 * std::sin(0.05 * i); }
  std::vector<Obstacle> obstacles;
  for (int o = 0; o < M; ++o) {
    Matrix dim(2, N), pose(4, N);
    for (int t = 0; t < N; ++t) {
      dim(0, t) = 4.79; dim(1, t) = 2.16;
      pose(0, t) = 15 + 12 * o; pose(1, t) = (o % 2) ? -1.0 : 0.8; pose(2, t) = 0; pose(3, t) = 0.1 * o;
    }
    obstacles.emplace_back(params, dim, pose);
  }
  planner.set_global_plan(path);
  planner.set_Obstacle(obstacles);
  planner.clear_uncertainty_map();
  const double ego[4] = {0, 0.1, 3.0, 0.02};
  printf("{\"ticks\": [");
  for (int tick = 0; tick < 2; ++tick) {
    planner.run_step(ego);
    printf("%s{\"iterations\": %d, \"exit\": %d, \"J\": %.17g, \"U\": [", tick ? ", " : "", planner.last_iterations, planner.last_exit,
           planner.last_cost);
    for (size_t i = 0; i < planner.U_result.a.size(); ++i) printf("%s%.17g", i ? ", " : "", planner.U_result.a[i]);
    printf("], \"X\": [");
    for (size_t i = 0; i < planner.X_result.a.size(); ++i) printf("%s%.17g", i ? ", " : "", planner.X_result.a[i]);
    printf("], \"n_ref\": %d}", planner.ref_traj_result.cols);
  }
  // candidates: the nominal ego state plus perturbed copies, one launch, minimum-cost pick
  std::vector<double> cands;
  for (int c = 0; c < 8; ++c) { cands.push_back(0.0 + 0.05 * c); cands.push_back(0.1 - 0.04 * c); cands.push_back(3.0); cands.push_back(0.02 + 0.01 * c); }
  const int best = planner.run_candidates(cands);
  printf("], \"best\": %d, \"best_J\": %.17g}\n", best, planner.last_cost);
  return 0;
}
