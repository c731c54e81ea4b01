// adapter_uncertainty.cpp — the reference node's per-tick sequence with the uncertainty map
// (Uncertainty vehicle_map(...); set_uncertainty_map; set_global_plan; run_step — I/ilqr_uncertainty_node.cpp:111-119) through
// cilqr_host::iLQR on the known-answer scene of SURVEY §8(c), then the same tick after clear_uncertainty_map on a fresh planner.
// Usage: adapter_uncertainty layer.bin rows cols len_x len_y res pos_x pos_y pose_x pose_y pose_theta
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ilqr_adapter.h"

using namespace cilqr_host;

static void print_tick(const char* name, const iLQR& pl, bool last) {
  printf("\"%s\": {\"iterations\": %d, \"exit\": %d, \"U\": [", name, pl.last_iterations, pl.last_exit);
  for (size_t i = 0; i < pl.U_result.a.size(); ++i) printf("%s%.17g", i ? ", " : "", pl.U_result.a[i]);
  printf("]}%s", last ? "" : ", ");
}

int main(int argc, char** argv) {
  if (argc < 12) { fprintf(stderr, "usage: see the header comment\n"); return 2; }
  const int N = 50, M = 4;
  Uncertainty um;
  const int rows = atoi(argv[2]), cols = atoi(argv[3]);
  um.layer.resize((size_t)rows * cols);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(um.layer.data(), sizeof(float), um.layer.size(), f) != um.layer.size()) { fprintf(stderr, "cannot read the layer\n"); return 2; }
  fclose(f);
  if (cilqr_map_geom_set(&um.geom, atof(argv[4]), atof(argv[5]), atof(argv[6]), atof(argv[7]), atof(argv[8])) != CILQR_OK) return 2;
  if (um.geom.rows != rows || um.geom.cols != cols) { fprintf(stderr, "geometry does not match the layer\n"); return 2; }
  um.pose_x = atof(argv[9]); um.pose_y = atof(argv[10]); um.pose_theta = atof(argv[11]);
  Parameters params = default_parameters();
  params.horizon = N;
  params.safe_length = 1.1;  // the launch file's values, which the node hands to the Uncertainty constructor (Experiment.launch:7-8)
  params.safe_width = 0.9;
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
  const double ego[4] = {0, 0.1, 3.0, 0.02};
  printf("{");
  {
    iLQR planner(params, 0, 8, 1);
    planner.set_Obstacle(obstacles);
    planner.set_uncertainty_map(um);
    planner.set_global_plan(path);
    planner.run_step(ego);
    print_tick("with_map", planner, false);
  }
  {
    iLQR planner(params, 0, 8, 1);
    planner.set_Obstacle(obstacles);
    planner.set_uncertainty_map(um);
    planner.clear_uncertainty_map();
    planner.set_global_plan(path);
    planner.run_step(ego);
    print_tick("cleared", planner, true);
  }
  printf("}\n");
  return 0;
}
