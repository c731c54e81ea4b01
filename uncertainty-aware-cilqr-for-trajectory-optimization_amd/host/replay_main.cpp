// replay_main.cpp — ROS-free replay of recorded planner ticks through the adapter: what the node's
// odomCallback does per odometry message (I/ilqr_uncertainty_node.cpp:111-130: set_global_plan → set_Obstacle → run_step →
// publishExperimentData), fed from a text log instead of ROS topics.
//
//   cilqr_replay <log> [device]     → one line per tick on stdout
//
// Log (whitespace-separated, '#' starts a comment line):
//   cilqr-replay 1
//   horizon <N>
//   path <P>            followed by P lines "x y"                       (nav_msgs/Path of the global plan)
//   tick                one per odometry message, each followed by
//   ego <x> <y> <v> <theta>
//   obstacles <M>       followed by M lines "x y v theta length width" (held over the horizon, as the node's static-obstacle
//                       callback replicates them, I/ilqr_uncertainty_node.cpp:175-185)
// Output per tick: "experiment <start_pos 4> <planning_time> <iterations> <exit> <J> X <4(N+1) values> U <2N values>" —
// X and U in the vehiclepub/Experiment flattening (:243-284).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "ilqr_adapter.h"

using namespace cilqr_host;

namespace {

struct Tokens {
  std::vector<std::string> t;
  size_t at = 0;
  explicit Tokens(std::istream& in) {
    std::string line, w;
    while (std::getline(in, line)) {
      const size_t h = line.find_first_not_of(" \t");
      if (h == std::string::npos || line[h] == '#') continue;
      std::istringstream ls(line);
      while (ls >> w) t.push_back(w);
    }
  }
  bool done() const { return at >= t.size(); }
  std::string word() {
    if (done()) throw std::runtime_error("replay log ends in the middle of a record");
    return t[at++];
  }
  void expect(const char* w) {
    const std::string g = word();
    if (g != w) throw std::runtime_error("replay log: expected '" + std::string(w) + "', found '" + g + "'");
  }
  double num() { return std::stod(word()); }
  int integer() { return std::stoi(word()); }
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s <log> [device]\n", argv[0]);
    return 2;
  }
  try {
    std::ifstream f(argv[1]);
    if (!f) throw std::runtime_error(std::string("cannot open ") + argv[1]);
    Tokens in(f);
    in.expect("cilqr-replay");
    if (in.integer() != 1) throw std::runtime_error("replay log: unknown version");
    in.expect("horizon");
    const int N = in.integer();
    in.expect("path");
    const int P = in.integer();
    if (N < 1 || P < 1) throw std::runtime_error("replay log: horizon and path length must be positive");
    Matrix path(2, P);
    for (int i = 0; i < P; ++i) { path(0, i) = in.num(); path(1, i) = in.num(); }

    Parameters params = default_parameters();
    params.horizon = N;
    iLQR planner(params, argc > 2 ? atoi(argv[2]) : 0, 64, 1);
    while (!in.done()) {
      in.expect("tick");
      in.expect("ego");
      double ego[4];
      for (double& v : ego) v = in.num();
      in.expect("obstacles");
      const int M = in.integer();
      std::vector<Obstacle> obstacles;
      for (int o = 0; o < M; ++o) {
        double rec[6];
        for (double& v : rec) v = in.num();
        Matrix dim(2, N), pose(4, N);
        for (int t = 0; t < N; ++t) {
          for (int r = 0; r < 4; ++r) pose(r, t) = rec[r];
          dim(0, t) = rec[4];
          dim(1, t) = rec[5];
        }
        obstacles.emplace_back(params, dim, pose);
      }
      planner.set_global_plan(path);
      if (M) planner.set_Obstacle(obstacles); else planner.clear_Obstacle();
      const auto t0 = std::chrono::high_resolution_clock::now();
      planner.run_step(ego);
      const double secs = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
      const Experiment e = flatten_experiment(ego, secs, planner.X_result, planner.U_result);
      printf("experiment %.17g %.17g %.17g %.17g %.9g %d %d %.17g X", e.start_pos[0], e.start_pos[1], e.start_pos[2], e.start_pos[3],
             e.planning_time, planner.last_iterations, planner.last_exit, planner.last_cost);
      for (double v : e.X) printf(" %.17g", v);
      printf(" U");
      for (double v : e.U) printf(" %.17g", v);
      printf("\n");
    }
  } catch (const std::exception& ex) {
    fprintf(stderr, "cilqr_replay: %s\n", ex.what());
    return 1;
  }
  return 0;
}
