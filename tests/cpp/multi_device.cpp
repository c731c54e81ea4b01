// multi_device.cpp — a C++ host (no Python, no torch) running one batch over every GPU of the node through the C-ABI:
// cilqr_create_multi + cilqr_multi_solve_batch (scene-sharded, the RCCL exchange step inside) against the same batch on one
// handle with the pick done on the host.  Prints JSON; exit code 1 on any difference.  Usage: multi_device [B] [n_devices].
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cilqr.h"

#define CHECK(call)                                                            \
  do {                                                                         \
    int rc_ = (call);                                                          \
    if (rc_ != CILQR_OK) { fprintf(stderr, "%s: %s\n", #call, cilqr_last_error()); return 2; } \
  } while (0)

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 37, N = 50, M = 4;
  int n_dev = cilqr_device_count();
  if (argc > 2) n_dev = atoi(argv[2]);
  if (n_dev < 1) { fprintf(stderr, "no gfx950 device\n"); return 2; }
  cilqr_params p;
  cilqr_params_default(&p);
  p.horizon = N;
  std::vector<double> path(2 * 200);
  for (int i = 0; i < 200; ++i) { path[2 * i] = i; path[2 * i + 1] = 0.5 * std::sin(0.05 * i); }
  std::vector<double> x0(4 * B), U(2 * N * (size_t)B), poly(6 * (size_t)B), fl(2 * (size_t)B), pose((size_t)B * M * N * 4), dim((size_t)B * M * N * 2);
  for (int b = 0; b < B; ++b) {
    double* e = &x0[4 * b];
    e[0] = 0.3 * b; e[1] = 0.1 + 0.02 * std::sin(1.3 * b); e[2] = 3.0 + 0.05 * (b % 7); e[3] = 0.02 - 0.001 * (b % 5);
    double ref[40];
    int n = 0;
    CHECK(cilqr_local_plan(&p, path.data(), 200, e, &poly[6 * b], ref, &n));
    fl[2 * b] = ref[0]; fl[2 * b + 1] = ref[2 * (n - 1)];
    CHECK(cilqr_default_control_seq(N, &U[2 * N * (size_t)b]));
    for (int o = 0; o < M; ++o)
      for (int t = 0; t < N; ++t) {
        double* q = &pose[(((size_t)b * M + o) * N + t) * 4];
        q[0] = 15 + 12 * o + 0.1 * (b % 3); q[1] = (o % 2) ? -1.0 : 0.8; q[2] = 0; q[3] = 0.1 * o;
        double* d = &dim[(((size_t)b * M + o) * N + t) * 2];
        d[0] = 4.79; d[1] = 2.16;
      }
  }
  // one handle, pick on the host
  std::vector<double> U1 = U, X1(4 * (N + 1) * (size_t)B), J1(B);
  std::vector<int32_t> it1(B), st1(B);
  cilqr_handle* h = nullptr;
  CHECK(cilqr_create(&p, B, N, M, 0, &h));
  CHECK(cilqr_solve_batch(h, B, N, M, x0.data(), U1.data(), poly.data(), fl.data(), pose.data(), dim.data(), nullptr, X1.data(), J1.data(),
                          it1.data(), st1.data(), CILQR_FLAG_NONE));
  CHECK(cilqr_destroy(h));
  long best1 = -1;
  for (int b = 0; b < B; ++b)
    if (J1[b] == J1[b] && (best1 < 0 || J1[b] < J1[best1])) best1 = b;
  // every device, pick by the RCCL step
  std::vector<double> U2 = U, X2(4 * (N + 1) * (size_t)B), J2(B);
  std::vector<int32_t> it2(B), st2(B);
  cilqr_multi* m = nullptr;
  const int per = (B + n_dev - 1) / n_dev;
  CHECK(cilqr_create_multi(&p, per, N, M, n_dev, nullptr, &m));
  int64_t best2 = -2;
  double bestJ = 0.0;
  for (int rep = 0; rep < 2; ++rep) {  // twice: the communicators and staging are reused
    U2 = U;
    CHECK(cilqr_multi_solve_batch(m, B, N, M, x0.data(), U2.data(), poly.data(), fl.data(), pose.data(), dim.data(), nullptr, X2.data(),
                                  J2.data(), it2.data(), st2.data(), CILQR_FLAG_NONE, &best2, &bestJ));
  }
  const int devs = cilqr_multi_device_count(m);
  CHECK(cilqr_multi_destroy(m));
  bool same = best1 == best2 && bestJ == J1[best1];
  for (size_t i = 0; i < U1.size(); ++i) same = same && U1[i] == U2[i];
  for (size_t i = 0; i < X1.size(); ++i) same = same && X1[i] == X2[i];
  for (int b = 0; b < B; ++b) same = same && J1[b] == J2[b] && it1[b] == it2[b] && st1[b] == st2[b];
  printf("{\"devices\": %d, \"B\": %d, \"best_single\": %ld, \"best_multi\": %ld, \"best_J\": %.17g, \"bit_equal\": %s}\n", devs, B, best1,
         (long)best2, bestJ, same ? "true" : "false");
  return same ? 0 : 1;
}
