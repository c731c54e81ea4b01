// multi_device.cpp — a C++ host (no Python, no torch) running one batch over several GPUs through the C-ABI, both host models of
// include/cilqr.h, each compared bit for bit with the same batch on ONE handle and a pick done on the host:
//   * one process, every GPU: cilqr_create_multi + cilqr_multi_solve_batch for n = 1 .. cilqr_device_count() devices
//     (scene-sharded by cilqr_shard_range, the RCCL exchange step inside); `--shards n` instead puts n shards on device 0
//     (how a one-GPU box rehearses the n-shard arithmetic; no RCCL there, see cilqr_multi_uses_rccl);
//   * one process per GPU: launched with RANK / WORLD_SIZE / LOCAL_RANK in the environment (mpirun, torchrun --no-python, a shell
//     loop), rank 0 makes the RCCL id (cilqr_comm_unique_id) and hands it over through a file, every rank joins
//     (cilqr_comm_init_rank), solves its shard and calls cilqr_argmin_global_device.
// Options: --B n (default 37: a packed call; ≥ 1024 takes the array-by-array copies), --pinned (buffers from cilqr_host_alloc),
// --fail d (force an enqueue failure on shard d with copies in flight: the call must fail cleanly and the next one succeed).
// Prints one JSON line; exit code 1 on any difference.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "cilqr.h"

#define CHECK(call)                                                            \
  do {                                                                         \
    int rc_ = (call);                                                          \
    if (rc_ != CILQR_OK) { fprintf(stderr, "%s: %s\n", #call, cilqr_last_error()); return 2; } \
  } while (0)

namespace {

template <typename T>
struct Buf {  // pageable (new[]) or pinned (cilqr_host_alloc) host array
  T* p = nullptr;
  size_t n = 0;
  bool pinned = false;
  Buf(size_t n_, bool pin) : n(n_), pinned(pin) {
    p = pin ? (T*)cilqr_host_alloc(sizeof(T) * (n ? n : 1)) : new T[n ? n : 1];
    memset(p, 0, sizeof(T) * (n ? n : 1));
  }
  ~Buf() { if (pinned) cilqr_host_free(p); else delete[] p; }
  Buf(const Buf&) = delete;
  T* data() { return p; }
  T& operator[](size_t i) { return p[i]; }
};

struct Scene {
  int B, N, M;
  bool pin;
  Buf<double> x0, U, poly, fl, pose, dim;
  Scene(int B_, int N_, int M_, bool pin_)
      : B(B_), N(N_), M(M_), pin(pin_), x0(4 * (size_t)B_, pin_), U(2 * (size_t)N_ * B_, pin_), poly(6 * (size_t)B_, pin_), fl(2 * (size_t)B_, pin_),
        pose((size_t)B_ * M_ * N_ * 4, pin_), dim((size_t)B_ * M_ * N_ * 2, pin_) {}
};

int fill(Scene& s, const cilqr_params& p) {
  std::vector<double> path(2 * 200);
  for (int i = 0; i < 200; ++i) { path[2 * i] = i; path[2 * i + 1] = 0.5 * std::sin(0.05 * i); }
  for (int b = 0; b < s.B; ++b) {
    double* e = &s.x0[4 * (size_t)b];
    e[0] = 0.3 * (b % 400); e[1] = 0.1 + 0.02 * std::sin(1.3 * b); e[2] = 3.0 + 0.05 * (b % 7); e[3] = 0.02 - 0.001 * (b % 5);
    double ref[40];
    int n = 0;
    CHECK(cilqr_local_plan(&p, path.data(), 200, e, &s.poly[6 * (size_t)b], ref, &n));
    s.fl[2 * (size_t)b] = ref[0]; s.fl[2 * (size_t)b + 1] = ref[2 * (n - 1)];
    CHECK(cilqr_default_control_seq(s.N, &s.U[2 * (size_t)s.N * b]));
    for (int o = 0; o < s.M; ++o)
      for (int t = 0; t < s.N; ++t) {
        double* q = &s.pose[(((size_t)b * s.M + o) * s.N + t) * 4];
        q[0] = e[0] + 15 + 12 * o + 0.1 * (b % 3); q[1] = (o % 2) ? -1.0 : 0.8; q[2] = 0; q[3] = 0.1 * o;
        double* d = &s.dim[(((size_t)b * s.M + o) * s.N + t) * 2];
        d[0] = 4.79; d[1] = 2.16;
      }
  }
  return 0;
}

struct Result {
  Buf<double> U, X, J;
  Buf<int32_t> it, st;
  Result(const Scene& s) : U(2 * (size_t)s.N * s.B, s.pin), X(4 * (size_t)(s.N + 1) * s.B, s.pin), J(s.B, s.pin), it(s.B, s.pin), st(s.B, s.pin) {}
};

long host_pick(Result& r, int B) {
  long best = -1;
  for (int b = 0; b < B; ++b)
    if (r.J[b] == r.J[b] && (best < 0 || r.J[b] < r.J[best])) best = b;
  return best;
}

bool equal_range(Result& a, Result& b, const Scene& s, int first, int count) {
  bool same = true;
  const size_t N = s.N;
  for (size_t i = 2 * N * first; i < 2 * N * (size_t)(first + count); ++i) same = same && a.U[i] == b.U[i];
  for (size_t i = 4 * (N + 1) * first; i < 4 * (N + 1) * (size_t)(first + count); ++i) same = same && a.X[i] == b.X[i];
  for (int q = first; q < first + count; ++q) same = same && a.J[q] == b.J[q] && a.it[q] == b.it[q] && a.st[q] == b.st[q];
  return same;
}

int solve_single(const cilqr_params& p, Scene& s, Result& r, int device) {
  memcpy(r.U.data(), s.U.data(), sizeof(double) * s.U.n);
  cilqr_handle* h = nullptr;
  CHECK(cilqr_create(&p, s.B, s.N, s.M, device, &h));
  CHECK(cilqr_solve_batch(h, s.B, s.N, s.M, s.x0.data(), r.U.data(), s.poly.data(), s.fl.data(), s.pose.data(), s.dim.data(), nullptr, r.X.data(),
                          r.J.data(), r.it.data(), r.st.data(), CILQR_FLAG_NONE));
  CHECK(cilqr_destroy(h));
  return 0;
}

// ---- one process per GPU ------------------------------------------------------------------------------------------------------
int rank_route(const cilqr_params& p, Scene& s, int rank, int world, int device) {
  const char* port = getenv("MASTER_PORT");
  std::string idfile = getenv("CILQR_ID_FILE") ? getenv("CILQR_ID_FILE") : std::string("/tmp/cilqr_comm_id_") + (port ? port : "0");
  unsigned char id[CILQR_COMM_ID_BYTES];
  if (rank == 0) {
    CHECK(cilqr_comm_unique_id(id));
    const std::string tmp = idfile + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id)) { fprintf(stderr, "cannot write %s\n", tmp.c_str()); return 2; }
    fclose(f);
    rename(tmp.c_str(), idfile.c_str());  // atomic: a reader sees the whole id or no file
  } else {
    FILE* f = nullptr;
    for (int tries = 0; tries < 600 && !(f = fopen(idfile.c_str(), "rb")); ++tries) std::this_thread::sleep_for(std::chrono::milliseconds(100));
    if (!f || fread(id, 1, sizeof(id), f) != sizeof(id)) { fprintf(stderr, "rank %d: no id in %s\n", rank, idfile.c_str()); return 2; }
    fclose(f);
  }
  int first = 0, count = 0;
  CHECK(cilqr_shard_range(s.B, world, rank, &first, &count));
  Result ref(s), mine(s);
  if (solve_single(p, s, ref, device)) return 2;  // the whole batch on this rank's device: what the shards must reproduce
  const long want = host_pick(ref, s.B);
  cilqr_handle* h = nullptr;
  CHECK(cilqr_create(&p, count > 0 ? count : 1, s.N, s.M, device, &h));
  CHECK(cilqr_comm_init_rank(h, world, rank, id));
  memcpy(mine.U.data(), s.U.data(), sizeof(double) * s.U.n);
  const size_t f = first, N = s.N, M = s.M;
  if (count > 0)
    CHECK(cilqr_solve_batch(h, count, s.N, s.M, s.x0.data() + 4 * f, mine.U.data() + 2 * N * f, s.poly.data() + 6 * f, s.fl.data() + 2 * f,
                            s.pose.data() + f * M * N * 4, s.dim.data() + f * M * N * 2, nullptr, mine.X.data() + 4 * (N + 1) * f, mine.J.data() + f,
                            mine.it.data() + f, mine.st.data() + f, CILQR_FLAG_NONE));
  double *dJ = nullptr, *dpair = nullptr;
  if (hipSetDevice(device) != hipSuccess || hipMalloc((void**)&dJ, sizeof(double) * (count > 0 ? count : 1)) != hipSuccess ||
      hipMalloc((void**)&dpair, 2 * sizeof(double)) != hipSuccess)
    return 2;
  if (count > 0 && hipMemcpy(dJ, mine.J.data() + f, sizeof(double) * count, hipMemcpyHostToDevice) != hipSuccess) return 2;
  double pair[2] = {0, 0};
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(cilqr_argmin_global_device(h, nullptr, count, dJ, first, dpair));
    if (hipMemcpy(pair, dpair, sizeof(pair), hipMemcpyDeviceToHost) != hipSuccess) return 2;
  }
  (void)hipFree(dJ); (void)hipFree(dpair);
  const int ranks = cilqr_comm_size(h);
  CHECK(cilqr_destroy(h));
  const bool same = equal_range(ref, mine, s, first, count) && (long)pair[1] == want && pair[0] == ref.J[want] && ranks == world;
  printf("{\"mode\": \"one process per GPU\", \"rank\": %d, \"world\": %d, \"device\": %d, \"B\": %d, \"shard\": [%d, %d], \"best_single\": %ld, "
         "\"best_global\": %ld, \"bit_equal\": %s}\n", rank, world, device, s.B, first, count, want, (long)pair[1], same ? "true" : "false");
  if (rank == 0) remove(idfile.c_str());
  return same ? 0 : 1;
}

}  // namespace

int main(int argc, char** argv) {
  int B = 37, shards = 0, fail_at = -1;
  bool pin = false;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--B") && i + 1 < argc) B = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--shards") && i + 1 < argc) shards = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--fail") && i + 1 < argc) fail_at = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--pinned")) pin = true;
    else if (argv[i][0] != '-') B = atoi(argv[i]);  // (old usage: multi_device B)
    else { fprintf(stderr, "usage: multi_device [--B n] [--shards n] [--pinned] [--fail d]\n"); return 2; }
  }
  const int N = 50, M = 4;
  const int n_present = cilqr_device_count();
  if (n_present < 1) { fprintf(stderr, "no gfx950 device\n"); return 2; }
  cilqr_params p;
  cilqr_params_default(&p);
  p.horizon = N;
  Scene s(B, N, M, pin);
  if (fill(s, p)) return 2;
  if (getenv("WORLD_SIZE") && getenv("RANK")) {
    const int world = atoi(getenv("WORLD_SIZE")), rank = atoi(getenv("RANK"));
    const int device = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : rank % n_present;
    return rank_route(p, s, rank, world, device);
  }

  Result one(s), many(s);
  if (solve_single(p, s, one, 0)) return 2;
  const long best1 = host_pick(one, B);
  bool same = true;
  int cases = 0, failures_seen = 0, rccl_cases = 0;
  std::vector<int> counts;
  if (shards > 0) counts.push_back(shards);
  else for (int n = 1; n <= n_present; ++n) counts.push_back(n);
  for (int n : counts) {
    std::vector<int> devs(n);
    for (int d = 0; d < n; ++d) devs[d] = shards > 0 ? 0 : d;
    int per = 0, f0 = 0;
    CHECK(cilqr_shard_range(B, n, 0, &f0, &per));  // shard 0 is a largest one
    cilqr_multi* m = nullptr;
    CHECK(cilqr_create_multi(&p, per > 0 ? per : 1, N, M, n, devs.data(), &m));
    rccl_cases += cilqr_multi_uses_rccl(m);
    for (int rep = 0; rep < 3; ++rep) {  // several calls: communicators, arenas and staging are reused
      const bool inject = fail_at >= 0 && fail_at < n && rep == 1;
      if (inject) CHECK(cilqr_debug_fail_enqueue(cilqr_multi_handle(m, fail_at), 1));
      memcpy(many.U.data(), s.U.data(), sizeof(double) * s.U.n);
      int64_t best = -2;
      double bestJ = 0.0;
      const int rc = cilqr_multi_solve_batch(m, B, N, M, s.x0.data(), many.U.data(), s.poly.data(), s.fl.data(), s.pose.data(), s.dim.data(), nullptr,
                                             many.X.data(), many.J.data(), many.it.data(), many.st.data(), CILQR_FLAG_NONE, &best, &bestJ);
      if (inject) {  // the forced failure must surface as an error — and leave nothing behind: the next repetition must succeed
        if (rc == CILQR_OK || !strstr(cilqr_last_error(), "forced failure")) { fprintf(stderr, "forced failure not reported (rc %d: %s)\n", rc, cilqr_last_error()); same = false; }
        ++failures_seen;
        continue;
      }
      if (rc != CILQR_OK) { fprintf(stderr, "cilqr_multi_solve_batch (n = %d, rep %d): %s\n", n, rep, cilqr_last_error()); return 2; }
      same = same && best1 == best && (best1 < 0 || bestJ == one.J[best1]) && equal_range(one, many, s, 0, B);
      ++cases;
    }
    CHECK(cilqr_multi_destroy(m));
  }
  printf("{\"mode\": \"one process, every GPU\", \"devices\": %d, \"shards\": %d, \"B\": %d, \"pinned\": %s, \"cases\": %d, \"rccl_cases\": %d, "
         "\"forced_failures\": %d, \"best_single\": %ld, \"bit_equal\": %s}\n", n_present, shards, B, pin ? "true" : "false", cases, rccl_cases,
         failures_seen, best1, same ? "true" : "false");
  return same ? 0 : 1;
}
