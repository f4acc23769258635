// kbench.cpp -- standalone developer harness for libgeot_hip.so (no torch, starts in ms).
//
//   kbench check                 edge-case sweep of all ops against naive host loops
//   kbench sweep [nnz keys feat] time the tile kernel over tuning variants on the power-law
//                                workload of BASELINE.json configs[1] (10M edges, 1M keys, F=64)
//   kbench copy                  float4 copy / read-only ceilings measured the same way
//
// Build: hipcc -O2 --offload-arch=gfx950 -Iinclude tools/kbench.cpp -Lgeot_amd -lgeot_hip -o tools/kbench
// The host loops here are the harness's own 5-line checks (the CPU oracle lives in oracle/ and is
// only used by tests/).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include "geot_hip.h"
#include "geot_hip_dev.h"

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) {                                                                     \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);    \
      exit(2);                                                                                  \
    }                                                                                           \
  } while (0)

template <typename T> struct DBuf {
  T *p = nullptr;
  size_t n = 0;
  explicit DBuf(size_t n_) : n(n_) { CK(hipMalloc(&p, std::max<size_t>(n * sizeof(T), 256))); }
  DBuf(const std::vector<T> &h) : DBuf(h.size()) { up(h); }
  ~DBuf() { (void)hipFree(p); }
  void up(const std::vector<T> &h) { if (!h.empty()) CK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); }
  std::vector<T> down() const {
    std::vector<T> h(n);
    if (n) CK(hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost));
    return h;
  }
};

// power-law sorted keys: w_k ~ rank^(-1/1.5), ranks randomly permuted
static std::vector<int64_t> powerlaw_index(int64_t nnz, int64_t K, uint64_t seed) {
  std::mt19937_64 rng(seed);
  std::vector<double> cdf(K);
  std::vector<int64_t> perm(K);
  std::iota(perm.begin(), perm.end(), 0);
  std::shuffle(perm.begin(), perm.end(), rng);
  double s = 0;
  for (int64_t r = 0; r < K; ++r) { s += std::pow((double)(r + 1), -1.0 / 1.5); cdf[r] = s; }
  std::vector<int64_t> cnt(K, 0);
  std::uniform_real_distribution<double> U(0.0, s);
  for (int64_t i = 0; i < nnz; ++i) {
    const double u = U(rng);
    const int64_t r = std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin();
    cnt[perm[std::min<int64_t>(r, K - 1)]]++;
  }
  if (cnt[K - 1] == 0) { // force index[-1] = K-1
    int64_t big = std::max_element(cnt.begin(), cnt.end()) - cnt.begin();
    cnt[big]--; cnt[K - 1]++;
  }
  std::vector<int64_t> idx; idx.reserve(nnz);
  for (int64_t k = 0; k < K; ++k) idx.insert(idx.end(), cnt[k], k);
  return idx;
}

static double max_rel_err(const std::vector<float> &got, const std::vector<double> &ref,
                          const std::vector<double> &mag) {
  double worst = 0;
  for (size_t i = 0; i < ref.size(); ++i) {
    const double d = std::fabs((double)got[i] - ref[i]);
    const double m = mag[i] > 1e-30 ? mag[i] : 1e-30;
    const double r = d / m;
    if (d > 0 && r > worst) worst = r;
    if (std::isnan(got[i])) return 1e30;
  }
  return worst;
}

struct Case { int64_t nnz, K, F, H; int mode; bool sorted; std::string name; };

static int g_fail = 0;

// mode 0 index_scatter, 1 gs, 2 gws, 3 mh edge-major, 4 mh head-major
static void run_case(const Case &c, const std::vector<int64_t> &dst_index, uint64_t seed) {
  std::mt19937_64 rng(seed);
  std::uniform_real_distribution<float> U(0.f, 1.f);
  const int64_t nnz = c.nnz, F = c.F * c.H, K = c.K;
  const int64_t src_rows = c.mode == 0 ? nnz : std::max<int64_t>(K, 3);
  std::vector<float> src((size_t)src_rows * F);
  for (auto &x : src) x = U(rng);
  std::vector<int64_t> sidx(nnz);
  for (auto &x : sidx) x = (int64_t)(rng() % (uint64_t)src_rows);
  std::vector<float> w((size_t)nnz * c.H);
  for (auto &x : w) x = U(rng);

  std::vector<double> ref((size_t)K * F, 0.0), mag((size_t)K * F, 0.0);
  for (int64_t e = 0; e < nnz; ++e) {
    const int64_t d = dst_index[e];
    if (d < 0 || d >= K) continue;
    const int64_t r = c.mode == 0 ? e : sidx[e];
    for (int64_t f = 0; f < F; ++f) {
      double v = src[(size_t)r * F + f];
      if (c.mode == 2) v *= w[e];
      if (c.mode == 3) v *= w[(size_t)e * c.H + f / c.F];
      if (c.mode == 4) v *= w[(size_t)(f / c.F) * nnz + e];
      ref[(size_t)d * F + f] += v;
      mag[(size_t)d * F + f] += std::fabs(v);
    }
  }
  DBuf<int64_t> d_di(dst_index), d_si(sidx);
  DBuf<float> d_src(src), d_w(w), d_dst((size_t)K * F);
  CK(hipMemset(d_dst.p, 0xFF, std::max<size_t>((size_t)K * F * 4, 4))); // poison: NaN pattern
  const size_t wsb = c.mode >= 3 ? geot_mh_workspace_bytes(nnz, c.H, c.F, K, GEOT_F32) : geot_workspace_bytes(nnz, F, K, GEOT_F32);
  DBuf<char> ws(wsb);
  CK(hipMemset(ws.p, 0xFF, wsb)); // poison everything, then zero only the control words
  geot_workspace_init(ws.p, wsb, nullptr);
  int rc = 0;
  switch (c.mode) {
  case 0: rc = geot_index_scatter(d_di.p, d_src.p, d_dst.p, nnz, F, K, GEOT_F32, c.sorted, ws.p, wsb, nullptr); break;
  case 1: rc = geot_gather_scatter(d_si.p, d_di.p, d_src.p, d_dst.p, nnz, F, src_rows, K, GEOT_F32, ws.p, wsb, nullptr); break;
  case 2: rc = geot_gather_weight_scatter(d_si.p, d_di.p, d_w.p, d_src.p, d_dst.p, nnz, F, src_rows, K, GEOT_F32, ws.p, wsb, nullptr); break;
  case 3: rc = geot_mh_spmm(d_si.p, d_di.p, d_w.p, d_src.p, d_dst.p, nnz, c.H, c.F, src_rows, K, GEOT_W_EDGE_MAJOR, GEOT_F32, ws.p, wsb, nullptr); break;
  case 4: rc = geot_mh_spmm(d_si.p, d_di.p, d_w.p, d_src.p, d_dst.p, nnz, c.H, c.F, src_rows, K, GEOT_W_HEAD_MAJOR, GEOT_F32, ws.p, wsb, nullptr); break;
  }
  CK(hipDeviceSynchronize());
  if (rc != GEOT_OK) { printf("FAIL %-40s rc=%d %s\n", c.name.c_str(), rc, geot_last_error()); g_fail++; return; }
  const auto got = d_dst.down();
  const double err = max_rel_err(got, ref, mag);
  // exact-zero check for empty rows
  bool zero_ok = true;
  for (size_t i = 0; i < ref.size(); ++i) if (mag[i] == 0.0 && got[i] != 0.0f) { zero_ok = false; break; }
  const bool ok = err < 2e-6 && zero_ok;
  if (!ok) g_fail++;
  printf("%s %-44s nnz=%-8ld K=%-8ld F=%-4ld relerr=%.2e zero_rows=%s\n", ok ? "ok  " : "FAIL", c.name.c_str(),
         (long)nnz, (long)K, (long)F, err, zero_ok ? "ok" : "BAD");
}

static std::vector<int64_t> uniform_sorted(int64_t nnz, int64_t K, uint64_t seed, bool force_last = true) {
  std::mt19937_64 rng(seed);
  std::vector<int64_t> v(nnz);
  for (auto &x : v) x = (int64_t)(rng() % (uint64_t)K);
  std::sort(v.begin(), v.end());
  if (force_last && nnz > 0) v[nnz - 1] = K - 1;
  return v;
}

static int cmd_check() {
  uint64_t seed = 1;
  const int64_t feats[] = {1, 2, 3, 4, 7, 8, 16, 31, 32, 33, 64, 100, 128, 256, 260, 512};
  for (int64_t F : feats) {
    run_case({5000, 700, F, 1, 0, true, "index_scatter uniform F=" + std::to_string(F)}, uniform_sorted(5000, 700, seed), seed); seed++;
  }
  for (int cg : {0, 8, 16, 64, 128}) {
    geot_tune(cg, 0, -1, -1);
    run_case({100000, 10000, 32, 1, 0, true, "cfg1 100k x32 -> 10k cg=" + std::to_string(cg)}, uniform_sorted(100000, 10000, seed), seed); seed++;
    run_case({40000, 37, 64, 1, 0, true, "long segments cg=" + std::to_string(cg)}, uniform_sorted(40000, 37, seed), seed); seed++;
  }
  geot_tune(0, 0, -1, -1);
  for (int vec : {1, 2, 4}) for (int l : {-1, 5, 6}) {
    geot_tune(0, vec, -1, l);
    run_case({30000, 2500, 64, 1, 0, true, "F=64 vec=" + std::to_string(vec) + " lpr_log2=" + std::to_string(l)}, uniform_sorted(30000, 2500, seed), seed); seed++;
  }
  for (int nt : {0, 1, 2, 3}) {
    geot_tune(0, 0, nt, -1);
    run_case({30000, 2500, 64, 1, 0, true, "F=64 nontemporal policy " + std::to_string(nt)}, uniform_sorted(30000, 2500, seed), seed); seed++;
  }
  geot_tune(0, 0, -1, -1);
  { // single segment (hub over many tiles)
    std::vector<int64_t> v(50000, 0);
    run_case({50000, 1, 64, 1, 0, true, "single segment key 0"}, v, seed); seed++;
    std::vector<int64_t> v2(50000, 5);
    run_case({50000, 6, 64, 1, 0, true, "single segment key 5 (rows 0-4 empty)"}, v2, seed); seed++;
  }
  for (int hub : {1, 0}) { // hub chains across whole 64-tile windows: window sums (seg_wsum_kernel) forced on / off
    geot_set_option("hub", hub);
    const std::string tag = hub ? " [window sums]" : " [tile walk]";
    for (int64_t F : {64, 5, 100}) {
      std::vector<int64_t> one(300000, 0);
      run_case({300000, 1, F, 1, 0, true, "one key, 300k edges F=" + std::to_string(F) + tag}, one, seed); seed++;
    }
    { // three hubs with short runs between them, chain ends at awkward offsets
      std::vector<int64_t> v;
      for (int i = 0; i < 70001; ++i) v.push_back(2);
      for (int i = 0; i < 300; ++i) v.push_back(3 + i / 7);
      for (int i = 0; i < 131072 + 513; ++i) v.push_back(50);
      for (int i = 0; i < 99; ++i) v.push_back(51 + i);
      for (int i = 0; i < 200000; ++i) v.push_back(400);
      const int64_t n = (int64_t)v.size();
      run_case({n, 401, 64, 1, 0, true, "three hubs + short runs F=64" + tag}, v, seed); seed++;
      run_case({n, 401, 32, 1, 1, true, "three hubs gather_scatter F=32" + tag}, v, seed); seed++;
      run_case({n, 401, 16, 1, 2, true, "three hubs gws F=16" + tag}, v, seed); seed++;
      geot_tune(16, 0, -1, -1); // 64-edge tiles at F=64: thousands of tiles, hundreds of windows
      run_case({n, 401, 64, 1, 0, true, "three hubs, small tiles F=64" + tag}, v, seed); seed++;
      geot_tune(0, 0, -1, -1);
    }
  }
  geot_set_option("hub", -1);
  { // all-unit segments
    std::vector<int64_t> v(20000);
    std::iota(v.begin(), v.end(), 0);
    run_case({20000, 20000, 64, 1, 0, true, "arange (unit segments)"}, v, seed); seed++;
    for (auto &x : v) x *= 3;
    run_case({20000, 3 * 19999 + 1, 32, 1, 0, true, "3*arange (small gaps everywhere)"}, v, seed); seed++;
    for (auto &x : v) x = x / 3 * 40;
    run_case({20000, 40 * 19999 + 1, 8, 1, 0, true, "40*arange (large gaps everywhere)"}, v, seed); seed++;
  }
  { // first key > 0 with a large leading gap, big gap in the middle, nnz = 1
    std::vector<int64_t> v = uniform_sorted(9000, 300, seed);
    for (auto &x : v) x += (x >= 150 ? 100000 : 5000);
    run_case({9000, v.back() + 1, 64, 1, 0, true, "leading gap 5000 + middle gap 100k"}, v, seed); seed++;
    std::vector<int64_t> one = {41};
    run_case({1, 42, 64, 1, 0, true, "nnz=1 key 41"}, one, seed); seed++;
    std::vector<int64_t> one0 = {0};
    run_case({1, 1, 7, 1, 0, true, "nnz=1 key 0 F=7"}, one0, seed); seed++;
  }
  { // hub + power law
    auto v = powerlaw_index(300000, 20000, 7);
    run_case({300000, 20000, 64, 1, 0, true, "power-law 300k -> 20k"}, v, seed); seed++;
    run_case({300000, 20000, 64, 1, 1, true, "gather_scatter power-law"}, v, seed); seed++;
    run_case({300000, 20000, 128, 1, 2, true, "gather_weight_scatter power-law F=128"}, v, seed); seed++;
    run_case({300000, 20000, 64, 4, 3, true, "mh_spmm [nnz,H] H=4 F=64"}, v, seed); seed++;
    run_case({300000, 20000, 64, 4, 4, true, "mh_spmm [H,nnz] H=4 F=64"}, v, seed); seed++;
    run_case({300000, 20000, 6, 3, 3, true, "mh_spmm [nnz,H] H=3 F=6"}, v, seed); seed++;
  }
  for (int64_t F : {1, 5, 32, 100}) {
    auto v = uniform_sorted(1000, 100, seed);
    run_case({1000, 100, F, 1, 1, true, "gather_scatter ref-test shape F=" + std::to_string(F)}, v, seed); seed++;
    run_case({1000, 100, F, 1, 2, true, "gws ref-test shape F=" + std::to_string(F)}, v, seed); seed++;
  }
  { // unsorted (atomics)
    std::mt19937_64 rng(99);
    std::vector<int64_t> v(100000);
    for (auto &x : v) x = (int64_t)(rng() % 5000);
    v.back() = 4999;
    run_case({100000, 5000, 64, 1, 0, false, "unsorted atomics F=64"}, v, seed); seed++;
    run_case({100000, 5000, 33, 1, 0, false, "unsorted atomics F=33"}, v, seed); seed++;
    std::vector<int64_t> few(200000);
    for (auto &x : few) x = (int64_t)(rng() % 10);
    few.back() = 9;
    run_case({200000, 10, 32, 1, 0, false, "unsorted, 10 keys (LDS-binned) F=32"}, few, seed); seed++;
    run_case({200000, 10, 7, 1, 0, false, "unsorted, 10 keys (LDS-binned) F=7"}, few, seed); seed++;
    for (auto &x : few) x = (int64_t)(rng() % 300);
    few.back() = 299;
    run_case({200000, 300, 40, 1, 0, false, "unsorted, 300 keys x 40 (LDS-binned, 48 KB)"}, few, seed); seed++;
    auto s = uniform_sorted(1000, 10, seed);
    run_case({1000, 10, 32, 1, 0, false, "reference test: sorted data, sorted=False"}, s, seed); seed++;
  }
  printf("%s (%d failures)\n", g_fail ? "CHECK FAILED" : "CHECK PASSED", g_fail);
  return g_fail ? 1 : 0;
}

__global__ void copy4_kernel(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void read4_kernel(const float4 *__restrict__ a, float *out, size_t n) {
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = a[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) out[0] = s;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

static int cmd_copy() {
  const size_t n = (size_t)2560 << 20; // bytes
  DBuf<char> A(n), B(n);
  CK(hipMemset(A.p, 1, n));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int blocks : {2048, 4096, 8192, 16384}) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(copy4_kernel, dim3(blocks), dim3(256), 0, 0, (const float4 *)A.p, (float4 *)B.p, n / 16);
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(copy4_kernel, dim3(blocks), dim3(256), 0, 0, (const float4 *)A.p, (float4 *)B.p, n / 16);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    double ms = time_ms(e0, e1) / 10;
    printf("copy4  blocks=%-6d %.3f ms  %.2f TB/s (read+write)\n", blocks, ms, 2.0 * n / ms / 1e9);
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(read4_kernel, dim3(blocks), dim3(256), 0, 0, (const float4 *)A.p, (float *)B.p, n / 16);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    ms = time_ms(e0, e1) / 10;
    printf("read4  blocks=%-6d %.3f ms  %.2f TB/s (read only)\n", blocks, ms, 1.0 * n / ms / 1e9);
  }
  return 0;
}

static int cmd_sweep(int64_t nnz, int64_t K, int64_t F, int iters, int one_cg = -1, int one_vec = 0, int one_nt = -1) {
  printf("workload: power-law nnz=%ld keys=%ld feat=%ld\n", (long)nnz, (long)K, (long)F);
  auto idx = powerlaw_index(nnz, K, 0);
  {
    int64_t maxdeg = 0, run = 0, empty = 0, prev = -1, nseg = 0;
    for (int64_t i = 0; i < nnz; ++i) {
      if (idx[i] != prev) { nseg++; empty += idx[i] - prev - 1; prev = idx[i]; run = 0; }
      run++; maxdeg = std::max(maxdeg, run);
    }
    printf("segments=%ld empty_keys=%ld max_degree=%ld\n", (long)nseg, (long)empty, (long)maxdeg);
  }
  std::vector<float> src((size_t)nnz * F);
  {
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    for (auto &x : src) x = U(rng);
  }
  // host reference (double)
  std::vector<double> ref((size_t)K * F, 0.0);
  for (int64_t e = 0; e < nnz; ++e)
    for (int64_t f = 0; f < F; ++f) ref[(size_t)idx[e] * F + f] += src[(size_t)e * F + f];
  DBuf<int64_t> d_idx(idx);
  DBuf<float> d_src(src), d_dst((size_t)K * F);
  const double alg_bytes = (double)nnz * (4.0 * F + 8) + (double)K * 4.0 * F;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct V { int cg, vec, nt, l; };
  std::vector<V> vs;
  for (int cg : {16, 32, 48, 64, 128}) vs.push_back({cg, 4, 1, -1});
  for (int cg : {32, 64}) vs.push_back({cg, 4, 0, -1});
  for (int cg : {32, 64}) vs.push_back({cg, 4, 3, -1});
  for (int cg : {32, 64}) vs.push_back({cg, 4, 2, -1});
  for (int cg : {32, 64, 128}) vs.push_back({cg, 2, 1, -1});
  for (int cg : {64, 128, 256}) vs.push_back({cg, 1, 1, -1});
  vs.push_back({32, 4, 1, 5});
  if (one_cg >= 0) { vs.clear(); vs.push_back({one_cg, one_vec, one_nt, -1}); }
  if (getenv("KB_NTKEYS")) { geot_set_option("nt_keys", atoi(getenv("KB_NTKEYS"))); printf("nt_keys=%s ", getenv("KB_NTKEYS")); }
  if (getenv("KB_LANE_E")) { geot_set_option("lane_e", atoi(getenv("KB_LANE_E"))); printf("lane_e=%s ", getenv("KB_LANE_E")); }
  if (getenv("KB_NARROW")) { geot_set_option("narrow", atoi(getenv("KB_NARROW"))); printf("narrow=%s ", getenv("KB_NARROW")); }
  if (one_cg >= 0 && getenv("KB_UNROLL")) { geot_set_option("unroll", atoi(getenv("KB_UNROLL"))); printf("unroll=%s ", getenv("KB_UNROLL")); }
  for (const V &v : vs) {
    geot_tune(v.cg, v.vec, v.nt, v.l);
    const size_t wsb = geot_workspace_bytes(nnz, F, K, GEOT_F32);
    DBuf<char> ws(wsb);
    geot_workspace_init(ws.p, wsb, nullptr);
    CK(hipMemset(d_dst.p, 0xFF, (size_t)K * F * 4));
    int rc = geot_index_scatter(d_idx.p, d_src.p, d_dst.p, nnz, F, K, GEOT_F32, 1, ws.p, wsb, nullptr);
    CK(hipDeviceSynchronize());
    if (rc) { printf("rc=%d %s\n", rc, geot_last_error()); continue; }
    auto got = d_dst.down();
    double worst = 0;
    for (size_t i = 0; i < ref.size(); ++i) {
      if (ref[i] == 0) { if (got[i] != 0) worst = 1e30; continue; }
      const double d = std::fabs(got[i] - ref[i]) / ref[i];
      if (!(d <= worst)) worst = d; // also catches NaN
    }
    for (int i = 0; i < 3; ++i) geot_index_scatter(d_idx.p, d_src.p, d_dst.p, nnz, F, K, GEOT_F32, 1, ws.p, wsb, nullptr);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) geot_index_scatter(d_idx.p, d_src.p, d_dst.p, nnz, F, K, GEOT_F32, 1, ws.p, wsb, nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    const double ms = time_ms(e0, e1) / iters;
    geot_profile_enable(1); geot_profile_reset();
    for (int i = 0; i < iters; ++i) geot_index_scatter(d_idx.p, d_src.p, d_dst.p, nnz, F, K, GEOT_F32, 1, ws.p, wsb, nullptr);
    double mm, fm, am; int64_t calls;
    geot_profile_read(&mm, &fm, &am, &calls);
    geot_profile_enable(0);
    printf("cg=%-4d vec=%d nt=%d lpr=%-2d | call %.4f ms  %.2f Gedge/s  %.2f TB/s (%.1f%% of 8.0) | main %.4f fixup %.4f aux %.4f ms | relerr %.1e\n",
           v.cg, v.vec, v.nt, v.l, ms, nnz / ms / 1e6, alg_bytes / ms / 1e9, alg_bytes / ms / 1e9 / 8000.0 * 100.0,
           mm / calls, fm / calls, am / calls, worst);
  }
  // atomic (sorted=0) for comparison: what the reference's flush strategy costs here
  for (int avec : {0, 1}) {
    geot_tune(0, avec, -1, -1);
    printf("[atomic path vec=%d] ", avec);
    const size_t wsb = geot_workspace_bytes(nnz, F, K, GEOT_F32);
    DBuf<char> ws(wsb);
    geot_workspace_init(ws.p, wsb, nullptr);
    for (int i = 0; i < 2; ++i) geot_index_scatter(d_idx.p, d_src.p, d_dst.p, nnz, F, K, GEOT_F32, 0, ws.p, wsb, nullptr);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) geot_index_scatter(d_idx.p, d_src.p, d_dst.p, nnz, F, K, GEOT_F32, 0, ws.p, wsb, nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    const double ms = time_ms(e0, e1) / iters;
    printf("sorted=0 (memset + atomic flush)  call %.4f ms  %.2f Gedge/s  %.2f TB/s\n", ms, nnz / ms / 1e6, alg_bytes / ms / 1e9);
  }
  return 0;
}

static int cmd_unsorted(int64_t nnz, int64_t F) {
  std::mt19937_64 rng(5);
  std::vector<float> src((size_t)nnz * F, 0.5f);
  DBuf<float> d_src(src);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int64_t K : {10, 100, 1000, 100000, 1000000}) {
    std::vector<int64_t> idx(nnz);
    for (auto &x : idx) x = (int64_t)(rng() % (uint64_t)K);
    idx.back() = K - 1;
    DBuf<int64_t> d_idx(idx);
    DBuf<float> d_dst((size_t)K * F);
    for (int direct : {0, 1}) {
      geot_tune(0, 0, -1, direct ? 7 : -1);
      const size_t wsb = geot_workspace_bytes(nnz, F, K, GEOT_F32);
      DBuf<char> ws(wsb);
      geot_workspace_init(ws.p, wsb, nullptr);
      for (int i = 0; i < 2; ++i) geot_index_scatter(d_idx.p, d_src.p, d_dst.p, nnz, F, K, GEOT_F32, 0, ws.p, wsb, nullptr);
      CK(hipEventRecord(e0));
      for (int i = 0; i < 5; ++i) geot_index_scatter(d_idx.p, d_src.p, d_dst.p, nnz, F, K, GEOT_F32, 0, ws.p, wsb, nullptr);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      const double ms = time_ms(e0, e1) / 5;
      auto got = d_dst.down();
      double tot = 0; for (float v : got) tot += v;
      printf("unsorted random keys K=%-8ld F=%ld nnz=%ld %s: %.4f ms  %.2f Gedge/s  (sum check %.6g vs %.6g)\n", (long)K, (long)F,
             (long)nnz, direct ? "direct atomics " : (K * F * 4 <= 48 * 1024 ? "LDS-binned     " : "direct (auto)  "), ms, nnz / ms / 1e6, tot, 0.5 * nnz * F);
    }
  }
  geot_tune(0, 0, -1, -1);
  return 0;
}

int main(int argc, char **argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const std::string cmd = argc > 1 ? argv[1] : "check";
  printf("%s\n", geot_build_info());
  if (cmd == "check") return cmd_check();
  if (cmd == "copy") return cmd_copy();
  if (cmd == "unsorted") return cmd_unsorted(argc > 2 ? atoll(argv[2]) : 10000000, argc > 3 ? atoll(argv[3]) : 32);
  if (cmd == "sweep") {
    const int64_t nnz = argc > 2 ? atoll(argv[2]) : 10000000;
    const int64_t K = argc > 3 ? atoll(argv[3]) : 1000000;
    const int64_t F = argc > 4 ? atoll(argv[4]) : 64;
    const int iters = argc > 5 ? atoi(argv[5]) : 20;
    const int one_cg = argc > 6 ? atoi(argv[6]) : -1;
    const int one_vec = argc > 7 ? atoi(argv[7]) : 0;
    const int one_nt = argc > 8 ? atoi(argv[8]) : -1;
    return cmd_sweep(nnz, K, F, iters, one_cg, one_vec, one_nt);
  }
  fprintf(stderr, "usage: kbench check|copy|sweep [nnz keys feat iters]\n");
  return 64;
}
