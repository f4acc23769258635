// kexp2.hip -- round-2 probes (developer tool, not product).
//
//   slab  Can the 12-28x re-gather ratio of the gather modes be cut by SOURCE-BLOCKED processing?  Every wave
//         owns a group of dst rows and walks its edges ordered by (source slab, ...): all waves of an XCD then
//         gather from the same few-MiB slab of the table at the same time (natural lockstep: same start, same
//         work per slab), so the slab is served by that XCD's 4 MiB L2 instead of the Infinity Cache / HBM.
//         The probe gathers `nnz` rows of `rb` bytes from an `n`-row table through a per-wave index stream and
//         compares: uniform-random order (today's access pattern) vs slab order at several slab sizes.
//   mfma  SURVEY section 7's open question: a 32-dst-row x T-edge selector product on the matrix cores
//         (v_mfma_f32_32x32x2_f32: A = selector * weight [32 x 2 edges], B = gathered rows [2 edges x 32 feats])
//         against the VALU walk over the same gathered rows, Reddit-degree tiles.
//
// Build: hipcc -O3 --offload-arch=gfx950 tools/kexp2.hip -o tools/kexp2
// Run:   tools/kexp2 slab [n_rows rb nnz]      tools/kexp2 mfma [n_rows degree]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

static inline uint64_t splitmix(uint64_t &s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// ---- slab probe ---------------------------------------------------------------------------------------------
// wave w of the grid owns edges [w*chunk, (w+1)*chunk) of the index stream; 64 indices per coalesced load, then
// the rows one after the other, U in flight, every lane 16 B of the row (rb/16 lanes per row; rb = 1024 -> the
// whole wave on one row, rb = 512 -> two rows per instruction).
template <int U>
__global__ __launch_bounds__(256) void slab_kernel(const float *__restrict__ table, const int *__restrict__ idx,
                                                     float *__restrict__ sink, int64_t chunk, int rb, int xcd_map) {
  const int lane = threadIdx.x & 63;
  int64_t blk = blockIdx.x;
  if (xcd_map) { // contiguous wave ranges per XCD (blocks are dealt round-robin over the 8 XCDs)
    const int64_t per = gridDim.x / 8;
    if (blk < per * 8) blk = (blk % 8) * per + blk / 8;
  }
  const int64_t w = blk * 4 + (threadIdx.x >> 6);
  const int *my = idx + w * chunk;
  const int lpr = rb / 16;            // lanes per row
  const int rpi = 64 / lpr;           // rows per wave instruction
  const int sub = lane / lpr, c = lane % lpr;
  f4 acc = {0, 0, 0, 0};
  for (int64_t e0 = 0; e0 < chunk; e0 += 64) {
    const int mine = my[e0 + lane];
    for (int j = 0; j < 64; j += U * rpi) {
      f4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = __shfl(mine, j + u * rpi + sub, 64);
        v[u] = *reinterpret_cast<const f4 *>(reinterpret_cast<const char *>(table) + (int64_t)row * rb + c * 16);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u];
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

static int run_slab(int argc, char **argv) {
  const int64_t n = argc > 2 ? atoll(argv[2]) : 232965;
  const int rb = argc > 3 ? atoi(argv[3]) : 1024;
  int64_t nnz = argc > 4 ? atoll(argv[4]) : 114615892;
  const int waves_per_cu = 8;
  const int64_t nwaves_res = 256 * waves_per_cu;              // resident waves
  float *table, *sink;
  CK(hipMalloc(&table, (size_t)n * rb));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(table, 0, (size_t)n * rb));
  printf("table: %lld rows x %d B = %.1f MB; %lld gathers = %.1f GB of rows\n", (long long)n, rb, n * (double)rb / 1e6,
         (long long)nnz, nnz * (double)rb / 1e9);
  int *idx;
  std::vector<int> h;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // rounds: every resident wave processes `rounds` groups one after the other (grid = rounds x resident waves)
  for (int rounds : {1, 7}) {
    const int64_t nwaves = nwaves_res * rounds;
    const int64_t chunk = (nnz / nwaves) / 64 * 64;
    const int64_t tot = chunk * nwaves;
    h.resize(tot);
    CK(hipMalloc(&idx, (size_t)tot * 4));
    for (double slab_mib : {0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 6.0}) {
      const int64_t slab_rows = slab_mib == 0.0 ? n : std::max<int64_t>(1, (int64_t)(slab_mib * 1048576 / rb));
      const int64_t nslabs = (n + slab_rows - 1) / slab_rows;
      uint64_t seed = 42;
      for (int64_t w = 0; w < nwaves; ++w) {
        // the wave's chunk ordered by slab: equal shares per slab (what random sources give on average)
        for (int64_t e = 0; e < chunk; ++e) {
          const int64_t s = e * nslabs / chunk;
          const int64_t lo = s * slab_rows, cnt = std::min(slab_rows, n - lo);
          h[w * chunk + e] = (int)(lo + (int64_t)(splitmix(seed) % (uint64_t)cnt));
        }
      }
      CK(hipMemcpy(idx, h.data(), (size_t)tot * 4, hipMemcpyHostToDevice));
      for (int xcd_map : {0, 1}) {
        const dim3 grid((unsigned)(nwaves / 4));
        float best = 1e30f;
        for (int it = 0; it < 4; ++it) {
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(slab_kernel<8>, grid, dim3(256), 0, 0, table, idx, sink, chunk, rb, xcd_map);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (it > 0 && ms < best) best = ms;
        }
        printf("rounds=%d slab=%s (%lld rows, %lld slabs) xcd_map=%d: %.3f ms  %.2f TB/s of gathered rows  %.2f Gedge/s\n", rounds,
               slab_mib == 0.0 ? "none(uniform)" : (std::to_string(slab_mib) + " MiB").c_str(), (long long)slab_rows,
               (long long)nslabs, xcd_map, best, tot * (double)rb / best / 1e9, tot / best / 1e6);
      }
    }
    CK(hipFree(idx));
  }
  return 0;
}

// ---- MFMA probe -----------------------------------------------------------------------------------------------
// One wave = one tile of 32 consecutive dst rows x `deg` edges each (dst-sorted), F = 64 features.
// VALU: the product kernel's walk - 16 lanes x 16 B per row, 4 rows per wave instruction, fma with the weight,
//       run sums in registers, a store per dst row.
// MFMA: per step of 2 edges, A[m][k] = (dst_local[e_k] == m) ? w[e_k] : 0  (lane l: m = l % 32, k = l / 32),
//       B[k][n] = row(e_k)[n0 + n] (lane l: k = l / 32, n = l % 32: a 4-B load per lane), two 32x32x2 MFMAs
//       (n0 = 0, 32) into two 16-VGPR accumulators; the tile's 32 x 64 result is stored at the end.
__global__ __launch_bounds__(256) void tile_valu_kernel(const float *__restrict__ table, const int *__restrict__ src_index,
                                                          const float *__restrict__ weight, float *__restrict__ out,
                                                          int deg) {
  const int lane = threadIdx.x & 63;
  const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int64_t ebase = tile * 32 * deg;
  for (int r = 0; r < 32; ++r) {
    f4 acc = {0, 0, 0, 0};
    const int64_t e0 = ebase + (int64_t)r * deg;
    for (int j = g; j < deg; j += 4 * 4) {
      f4 v[4];
      float w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int jj = j + 4 * u < deg ? j + 4 * u : deg - 1;
        const int row = src_index[e0 + jj];
        w[u] = j + 4 * u < deg ? weight[e0 + jj] : 0.f;
        v[u] = *reinterpret_cast<const f4 *>(table + (int64_t)row * 64 + c * 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += v[u] * w[u];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] += __shfl_xor(acc[i], 16, 64);
      acc[i] += __shfl_xor(acc[i], 32, 64);
    }
    if (g == 0) *reinterpret_cast<f4 *>(out + (tile * 32 + r) * 64 + c * 4) = acc;
  }
}

__global__ __launch_bounds__(256) void tile_mfma_kernel(const float *__restrict__ table, const int *__restrict__ src_index,
                                                          const float *__restrict__ weight, float *__restrict__ out,
                                                          int deg) {
  const int lane = threadIdx.x & 63;
  const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int k = lane >> 5, mn = lane & 31;
  const int64_t ebase = tile * 32 * deg;
  const int T = 32 * deg;
  f16v acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
  for (int e = 0; e < T; e += 2 * 4) {
    float a[4], b0[4], b1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ee = e + 2 * u + k;                    // this lane's edge of the K-step
      const int ec = ee < T ? ee : T - 1;
      const int row = src_index[ebase + ec];
      const int dl = ec / deg;                         // dst row inside the tile (dst-sorted, equal degrees)
      a[u] = (ee < T && dl == mn) ? weight[ebase + ec] : 0.f;
      b0[u] = table[(int64_t)row * 64 + mn];
      b1[u] = table[(int64_t)row * 64 + 32 + mn];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b0[u], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b1[u], acc1, 0, 0, 0);
    }
  }
  // C layout of 32x32: lane l, register i -> row (i / 4) * 8 + (l / 32) * 4 + i % 4, column l % 32
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int m = (i >> 2) * 8 + k * 4 + (i & 3);
    out[(tile * 32 + m) * 64 + mn] = acc0[i];
    out[(tile * 32 + m) * 64 + 32 + mn] = acc1[i];
  }
}

static int run_mfma(int argc, char **argv) {
  const int64_t n = argc > 2 ? atoll(argv[2]) : 232965;
  const int deg = argc > 3 ? atoi(argv[3]) : 492;
  const int64_t tiles = 256 * 4 * 4;                     // 4 waves per block, 4 blocks per CU
  const int64_t nnz = tiles * 32 * deg;
  std::vector<int> hs(nnz);
  std::vector<float> hw(nnz), ht((size_t)n * 64);
  uint64_t seed = 7;
  for (auto &x : ht) x = (float)(splitmix(seed) % 1000) / 1000.f;
  float *table, *w, *o1, *o2;
  int *si;
  CK(hipMalloc(&table, (size_t)n * 256));
  CK(hipMalloc(&w, nnz * 4));
  CK(hipMalloc(&si, nnz * 4));
  CK(hipMalloc(&o1, tiles * 32 * 256));
  CK(hipMalloc(&o2, tiles * 32 * 256));
  CK(hipMemcpy(table, ht.data(), (size_t)n * 256, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int64_t span : {(int64_t)2048, n}) {             // sources within 2048 rows (L2-resident) / the whole table
    for (int64_t i = 0; i < nnz; ++i) {
      hs[i] = (int)(splitmix(seed) % (uint64_t)span);
      hw[i] = (float)(splitmix(seed) % 1000) / 1000.f;
    }
    CK(hipMemcpy(si, hs.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), nnz * 4, hipMemcpyHostToDevice));
    float best[2] = {1e30f, 1e30f};
    for (int it = 0; it < 4; ++it)
      for (int v = 0; v < 2; ++v) {
        CK(hipEventRecord(e0));
        if (v == 0) hipLaunchKernelGGL(tile_valu_kernel, dim3((unsigned)(tiles / 4)), dim3(256), 0, 0, table, si, w, o1, deg);
        else hipLaunchKernelGGL(tile_mfma_kernel, dim3((unsigned)(tiles / 4)), dim3(256), 0, 0, table, si, w, o2, deg);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best[v]) best[v] = ms;
      }
    std::vector<float> r1(tiles * 32 * 64), r2(tiles * 32 * 64);
    CK(hipMemcpy(r1.data(), o1, r1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r2.data(), o2, r2.size() * 4, hipMemcpyDeviceToHost));
    double maxrel = 0;
    for (size_t i = 0; i < r1.size(); ++i) maxrel = std::max(maxrel, (double)std::abs(r1[i] - r2[i]) / (std::abs(r1[i]) + 1e-6));
    printf("mfma probe: %lld tiles of 32 dst rows x %d edges, F=64, sources within %lld rows: VALU walk %.3f ms (%.2f Gedge/s), "
           "MFMA 32x32x2 selector product %.3f ms (%.2f Gedge/s)  -> MFMA is %.2fx %s; max rel diff %.1e\n",
           (long long)tiles, deg, (long long)span, best[0], nnz / best[0] / 1e6, best[1], nnz / best[1] / 1e6,
           best[1] > best[0] ? best[1] / best[0] : best[0] / best[1], best[1] > best[0] ? "SLOWER" : "faster", maxrel);
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc > 1 && !strcmp(argv[1], "slab")) return run_slab(argc, argv);
  if (argc > 1 && !strcmp(argv[1], "mfma")) return run_mfma(argc, argv);
  fprintf(stderr, "usage: kexp2 slab [n_rows rb nnz] | mfma [n_rows degree]\n");
  return 1;
}
