// kexp4 -- what does a per-edge gather of 512-byte rows from a table FAR beyond the caches (BASELINE.json configs[4]: 57 GB) cost on
// this part, piece by piece?  Round 5 concluded "4.7-4.9 TB/s whatever the tile shape" from runs of the product kernel alone; round 6's
// box probe (geot_profile_box_rows) read the same table at 6.4-6.6 TB/s.  This program takes the difference apart:
//   hash   row ids from a hash (no index traffic, nothing written): loads in flight per lane x workgroups per CU x nt
//   idx    row ids from an int64 index stream (the operator's src_index), nothing written
//   op     idx + fixed runs of R edges summed and written out (the whole gather_scatter with trivial segment handling)
// hipcc -O3 --offload-arch=gfx950 tools/kexp4.hip -o tools/kexp4;   ./tools/kexp4 [table GB = 56.9] [edges = 100e6]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void hash_rows(const f4 *__restrict__ p, float *sink, unsigned long long rows, int steps, unsigned seed) {
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  const unsigned group = tid >> 5, lane = tid & 31;
  f4 s = {0, 0, 0, 0};
  unsigned long long x = ((unsigned long long)group << 32) ^ seed;
  for (int i = 0; i < steps; i += U) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      x = x * 6364136223846793005ull + 1442695040888963407ull;
      const unsigned long long r = ((x >> 32) * rows) >> 32;
      const f4 *q = p + r * 32 + lane;
      v[u] = NT ? __builtin_nontemporal_load(q) : *q;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u];
  }
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) sink[0] = s[0];
}

__global__ void fill_index(int64_t *idx, int64_t n, unsigned long long rows, unsigned seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned long long x = (unsigned long long)i * 0x9E3779B97F4A7C15ull + seed;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    idx[i] = (int64_t)(((x >> 32) * rows) >> 32);
  }
}

// A lane group of 32 lanes owns CG = 32 consecutive edges per step (its lanes load one index each), gathers their rows U at a
// time; WRITE: every run of RUN consecutive edges is one output row (sum), written with a plain 16-byte store per lane.
template <int U, bool NT, int WRITE, int RUN>
__global__ __launch_bounds__(256) void idx_rows(const f4 *__restrict__ p, const int64_t *__restrict__ idx, f4 *__restrict__ out, float *sink,
                                                 int64_t nnz, int64_t edges_per_group) {
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  const int64_t group = tid >> 5;
  const int lane = tid & 31;
  const int64_t e0 = group * edges_per_group;
  f4 s = {0, 0, 0, 0};
  f4 hold[4];
  int held = 0;
  if (e0 >= nnz) return;
  int64_t my = idx[e0 + lane];
  for (int64_t off = 0; off < edges_per_group; off += 32) {
    const int64_t nxt = (off + 32 < edges_per_group) ? idx[e0 + off + 32 + lane] : 0;   // next chunk's indices under the gathers
#pragma unroll
    for (int b = 0; b < 32; b += U) {
      f4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = __shfl(my, b + u, 32);
        const f4 *q = p + r * 32 + lane;
        v[u] = NT ? __builtin_nontemporal_load(q) : *q;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        s += v[u];
        if (WRITE && ((b + u + 1) % RUN) == 0) {
          const int64_t row = (e0 + off + b + u) / RUN;
          if (WRITE == 1) __builtin_nontemporal_store(s, out + row * 32 + lane);
          else if (WRITE == 2) out[row * 32 + lane] = s;
          else {                                   // 3 / 4: four finished rows leave together (2 KB contiguous)
            hold[held++] = s;
            if (held == 4) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                if (WRITE == 3) __builtin_nontemporal_store(hold[q], out + (row - 3 + q) * 32 + lane);
                else out[(row - 3 + q) * 32 + lane] = hold[q];
              }
              held = 0;
            }
          }
          s = f4{0, 0, 0, 0};
        }
      }
    }
    my = nxt;
  }
  if (!WRITE && s[0] + s[1] + s[2] + s[3] == 123.456f) sink[0] = s[0];
}

template <typename F> static float best_ms(F launch, int iters = 4) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int it = 0; it < iters + 1; ++it) {
    CK(hipEventRecord(a));
    launch(it);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (it > 0 && ms < best) best = ms;
  }
  CK(hipGetLastError());
  return best;
}

int main(int argc, char **argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 56.9;
  const int64_t nnz = (int64_t)(argc > 2 ? atof(argv[2]) : 100e6) / 8192 * 8192;
  const unsigned long long rows = (unsigned long long)(gb * 1e9 / 512);
  f4 *table; float *sink; int64_t *idx; f4 *out;
  CK(hipMalloc(&table, rows * 512)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&idx, nnz * 8)); CK(hipMalloc(&out, (nnz / 8 + 64) * 512));
  CK(hipMemset(table, 0, rows * 512));
  fill_index<<<4096, 256>>>(idx, nnz, rows, 7);
  CK(hipDeviceSynchronize());
  printf("# table %.1f GB = %llu rows of 512 B; %lld edges\n", gb, rows, (long long)nnz);
  // ---- hash: loads in flight x workgroups per CU x nt
  for (int wg : {4}) {
    const int grid = 256 * wg;
    const int64_t groups = (int64_t)grid * 8;
    const int steps = (int)((int64_t(16) << 30) / 512 / groups / 16 * 16);
    const double bytes = (double)groups * steps * 512;
#define HASH(U, NT) { const float ms = best_ms([&](int it) { hash_rows<U, NT><<<grid, 256>>>(table, sink, rows, steps, 99u + it); }); \
      printf("hash  U=%2d nt=%d workgroups/CU=%d (%3d KB in flight per CU): %7.3f ms  %6.2f TB/s\n", U, NT, wg, U * 4 * wg, ms, bytes / ms / 1e9); }
    HASH(4, false) HASH(4, true) HASH(8, false) HASH(8, true) HASH(16, false) HASH(16, true)
  }
  // ---- idx / op: index stream, then the whole operator with fixed runs
  for (int wg : {4}) {
    const int grid = 256 * wg;
    const int64_t groups = (int64_t)grid * 8;
    const int64_t epg = nnz / groups / 32 * 32;
    const int64_t used = epg * groups;
    const double bytes = (double)used * 512;
#define IDX(U, NT, WR, RUN) { const float ms = best_ms([&](int) { idx_rows<U, NT, WR, RUN><<<grid, 256>>>(table, idx, out, sink, used, epg); }); \
      printf("%s U=%2d nt=%d workgroups/CU=%d runs of %2d: %7.3f ms  %6.2f TB/s of rows  (%.2f G edges/s)\n", WR == 0 ? "idx  " : (WR == 1 ? "op nt store   " : (WR == 2 ? "op plain store" : (WR == 3 ? "op 4 rows, nt " : "op 4 rows, pl "))), U, NT, wg, RUN, ms, bytes / ms / 1e9, used / ms / 1e6); }
    IDX(16, true, 0, 16) IDX(16, false, 0, 16)
    IDX(16, true, 1, 16) IDX(16, true, 2, 16) IDX(16, true, 3, 16) IDX(16, true, 4, 16)
    IDX(16, true, 1, 8) IDX(16, true, 2, 8) IDX(16, true, 3, 8) IDX(16, true, 4, 8)
    IDX(16, true, 1, 32) IDX(16, true, 2, 32)
    IDX(16, false, 2, 16)
  }
  printf("# done\n");
  return 0;
}
