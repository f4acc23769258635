// seg_plan.hip -- Phase A of the source-blocked kernels (seg_slab.hip) as DEVICE code: the plan of an edge list is built
// by a handful of kernels of this library, a few scans and two radix sorts; the host reads two small records back (the
// sizes of the arrays it has to allocate) and loops over nothing.
//
// What is built (include/geot_hip.h, geot_slab_plan): the dst rows are cut into GROUPS of <= R consecutive virtual rows
// with <= `budget` edges (a row with more than `cap` edges is split into virtual rows whose partial sums meet again in
// carry slots), groups are ordered by size, the edges of a group by (source slab, row in group, original position).
// The reference has no counterpart - its kernels gather per edge (csrc/cuda/mh_spmm_kernel.cuh:28-111); its only
// per-graph preprocessing is the row-pointer histogram of geot/match_replace/format_transform.py:5-25.
//
// The arrays are bit-identical to the ones the host layer's ATen formulation (torch_ops.cpp slab_build_aten, kept for
// CPU tensors and as the cross-check of tests/test_gpu_slab.py) produces; what changed is the cost: ~40 generic ATen
// passes with ~80 B/edge of transient memory and a host loop over the virtual rows (215 ms the first time in a process,
// 31 ms later, at 115 M edges) against three per-edge passes here.
//
//   stage 1 (rows)    row pointers of the ascending dst_index by binary search, rows with edges counted -> budget / cap;
//                     per row: virtual rows, "is split", carry slots -> one exclusive scan of the triple.   [read-back 1]
//   stage 2 (groups)  virtual-row tables; the greedy grouping "take rows while <= R rows and <= budget edges" is a walk
//                     i -> next(i) from 0: next() is computed for every i in parallel (<= R steps each), the set of
//                     group starts is the orbit of 0, found by pointer doubling (log2 V rounds).            [read-back 2]
//   stage 3 (edges)   groups sorted by size (one radix sort), one 32-bit key per edge, one stable radix sort of
//                     (key, edge id), one pass that writes e_src / e_dl / e_perm.                           [asynchronous]
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <cstdint>

#include "geot_hip.h"
#include "internal.h"

namespace {

constexpr int kThreads = 256;

inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }
inline unsigned blocks_for(int64_t n) { return (unsigned)((n + kThreads - 1) / kThreads > 0 ? (n + kThreads - 1) / kThreads : 1); }
inline int ceil_log2_i64(int64_t x) {
  int l = 0;
  while (((int64_t)1 << l) < x) ++l;
  return l;
}
inline int bit_width_u64(uint64_t x) {
  int b = 0;
  while (x) { ++b; x >>= 1; }
  return b < 1 ? 1 : b;
}

#define PLAN_TRY(expr)                                                                                                  \
  do {                                                                                                                  \
    const hipError_t e_ = (expr);                                                                                       \
    if (e_ != hipSuccess) return geot_internal_fail(GEOT_ELAUNCH, hipGetErrorString(e_));                               \
  } while (0)

struct Tri {
  int nv, split, pieces; // virtual rows of the row, 1 if the row is split, its carry slots (= nv if split)
};
struct TriPlus {
  __host__ __device__ Tri operator()(const Tri &a, const Tri &b) const { return Tri{a.nv + b.nv, a.split + b.split, a.pieces + b.pieces}; }
};

// device scalars of a job (int64 each)
enum { S_NONEMPTY = 0, S_BUDGET, S_CAP, S_V, S_NSPLIT, S_NCARRY, S_FIRST, S_LAST, S_G, S_COUNT = 16 };

// ---- stage 1 -----------------------------------------------------------------------------------------------------------
// rowptr[r] = first edge with dst >= r (r = 0 .. out_rows); rows with edges are counted
__global__ __launch_bounds__(kThreads) void plan_rowptr_kernel(const int64_t *__restrict__ dst, int64_t nnz, int64_t out_rows,
                                                                 int32_t *__restrict__ rowptr, int64_t *__restrict__ sc) {
  const int64_t r = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  int64_t lo = 0;
  if (r <= out_rows) {
    int64_t hi = nnz;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (dst[mid] < r) lo = mid + 1;
      else hi = mid;
    }
    rowptr[r] = (int32_t)lo;
  }
  // non-empty rows: row r has edges iff an edge with key r exists at its lower bound
  const bool has = r < out_rows && lo < nnz && dst[lo] == r;
  const unsigned long long b = __ballot(has);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(reinterpret_cast<unsigned long long *>(sc + S_NONEMPTY), (unsigned long long)__popcll(b));
  if (r == 0) {
    sc[S_FIRST] = dst[0];
    sc[S_LAST] = dst[nnz - 1];
  }
}

// budget / cap from the number of rows with edges (the host layer's arithmetic, on the device: no read-back in between)
__global__ void plan_params_kernel(int64_t *sc, int64_t nnz, int64_t R, int64_t units) {
  const int64_t nonempty = sc[S_NONEMPTY];
  int64_t rounds0 = (nonempty + R * units - 1) / (R * units);
  if (rounds0 < 1) rounds0 = 1;
  int64_t budget = (nnz + rounds0 * units - 1) / (rounds0 * units);
  if (budget < 256) budget = 256;
  int64_t cap = budget / 2;
  if (cap < 64) cap = 64;
  sc[S_BUDGET] = budget;
  sc[S_CAP] = cap;
}

__device__ __forceinline__ int row_count(const int32_t *rowptr, int64_t r) {
  const int c = rowptr[r + 1] - rowptr[r];
  return c > 0 ? c : 0; // (an index that is not ascending after all: memory-safe garbage, never a negative count)
}

__global__ __launch_bounds__(kThreads) void plan_row_tri_kernel(const int32_t *__restrict__ rowptr, int64_t out_rows,
                                                                  const int64_t *__restrict__ sc, Tri *__restrict__ tri) {
  const int64_t r = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (r >= out_rows) return;
  const int cap = (int)sc[S_CAP];
  const int c = row_count(rowptr, r);
  const int nv = (c + cap - 1) / cap;
  tri[r] = Tri{nv, nv > 1 ? 1 : 0, nv > 1 ? nv : 0};
}

__global__ void plan_row_totals_kernel(const Tri *tri_in, const Tri *tri_ex, int64_t out_rows, int64_t *sc) {
  const Tri a = tri_ex[out_rows - 1], b = tri_in[out_rows - 1];
  sc[S_V] = (int64_t)a.nv + b.nv;
  sc[S_NSPLIT] = (int64_t)a.split + b.split;
  sc[S_NCARRY] = (int64_t)a.pieces + b.pieces;
}

// ---- stage 2 -----------------------------------------------------------------------------------------------------------
// per row: its virtual rows (dst row, edge count, output slot) and, if it is split, its entry in the split tables
__global__ __launch_bounds__(kThreads) void plan_vrows_kernel(const int32_t *__restrict__ rowptr, const Tri *__restrict__ tri_ex,
                                                                int64_t out_rows, int cap, int32_t *__restrict__ v_cnt,
                                                                int64_t *__restrict__ v_out, int32_t *__restrict__ v_row,
                                                                int32_t *__restrict__ v_total, int64_t *__restrict__ c_row,
                                                                int64_t *__restrict__ c_first, int32_t *__restrict__ c_count,
                                                                int64_t *__restrict__ c_total) {
  const int64_t r = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (r >= out_rows) return;
  const int c = row_count(rowptr, r);
  const int nv = (c + cap - 1) / cap;
  const Tri ex = tri_ex[r];
  const bool split = nv > 1;
  // A row above `cap` edges is split into nv virtual rows by INTERLEAVING: edge j of the row goes to piece j % nv.  (Rounds 2-3 cut
  // contiguous ranges: on an edge list whose sources are SORTED inside every dst row - what a CSR / torch_geometric's coalesce()
  // leaves - each piece then covered one narrow source range, i.e. a few slabs, the groups made of hub pieces had all their
  // edges in a few steps of the sweep, and the lockstep waited for them: configs[3]'s stand-in with sorted sources ran 19.4 ms
  // instead of 8.0.  Interleaved pieces sample the whole row.)
  for (int p = 0; p < nv; ++p) {
    const int64_t v = (int64_t)ex.nv + p;
    v_cnt[v] = c / nv + (p < c % nv ? 1 : 0);
    v_row[v] = (int32_t)r;
    v_total[v] = c;
    v_out[v] = split ? -((int64_t)ex.pieces + p + 1) : r;
  }
  if (split) {
    c_row[ex.split] = r;
    c_first[ex.split] = ex.pieces;
    c_count[ex.split] = nv;
    c_total[ex.split] = c;
  }
}

// next[i]: where the group that starts at virtual row i ends (host layer's greedy rule: <= R rows, <= budget edges, the
// first row always taken); next[V] = V
__global__ __launch_bounds__(kThreads) void plan_next_kernel(const int32_t *__restrict__ v_cnt, int64_t V, int R, int64_t budget,
                                                               int32_t *__restrict__ next, int32_t *__restrict__ reach) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i > V) return;
  reach[i] = i == 0 ? 1 : 0;
  if (i == V) {
    next[i] = (int32_t)V;
    return;
  }
  int64_t j = i, e = 0;
  while (j < V && j - i < R && (j == i || e + v_cnt[j] <= budget)) e += v_cnt[j++];
  next[i] = (int32_t)j;
}

// one round of pointer doubling: everything reachable from 0 in < 2^(k+1) steps is marked after round k
__global__ __launch_bounds__(kThreads) void plan_double_kernel(const int32_t *__restrict__ jump_in, int32_t *__restrict__ jump_out,
                                                                 int32_t *__restrict__ reach, int64_t V) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i > V) return;
  const int32_t j = jump_in[i];
  if (i < V && reach[i] && j < V) reach[j] = 1;
  jump_out[i] = jump_in[j];
}

// ---- stage 3 -----------------------------------------------------------------------------------------------------------
// starts[g] = first virtual row of group g (gid = inclusive scan of the marks); starts[G] = V
__global__ __launch_bounds__(kThreads) void plan_starts_kernel(const int32_t *__restrict__ reach, const int32_t *__restrict__ gidinc,
                                                                 int64_t V, int64_t G, int32_t *__restrict__ starts) {
  const int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (v < V && reach[v]) starts[gidinc[v] - 1] = (int32_t)v;
  if (v == 0) starts[G] = (int32_t)V;
}

// per group: edges, and the sort key "most edges first" (stable LSD sort of budget - edges keeps equal sizes in order)
__global__ __launch_bounds__(kThreads) void plan_group_edges_kernel(const int32_t *__restrict__ starts, const int32_t *__restrict__ vpre,
                                                                      int64_t G, int64_t budget, int32_t *__restrict__ gedges,
                                                                      uint32_t *__restrict__ gkey, uint32_t *__restrict__ gval) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= G) return;
  const int32_t e = vpre[starts[g + 1]] - vpre[starts[g]];
  gedges[g] = e;
  const int64_t k = budget - e;
  gkey[g] = (uint32_t)(k < 0 ? 0 : k);
  gval[g] = (uint32_t)g;
}

__global__ __launch_bounds__(kThreads) void plan_group_tables_kernel(const uint32_t *__restrict__ order, const int32_t *__restrict__ starts,
                                                                       const int32_t *__restrict__ gedges, int64_t G,
                                                                       int32_t *__restrict__ pos, int64_t *__restrict__ ge_sorted,
                                                                       int32_t *__restrict__ g_vrow0, int32_t *__restrict__ g_nv) {
  const int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (p > G) return;
  if (p == G) {
    ge_sorted[p] = 0;
    return;
  }
  const uint32_t g = order[p];
  pos[g] = (int32_t)p;
  ge_sorted[p] = gedges[g];
  g_vrow0[p] = starts[g];
  g_nv[p] = starts[g + 1] - starts[g];
}

// one 32-bit key per edge: ((position of its group) * n_slabs + source slab) * R + row in group
__global__ __launch_bounds__(kThreads) void plan_edge_keys_kernel(const int64_t *__restrict__ src, const int64_t *__restrict__ dst,
                                                                    int64_t nnz, int64_t out_rows, const int32_t *__restrict__ rowptr,
                                                                    const Tri *__restrict__ tri_ex, const int32_t *__restrict__ gidinc,
                                                                    const int32_t *__restrict__ starts, const int32_t *__restrict__ pos,
                                                                    int64_t V, uint32_t cap, int slab_shift, int64_t n_slabs, uint32_t R,
                                                                    uint32_t *__restrict__ k32, uint32_t *__restrict__ v32) {
  const int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (e >= nnz) return;
  int64_t d = dst[e];
  d = d < 0 ? 0 : (d >= out_rows ? out_rows - 1 : d);               // (checked by stage 1; clamped all the same)
  int64_t off = e - (int64_t)rowptr[d];
  off = off < 0 ? 0 : off;
  const uint32_t cnt = (uint32_t)row_count(rowptr, d);
  const uint32_t nv = cnt > cap ? (cnt + cap - 1) / cap : 1;        // (plan_vrows_kernel's split: edge j of a row goes to piece j % nv)
  int64_t vrow = (int64_t)tri_ex[d].nv + (uint32_t)off % nv;
  vrow = vrow >= V ? V - 1 : vrow;                                   // (only for an index that is not ascending after all)
  const int32_t gid = gidinc[vrow] - 1;
  const uint32_t dl = (uint32_t)(vrow - starts[gid]);
  int64_t s = src[e];
  int64_t slab = s < 0 ? 0 : (s >> slab_shift);
  slab = slab > n_slabs - 1 ? n_slabs - 1 : slab;
  k32[e] = (uint32_t)(((int64_t)pos[gid] * n_slabs + slab) * R + dl);
  v32[e] = (uint32_t)e;
}

__global__ __launch_bounds__(kThreads) void plan_edge_out_kernel(const uint32_t *__restrict__ ks, const uint32_t *__restrict__ vs,
                                                                   const int64_t *__restrict__ src, int64_t nnz, uint32_t R,
                                                                   int32_t *__restrict__ e_src, uint8_t *__restrict__ e_dl,
                                                                   int32_t *__restrict__ e_perm) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= nnz) return;
  const uint32_t v = vs[i], k = ks[i];
  e_perm[i] = (int32_t)v;
  e_dl[i] = (uint8_t)(k % R);
  e_src[i] = (int32_t)src[v];
}

// ---- scratch layouts ---------------------------------------------------------------------------------------------------
struct S1 {
  size_t sc, rowptr, tri_in, tri_ex, tmp, total;
};
struct S2 {
  size_t v_cnt, vpre, jump_a, jump_b, reach, gidinc, tmp, total;
};
struct S3 {
  size_t starts, gedges, gk0, gk1, gv0, gv1, pos, ge_sorted, k0, k1, v0, v1, tmp, total;
};

size_t scan_tmp_tri(int64_t n) {
  size_t b = 0;
  (void)rocprim::exclusive_scan(nullptr, b, (const Tri *)nullptr, (Tri *)nullptr, Tri{0, 0, 0}, (size_t)n, TriPlus(), (hipStream_t) nullptr);
  return b;
}
size_t scan_tmp_i32(int64_t n) {
  size_t a = 0, b = 0;
  (void)rocprim::exclusive_scan(nullptr, a, (const int32_t *)nullptr, (int32_t *)nullptr, 0, (size_t)n, rocprim::plus<int32_t>(), (hipStream_t) nullptr);
  (void)rocprim::inclusive_scan(nullptr, b, (const int32_t *)nullptr, (int32_t *)nullptr, (size_t)n, rocprim::plus<int32_t>(), (hipStream_t) nullptr);
  return a > b ? a : b;
}
size_t scan_tmp_i64(int64_t n) {
  size_t b = 0;
  (void)rocprim::exclusive_scan(nullptr, b, (const int64_t *)nullptr, (int64_t *)nullptr, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), (hipStream_t) nullptr);
  return b;
}
size_t sort_tmp(int64_t n) {
  size_t b = 0;
  rocprim::double_buffer<uint32_t> k(nullptr, nullptr), v(nullptr, nullptr);
  (void)rocprim::radix_sort_pairs(nullptr, b, k, v, (size_t)n, 0u, 32u, (hipStream_t) nullptr);
  return b;
}

S1 layout1(int64_t out_rows) {
  S1 L;
  size_t o = 0;
  L.sc = o;      o += up256(S_COUNT * sizeof(int64_t));
  L.rowptr = o;  o += up256((size_t)(out_rows + 1) * sizeof(int32_t));
  L.tri_in = o;  o += up256((size_t)out_rows * sizeof(Tri));
  L.tri_ex = o;  o += up256((size_t)out_rows * sizeof(Tri));
  L.tmp = o;     o += up256(scan_tmp_tri(out_rows));
  L.total = o;
  return L;
}
S2 layout2(int64_t V) {
  S2 L;
  size_t o = 0;
  const size_t a = up256((size_t)(V + 1) * sizeof(int32_t));
  L.v_cnt = o;   o += a;
  L.vpre = o;    o += a;
  L.jump_a = o;  o += a;
  L.jump_b = o;  o += a;
  L.reach = o;   o += a;
  L.gidinc = o;  o += a;
  L.tmp = o;     o += up256(scan_tmp_i32(V + 1));
  L.total = o;
  return L;
}
S3 layout3(int64_t nnz, int64_t G) {
  S3 L;
  size_t o = 0;
  const size_t g = up256((size_t)(G + 1) * sizeof(int32_t)), e = up256((size_t)nnz * sizeof(uint32_t));
  L.starts = o;     o += g;
  L.gedges = o;     o += g;
  L.gk0 = o;        o += g;
  L.gk1 = o;        o += g;
  L.gv0 = o;        o += g;
  L.gv1 = o;        o += g;
  L.pos = o;        o += g;
  L.ge_sorted = o;  o += up256((size_t)(G + 1) * sizeof(int64_t));
  L.k0 = o;         o += e;
  L.k1 = o;         o += e;
  L.v0 = o;         o += e;
  L.v1 = o;         o += e;
  size_t t = sort_tmp(nnz);
  const size_t t2 = sort_tmp(G), t3 = scan_tmp_i64(G + 1);
  t = t > t2 ? t : t2;
  t = t > t3 ? t : t3;
  L.tmp = o;        o += up256(t);
  L.total = o;
  return L;
}

int read_scalars(const int64_t *dev, int64_t *host, int n, hipStream_t st) {
  PLAN_TRY(hipMemcpyAsync(host, dev, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  PLAN_TRY(hipStreamSynchronize(st));
  return GEOT_OK;
}

bool job_ok(const geot_slab_plan_job *j) {
  return j && j->dst_index && j->src_index && j->nnz > 0 && j->nnz < ((int64_t)1 << 31) && j->out_rows > 0 && j->out_rows < ((int64_t)1 << 31) &&
         j->rows_per_group >= 1 && j->rows_per_group <= 255 && j->units >= 1 && j->rowbytes > 0;
}

} // namespace

extern "C" {

size_t geot_slab_plan_scratch_bytes(const geot_slab_plan_job *job, int stage) {
  if (!job) return 0;
  if (stage == 1) return layout1(job->out_rows > 0 ? job->out_rows : 1).total;
  if (stage == 2) return layout2(job->n_vrows > 0 ? job->n_vrows : 1).total;
  if (stage == 3) return layout3(job->nnz > 0 ? job->nnz : 1, job->n_groups > 0 ? job->n_groups : 1).total;
  return 0;
}

int geot_slab_plan_rows(geot_slab_plan_job *job, void *scratch1, size_t scratch1_bytes, void *stream) {
  if (!job_ok(job) || !scratch1) return geot_internal_fail(GEOT_EINVAL, "slab_plan_rows: bad job (1 <= nnz, out_rows < 2^31; 1 <= rows_per_group <= 255)");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const S1 L = layout1(job->out_rows);
  if (scratch1_bytes < L.total || (reinterpret_cast<uintptr_t>(scratch1) & 255)) return geot_internal_fail(GEOT_EWORKSPACE, "slab_plan_rows: scratch too small or misaligned");
  char *b = static_cast<char *>(scratch1);
  int64_t *sc = reinterpret_cast<int64_t *>(b + L.sc);
  int32_t *rowptr = reinterpret_cast<int32_t *>(b + L.rowptr);
  Tri *tri_in = reinterpret_cast<Tri *>(b + L.tri_in), *tri_ex = reinterpret_cast<Tri *>(b + L.tri_ex);
  PLAN_TRY(hipMemsetAsync(sc, 0, S_COUNT * sizeof(int64_t), st));
  hipLaunchKernelGGL(plan_rowptr_kernel, dim3(blocks_for(job->out_rows + 1)), dim3(kThreads), 0, st, job->dst_index, job->nnz, job->out_rows, rowptr, sc);
  hipLaunchKernelGGL(plan_params_kernel, dim3(1), dim3(1), 0, st, sc, job->nnz, (int64_t)job->rows_per_group, job->units);
  hipLaunchKernelGGL(plan_row_tri_kernel, dim3(blocks_for(job->out_rows)), dim3(kThreads), 0, st, rowptr, job->out_rows, sc, tri_in);
  size_t tb = L.total - L.tmp;
  PLAN_TRY(rocprim::exclusive_scan(b + L.tmp, tb, tri_in, tri_ex, Tri{0, 0, 0}, (size_t)job->out_rows, TriPlus(), st));
  hipLaunchKernelGGL(plan_row_totals_kernel, dim3(1), dim3(1), 0, st, tri_in, tri_ex, job->out_rows, sc);
  PLAN_TRY(hipGetLastError());
  int64_t h[S_COUNT];
  const int rc = read_scalars(sc, h, S_COUNT, st);
  if (rc != GEOT_OK) return rc;
  if (h[S_FIRST] < 0 || h[S_LAST] >= job->out_rows) return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_plan_rows: dst_index outside [0, out_rows)");
  job->nonempty = h[S_NONEMPTY];
  job->budget = h[S_BUDGET];
  job->cap = h[S_CAP];
  job->n_vrows = h[S_V];
  job->n_split = h[S_NSPLIT];
  job->n_carry = h[S_NCARRY];
  int shift = 0; // slabs of 2^k source rows (the kernel finds an edge's slab with a shift)
  while (((int64_t)2 << shift) * job->rowbytes <= job->slab_bytes) ++shift;
  job->slab_shift = shift;
  const int64_t slab_rows = (int64_t)1 << shift;
  int64_t n_slabs = (job->src_rows + slab_rows - 1) / slab_rows;
  job->n_slabs = n_slabs < 1 ? 1 : n_slabs;
  if (job->n_vrows < 1 || job->n_vrows >= ((int64_t)1 << 31)) return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_plan_rows: no virtual rows");
  return GEOT_OK;
}

int geot_slab_plan_groups(geot_slab_plan_job *job, const void *scratch1, void *scratch2, size_t scratch2_bytes, int64_t *v_out,
                          int32_t *v_row, int32_t *v_total, int64_t *c_row, int64_t *c_first, int32_t *c_count, int64_t *c_total,
                          void *stream) {
  if (!job_ok(job) || !scratch1 || !scratch2 || !v_out || !v_row || !v_total || !c_row || !c_first || !c_count || !c_total)
    return geot_internal_fail(GEOT_EINVAL, "slab_plan_groups: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t V = job->n_vrows;
  const S1 L1 = layout1(job->out_rows);
  const S2 L = layout2(V);
  if (scratch2_bytes < L.total || (reinterpret_cast<uintptr_t>(scratch2) & 255)) return geot_internal_fail(GEOT_EWORKSPACE, "slab_plan_groups: scratch too small or misaligned");
  const char *b1 = static_cast<const char *>(scratch1);
  char *b = static_cast<char *>(scratch2);
  const int32_t *rowptr = reinterpret_cast<const int32_t *>(b1 + L1.rowptr);
  const Tri *tri_ex = reinterpret_cast<const Tri *>(b1 + L1.tri_ex);
  int32_t *v_cnt = reinterpret_cast<int32_t *>(b + L.v_cnt), *vpre = reinterpret_cast<int32_t *>(b + L.vpre);
  int32_t *ja = reinterpret_cast<int32_t *>(b + L.jump_a), *jb = reinterpret_cast<int32_t *>(b + L.jump_b);
  int32_t *reach = reinterpret_cast<int32_t *>(b + L.reach), *gidinc = reinterpret_cast<int32_t *>(b + L.gidinc);
  PLAN_TRY(hipMemsetAsync(v_cnt + V, 0, sizeof(int32_t), st)); // (the scan below runs over V + 1 entries: vpre[V] = all edges)
  hipLaunchKernelGGL(plan_vrows_kernel, dim3(blocks_for(job->out_rows)), dim3(kThreads), 0, st, rowptr, tri_ex, job->out_rows, (int)job->cap, v_cnt,
                     v_out, v_row, v_total, c_row, c_first, c_count, c_total);
  size_t tb = L.total - L.tmp;
  PLAN_TRY(rocprim::exclusive_scan(b + L.tmp, tb, v_cnt, vpre, 0, (size_t)(V + 1), rocprim::plus<int32_t>(), st));
  hipLaunchKernelGGL(plan_next_kernel, dim3(blocks_for(V + 1)), dim3(kThreads), 0, st, v_cnt, V, (int)job->rows_per_group, job->budget, ja, reach);
  const int rounds = ceil_log2_i64(V + 1);
  for (int k = 0; k < rounds; ++k) {
    hipLaunchKernelGGL(plan_double_kernel, dim3(blocks_for(V + 1)), dim3(kThreads), 0, st, ja, jb, reach, V);
    int32_t *t = ja;
    ja = jb;
    jb = t;
  }
  tb = L.total - L.tmp;
  PLAN_TRY(rocprim::inclusive_scan(b + L.tmp, tb, reach, gidinc, (size_t)V, rocprim::plus<int32_t>(), st));
  PLAN_TRY(hipGetLastError());
  int32_t G32 = 0;
  PLAN_TRY(hipMemcpyAsync(&G32, gidinc + (V - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
  PLAN_TRY(hipStreamSynchronize(st));
  job->n_groups = G32;
  if (job->n_groups < 1) return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_plan_groups: no groups");
  if (job->n_groups > (((int64_t)1 << 32) - 1) / ((int64_t)job->n_slabs * job->rows_per_group)) return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_plan_groups: groups x slabs x rows does not fit a 32-bit sort key");
  return GEOT_OK;
}

int geot_slab_plan_edges(const geot_slab_plan_job *job, const void *scratch1, const void *scratch2, void *scratch3, size_t scratch3_bytes,
                         int64_t *g_begin, int32_t *g_vrow0, int32_t *g_nv, int32_t *e_src, uint8_t *e_dl, int32_t *e_perm, void *stream) {
  if (!job_ok(job) || !scratch1 || !scratch2 || !scratch3 || !g_begin || !g_vrow0 || !g_nv || !e_src || !e_dl || !e_perm)
    return geot_internal_fail(GEOT_EINVAL, "slab_plan_edges: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t V = job->n_vrows, G = job->n_groups, nnz = job->nnz;
  const S1 L1 = layout1(job->out_rows);
  const S2 L2 = layout2(V);
  const S3 L = layout3(nnz, G);
  if (scratch3_bytes < L.total || (reinterpret_cast<uintptr_t>(scratch3) & 255)) return geot_internal_fail(GEOT_EWORKSPACE, "slab_plan_edges: scratch too small or misaligned");
  const char *b1 = static_cast<const char *>(scratch1), *b2 = static_cast<const char *>(scratch2);
  char *b = static_cast<char *>(scratch3);
  const int32_t *rowptr = reinterpret_cast<const int32_t *>(b1 + L1.rowptr);
  const Tri *tri_ex = reinterpret_cast<const Tri *>(b1 + L1.tri_ex);
  const int32_t *vpre = reinterpret_cast<const int32_t *>(b2 + L2.vpre), *reach = reinterpret_cast<const int32_t *>(b2 + L2.reach);
  const int32_t *gidinc = reinterpret_cast<const int32_t *>(b2 + L2.gidinc);
  int32_t *starts = reinterpret_cast<int32_t *>(b + L.starts), *gedges = reinterpret_cast<int32_t *>(b + L.gedges), *pos = reinterpret_cast<int32_t *>(b + L.pos);
  uint32_t *gk0 = reinterpret_cast<uint32_t *>(b + L.gk0), *gk1 = reinterpret_cast<uint32_t *>(b + L.gk1);
  uint32_t *gv0 = reinterpret_cast<uint32_t *>(b + L.gv0), *gv1 = reinterpret_cast<uint32_t *>(b + L.gv1);
  int64_t *ge_sorted = reinterpret_cast<int64_t *>(b + L.ge_sorted);
  uint32_t *k0 = reinterpret_cast<uint32_t *>(b + L.k0), *k1 = reinterpret_cast<uint32_t *>(b + L.k1);
  uint32_t *v0 = reinterpret_cast<uint32_t *>(b + L.v0), *v1 = reinterpret_cast<uint32_t *>(b + L.v1);
  void *tmp = b + L.tmp;
  const size_t tmp_bytes = L.total - L.tmp;

  hipLaunchKernelGGL(plan_starts_kernel, dim3(blocks_for(V)), dim3(kThreads), 0, st, reach, gidinc, V, G, starts);
  hipLaunchKernelGGL(plan_group_edges_kernel, dim3(blocks_for(G)), dim3(kThreads), 0, st, starts, vpre, G, job->budget, gedges, gk0, gv0);
  {
    rocprim::double_buffer<uint32_t> keys(gk0, gk1), vals(gv0, gv1);
    size_t tb = tmp_bytes;
    PLAN_TRY(rocprim::radix_sort_pairs(tmp, tb, keys, vals, (size_t)G, 0u, (unsigned)bit_width_u64((uint64_t)job->budget), st));
    hipLaunchKernelGGL(plan_group_tables_kernel, dim3(blocks_for(G + 1)), dim3(kThreads), 0, st, vals.current(), starts, gedges, G, pos, ge_sorted,
                       g_vrow0, g_nv);
  }
  {
    size_t tb = tmp_bytes;
    PLAN_TRY(rocprim::exclusive_scan(tmp, tb, ge_sorted, g_begin, (int64_t)0, (size_t)(G + 1), rocprim::plus<int64_t>(), st));
  }
  hipLaunchKernelGGL(plan_edge_keys_kernel, dim3(blocks_for(nnz)), dim3(kThreads), 0, st, job->src_index, job->dst_index, nnz, job->out_rows, rowptr,
                     tri_ex, gidinc, starts, pos, V, (uint32_t)job->cap, (int)job->slab_shift, (int64_t)job->n_slabs, (uint32_t)job->rows_per_group, k0, v0);
  {
    const uint64_t kmax = (uint64_t)G * (uint64_t)job->n_slabs * (uint64_t)job->rows_per_group - 1;
    rocprim::double_buffer<uint32_t> keys(k0, k1), vals(v0, v1);
    size_t tb = tmp_bytes;
    PLAN_TRY(rocprim::radix_sort_pairs(tmp, tb, keys, vals, (size_t)nnz, 0u, (unsigned)bit_width_u64(kmax), st));
    hipLaunchKernelGGL(plan_edge_out_kernel, dim3(blocks_for(nnz)), dim3(kThreads), 0, st, keys.current(), vals.current(), job->src_index, nnz,
                       (uint32_t)job->rows_per_group, e_src, e_dl, e_perm);
  }
  PLAN_TRY(hipGetLastError());
  return GEOT_OK;
}

} // extern "C"
