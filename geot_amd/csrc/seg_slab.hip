// seg_slab.hip -- source-blocked ("slab") form of the gather modes for DENSE graphs (Reddit-like: hundreds of
// edges per row, a source table of a few hundred MB), behind geot_slab_spmm (include/geot_hip.h).
//
// Why.  seg_tile_kernel gathers one src row per edge wherever it lives: at BASELINE.json configs[3] (232 965 nodes,
// 114.6 M edges, H*F = 256 -> 1-KiB rows) that is 117 GB of row gathers against 4.1 GB of compulsory traffic, served
// at the Infinity-Cache rate (7.3 TB/s, 16-17 ms).  An XCD's 4 MiB L2 serves the same row reads ~4x faster
// (profiles/r02/kexp2_slab_cfg4_table.txt: 27 TB/s when every wave of the chip gathers from the same ~1 MiB slab of
// the table at the same time, against 7.5 TB/s in dst order).  This kernel creates that situation:
//
//   * Phase A (host layer, once per edge list, cached): the dst rows are cut into GROUPS of <= R consecutive
//     (virtual) rows with about the same number of edges; hub rows are split into virtual rows of <= CAP edges.
//     Inside a group the edges are sorted by (source slab, row in group).  Groups are ordered by size so that
//     the groups running at the same time are equally long.
//   * Phase B (this kernel, persistent: 2 workgroups per CU, all resident): every lane group ("unit": the rowbytes/16
//     lanes that own one row) takes one group per ROUND, keeps the group's <= R output rows in LDS, and walks its
//     edges slab by slab.  All units start together and do the same amount of work per slab, so without any
//     synchronisation the whole chip sweeps the source table in step: each slab is fetched from HBM / Infinity
//     Cache once per XCD and round, every further read hits in L2.  Runs of equal (slab, row) are summed in
//     registers and added to the LDS row with ds_add_f32 (no return, no stall; one unit owns its rows: the order
//     of additions is fixed -> deterministic).  At the end of the round the rows are stored to dst with plain
//     coalesced stores; virtual rows of a split hub go to a carry buffer.
//   * Phase C (seg_slab_combine_kernel): the few split rows are summed from their carry slots in order.
//
// No global atomics, every dst row written once (dst is zero-filled first for the rows without edges).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "geot_hip.h"
#include "internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kU = 8; // row loads in flight per lane

struct SlabParams {
  geot_slab_plan plan;
  const void *weight;
  const void *src;
  void *dst;
  float *carry;
  int64_t src_rows, K, F;
  int H, Fh;
  uint32_t rowbytes;
  int lpr_log2;
  int rounds;
};

typedef float f4_t __attribute__((ext_vector_type(4)));

// WMODE: 0 none, 1 weight[e], 2 weight[e*H + h], 3 weight[h*nnz + e]
template <int WMODE>
__global__ __launch_bounds__(kThreads) void seg_slab_kernel(SlabParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const geot_slab_plan &P = p.plan;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lpr = 1 << p.lpr_log2;
  const int G = 64 >> p.lpr_log2;           // units per wave
  const int sub = lane >> p.lpr_log2, c = lane & (lpr - 1);
  const int R = P.rows_per_group;
  const int hw = WMODE == 0 ? 0 : (WMODE == 1 ? 1 : p.H);
  // LDS: accumulators [4 waves][G units][R rows][lpr lanes] float4 = 4 * R KiB, then the staged weights
  f4_t *accL = reinterpret_cast<f4_t *>(smem) + ((size_t)(wave * G + sub) * R) * lpr;
  float *wL = reinterpret_cast<float *>(smem + (size_t)4 * R * 1024) + (size_t)(wave * G + sub) * lpr * (hw > 0 ? hw : 1);
  const int64_t unit = ((int64_t)blockIdx.x * 4 + wave) * G + sub;
  const int64_t units = P.units;
  const char *src = static_cast<const char *>(p.src);
  const float *weight = static_cast<const float *>(p.weight);
  float *dst = static_cast<float *>(p.dst);
  const int h = WMODE >= 2 ? (c * 4) / p.Fh : 0;
  const uint32_t rb = p.rowbytes;

  for (int r = 0; r < p.rounds; ++r) {
    // serpentine over the size-sorted groups: no unit is always handed the larger group of its round
    const int64_t pos = (int64_t)r * units + ((r & 1) ? units - 1 - unit : unit);
    const bool has = pos < P.n_groups && unit < units;
    const int64_t e0 = has ? P.g_begin[pos] : 0;
    const int len = has ? (int)(P.g_begin[pos + 1] - e0) : 0;
    const int nv = has ? P.g_nv[pos] : 0;
    for (int l = 0; l < R; ++l) accL[(size_t)l * lpr + c] = f4_t{0.f, 0.f, 0.f, 0.f};
    int maxlen = len;
    for (int o = 32; o >= lpr && o > 0; o >>= 1) {   // max over the wave's units (wave-uniform loop bound)
      const int other = __shfl_xor(maxlen, o, 64);
      maxlen = other > maxlen ? other : maxlen;
    }
    f4_t acc = {0.f, 0.f, 0.f, 0.f};
    int cur = 255;                                   // no open row
    for (int off = 0; off < maxlen; off += lpr) {
      const bool valid = off + c < len;
      const int64_t ei = e0 + off + c;
      const int my_src = valid ? P.e_src[ei] : 0;
      const int my_dl = valid ? (int)P.e_dl[ei] : 255;
      if constexpr (WMODE != 0) {
        const int64_t pe = valid ? (int64_t)P.e_perm[ei] : 0;
        if constexpr (WMODE == 1) wL[c] = valid ? weight[pe] : 0.f;
        if constexpr (WMODE == 2) {
          if (p.H == 4) {
            const f4_t t = valid ? *reinterpret_cast<const f4_t *>(weight + pe * 4) : f4_t{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f4_t *>(wL + c * 4) = t;
          } else {
            for (int q = 0; q < p.H; ++q) wL[c * hw + q] = valid ? weight[pe * p.H + q] : 0.f;
          }
        }
        if constexpr (WMODE == 3) {
          for (int q = 0; q < p.H; ++q) wL[c * hw + q] = valid ? weight[(int64_t)q * P.nnz + pe] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();             // LDS is in order per wave: the reads below see these writes
      }
      int n_here = len - off;                        // edges of this unit in the chunk (<= 0: none)
      int n_max = maxlen - off;
      n_max = n_max < lpr ? n_max : lpr;
      for (int b = 0; b < n_max; b += kU) {
        f4_t v[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          int row = __shfl(my_src, b + u, lpr);      // (b+u) < lpr always: lpr >= 16 >= kU and b + kU <= lpr
          if (b + u >= n_here || (unsigned)row >= (unsigned)p.src_rows) row = 0;
          v[u] = *reinterpret_cast<const f4_t *>(src + (int64_t)row * rb + c * 16);
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int dl = b + u < n_here ? __shfl(my_dl, b + u, lpr) : 255;
          if (dl != cur) {
            if (cur != 255) {
              float *a = reinterpret_cast<float *>(accL + (size_t)cur * lpr + c);
#pragma unroll
              for (int i = 0; i < 4; ++i) __hip_atomic_fetch_add(a + i, acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            cur = dl;
            acc = f4_t{0.f, 0.f, 0.f, 0.f};
          }
          if constexpr (WMODE == 0) acc += v[u];
          else {
            const float w = b + u < n_here ? wL[(b + u) * hw + h] : 0.f;
            acc += v[u] * w;
          }
        }
      }
      if constexpr (WMODE != 0) __builtin_amdgcn_wave_barrier(); // weights of this chunk are consumed
    }
    if (cur != 255) {
      float *a = reinterpret_cast<float *>(accL + (size_t)cur * lpr + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) __hip_atomic_fetch_add(a + i, acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // the group's rows: dst for whole rows, the carry buffer for the pieces of a split hub
    if (has) {
      const int64_t v0 = P.g_vrow0[pos];
      for (int l = 0; l < nv; ++l) {
        const int64_t t = P.v_out[v0 + l];
        const f4_t row = accL[(size_t)l * lpr + c];
        if (t >= 0) {
          if (t < p.K) *reinterpret_cast<f4_t *>(dst + t * p.F + c * 4) = row;
        } else {
          *reinterpret_cast<f4_t *>(p.carry + (-t - 1) * p.F + c * 4) = row;
        }
      }
    }
  }
}

// split hubs: dst[row] = sum of its carry slots, in slot order (one lane group per split row)
__global__ __launch_bounds__(kThreads) void seg_slab_combine_kernel(SlabParams p) {
  const geot_slab_plan &P = p.plan;
  const int lpr = 1 << p.lpr_log2;
  const int g = threadIdx.x >> p.lpr_log2, c = threadIdx.x & (lpr - 1);
  const int ng = kThreads >> p.lpr_log2;
  for (int64_t s = (int64_t)blockIdx.x * ng + g; s < P.n_split; s += (int64_t)gridDim.x * ng) {
    const int64_t row = P.c_row[s];
    const int64_t first = P.c_first[s];
    const int n = P.c_count[s];
    f4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < n; ++i) acc += *reinterpret_cast<const f4_t *>(p.carry + (first + i) * p.F + c * 4);
    if (row >= 0 && row < p.K) *reinterpret_cast<f4_t *>(static_cast<float *>(p.dst) + row * p.F + c * 4) = acc;
  }
}

} // namespace

extern "C" {

int geot_slab_units(void) { return 256 * 2 * 4; } // waves of the persistent grid: 256 CUs x 2 workgroups x 4

static size_t slab_lds_bytes(int rows_per_group, int weight_mode, int64_t heads) {
  const size_t hw = weight_mode <= 1 ? 1 : (size_t)heads;
  return (size_t)4 * rows_per_group * 1024 + (size_t)4 * 64 * hw * sizeof(float);
}

// two workgroups per CU inside the classic 64 KB per workgroup: R KiB of accumulators per wave + the staged weights
int geot_slab_rows_per_group(int weight_mode, int64_t heads) {
  int r = 16;
  while (r > 1 && slab_lds_bytes(r, weight_mode, heads) > 64 * 1024) --r;
  return r;
}

size_t geot_slab_workspace_bytes(const geot_slab_plan *plan, int64_t feat_total) {
  if (!plan) return 0;
  return (size_t)(plan->n_carry > 0 ? plan->n_carry : 1) * (size_t)feat_total * sizeof(float) + 256;
}

int geot_slab_spmm(const geot_slab_plan *plan, const void *weight, int weight_mode, const void *src, void *dst,
                   int64_t heads, int64_t feat, int64_t src_rows, int64_t out_rows, int dtype, void *workspace,
                   size_t workspace_bytes, void *stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!plan || !src || !dst) return geot_internal_fail(GEOT_EINVAL, "slab_spmm: null pointer");
  if (dtype != GEOT_F32) return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_spmm: float32 only");
  if (heads < 1 || feat < 1 || out_rows < 0) return geot_internal_fail(GEOT_EINVAL, "slab_spmm: bad sizes");
  if (weight_mode < 0 || weight_mode > 3 || (weight_mode != 0 && !weight))
    return geot_internal_fail(GEOT_EINVAL, "slab_spmm: weight_mode 0..3 (and a weight pointer for 1..3)");
  const int64_t F = heads * feat;
  const int64_t rowbytes = F * 4;
  int lpr_log2 = -1;
  for (int l = 4; l <= 6; ++l)
    if (rowbytes == ((int64_t)16 << l)) lpr_log2 = l;
  if (lpr_log2 < 0) return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_spmm: rows of 256, 512 or 1024 bytes only");
  if ((4 * feat) % 16 != 0 && weight_mode >= 2) return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_spmm: feat per head must be a multiple of 4");
  if ((((uintptr_t)src) | ((uintptr_t)dst) | ((uintptr_t)workspace)) & 15) return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_spmm: 16-byte aligned operands");
  if (plan->units != geot_slab_units() * (64 >> lpr_log2)) return geot_internal_fail(GEOT_EINVAL, "slab_spmm: plan was built for a different unit count");
  if (plan->rows_per_group < 1 || plan->rows_per_group > 32) return geot_internal_fail(GEOT_EINVAL, "slab_spmm: rows_per_group 1..32");
  if (heads > 16) return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_spmm: at most 16 heads");
  const size_t need = geot_slab_workspace_bytes(plan, F);
  if (!workspace || workspace_bytes < need) return geot_internal_fail(GEOT_EWORKSPACE, "slab_spmm: workspace too small");
  if (out_rows == 0) return GEOT_OK;

  SlabParams p;
  p.plan = *plan;
  p.weight = weight;
  p.src = src;
  p.dst = dst;
  p.carry = reinterpret_cast<float *>(static_cast<char *>(workspace) + 256); // (the first 256 bytes are the tile kernels' control words)
  p.src_rows = src_rows;
  p.K = out_rows;
  p.F = F;
  p.H = (int)heads;
  p.Fh = (int)feat;
  p.rowbytes = (uint32_t)rowbytes;
  p.lpr_log2 = lpr_log2;
  p.rounds = (int)((plan->n_groups + plan->units - 1) / plan->units);

  hipError_t e = hipMemsetAsync(dst, 0, (size_t)out_rows * (size_t)rowbytes, st); // rows without edges
  if (e != hipSuccess) return geot_internal_fail(GEOT_ELAUNCH, hipGetErrorString(e));
  if (plan->n_groups > 0) {
    const size_t lds = slab_lds_bytes(plan->rows_per_group, weight_mode, heads);
    if (lds > 64 * 1024) return geot_internal_fail(GEOT_EINVAL, "slab_spmm: rows_per_group exceeds the LDS budget (geot_slab_rows_per_group)");
    const dim3 grid(256 * 2), blk(kThreads);
    switch (weight_mode) {
    case 0: hipLaunchKernelGGL(seg_slab_kernel<0>, grid, blk, lds, st, p); break;
    case 1: hipLaunchKernelGGL(seg_slab_kernel<1>, grid, blk, lds, st, p); break;
    case 2: hipLaunchKernelGGL(seg_slab_kernel<2>, grid, blk, lds, st, p); break;
    default: hipLaunchKernelGGL(seg_slab_kernel<3>, grid, blk, lds, st, p); break;
    }
    e = hipGetLastError();
    if (e != hipSuccess) return geot_internal_fail(GEOT_ELAUNCH, hipGetErrorString(e));
    if (plan->n_split > 0) {
      int64_t blocks = (plan->n_split + (kThreads >> lpr_log2) - 1) / (kThreads >> lpr_log2);
      if (blocks > 1024) blocks = 1024;
      hipLaunchKernelGGL(seg_slab_combine_kernel, dim3((unsigned)blocks), blk, 0, st, p);
      e = hipGetLastError();
      if (e != hipSuccess) return geot_internal_fail(GEOT_ELAUNCH, hipGetErrorString(e));
    }
  }
  return GEOT_OK;
}

} // extern "C"
