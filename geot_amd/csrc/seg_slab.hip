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
//     registers; when the row changes the open row is written back to LDS and the next one read (one unit owns
//     its rows: the order of additions is fixed -> deterministic).  At the end of the round the rows are stored to dst with plain
//     coalesced stores; virtual rows of a split hub go to a carry buffer.
//   * Phase C (seg_slab_combine_kernel): the few split rows are summed from their carry slots in order.
//
// No global atomics, every dst row written once (dst is zero-filled first for the rows without edges).
//
// Kernels: seg_slab_kernel (all weight modes and reductions; rows of 1 KiB run one row per wave-instruction on scalar bases - group
// bounds, row numbers, row switches in SGPRs -, 128-byte rows as lane groups of 8 lanes), seg_slab_wrow_kernel (every weight mode and multi-head
// weights on rows of 512 / 256 bytes: the scalar path at 8 / 4 bytes per lane, a unit = a wave), seg_slab_combine_kernel,
// seg_slab_sddmm_kernel / seg_slab_sddmm_wrow_kernel (d/dweight and attention scores over the same plan; results staged in plan order)
// + slab_unstage_kernel (into edge order, a group at a time through LDS), seg_slab_sddmm_mfma_kernel (the 16-bit multi-head SDDMM of
// 512-byte rows on the matrix cores), slab_stage_weights*_kernel (per-call weights into plan order), seg_slab_wpair_kernel (an
// experiment behind "slab_pair").  The waves of an XCD keep loose step through SlabStep.  Row gathers of the wave-row forms are
// buffer_loads with scalar row offsets (slab_row_load); "slab_probe" drops them to time a kernel without its gathers.
// Measurements and what bounds them (instruction issue, not the gathers): DESIGN.md section 3.1d.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <utility>

#include "geot_hip.h"
#include "internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kU = 8; // row loads in flight per lane (16: measured 4-35 % slower, r04)

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
  // loose per-XCD lockstep (see SlabStep): progress words of the waves of each XCD, slab size as a shift
  int *prog;            // [8][kProgSlots] current step of every registered wave (0x7f7f7f7f = not running)
  int *prog_cnt;        // [8] slot allocator (starts at 0x7f7f7f7f)
  int slab_shift;       // slab = source row >> slab_shift
  int n_slabs;
  int window;           // a wave may be at most `window` slabs ahead of the slowest wave of its XCD; < 0: no sync
  int far;              // ... but a slowest wave MORE than `far` steps behind is not waited for: nothing it reads can still be shared
  int nt_plan;          // experiment: non-temporal loads of the plan's edge fields and of the weights (read once per launch)
  int w_in_plan_order;  // WMODE 1 / 2: weight[] is indexed by plan position (a static weight permuted once, scores computed in plan order), not by edge id
  int probe;            // timing experiment ("slab_probe"): the gathered table's buffer descriptor has ZERO records - every row read of the
                        // wave-row forms is dropped by the range check (returns 0, no memory traffic), the instruction stream stays
  const int *gate;      // != NULL: the launch does its work only if (*gate != 0) == gate_want (the matrix-core SpMM and its vector-ALU
  int gate_want;        // twin are BOTH enqueued; a word written by slab_nonfinite_kernel on the stream says which one runs)
};
constexpr int kProgSlots = 512;
constexpr int kProgIdle = 0x7f7f7f7f;
constexpr int kSyncTries = 640;  // polls of one wait (8 loads + s_sleep 16 each, ~1.5 us: ~1 ms; a slab step is ~50-100 us)
constexpr int kSyncGiveUp = 4;   // waits that ran out in a row before a wave stops keeping step (see SlabStep)

typedef float f4_t __attribute__((ext_vector_type(4)));

// Storage types: float, and the 16-bit types with fp32 accumulation (as the per-edge kernels and the reference's CPU path,
// csrc/cpu/index_scatter_cpu.cpp:78-86,114-116: one rounding, at the very end).  A lane always moves 16 bytes of a row:
// 4 floats or 8 halves, i.e. NV = 1 or 2 float4 accumulators per lane.
typedef _Float16 half_t;
typedef __bf16 bf16_t;
template <typename T> constexpr const char *slab_tname() { // (as rocprofv3 spells the template argument)
  return sizeof(T) == 4 ? "float" : (__is_same(T, _Float16) ? "_Float16" : "__bf16");
}
template <typename T> struct SlabVec { static constexpr int VEC = 16 / (int)sizeof(T), NV = VEC / 4; };

template <typename T> __device__ __forceinline__ void slab_unpack(const f4_t &raw, f4_t (&m)[SlabVec<T>::NV]) {
  if constexpr (sizeof(T) == 4) m[0] = raw;
  else {
    typedef T t8 __attribute__((ext_vector_type(8)));
    const t8 x = __builtin_bit_cast(t8, raw);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      m[0][i] = (float)x[i];
      m[1][i] = (float)x[4 + i];
    }
  }
}
template <typename T> __device__ __forceinline__ f4_t slab_pack(const f4_t (&m)[SlabVec<T>::NV]) {
  if constexpr (sizeof(T) == 4) return m[0];
  else {
    typedef T t8 __attribute__((ext_vector_type(8)));
    t8 x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      x[i] = (T)m[0][i];
      x[4 + i] = (T)m[1][i];
    }
    return __builtin_bit_cast(f4_t, x);
  }
}

// reductions over the messages of a row (codes of include/geot_hip.h, csrc/reducetype.h:3); mean = sum, divided by the
// row's edge count when the row is written; min / max propagate NaN like ATen (csrc/cpu/index_scatter_cpu.cpp:124-134)
template <int RED> __device__ __forceinline__ float slab_ident() {
  if constexpr (RED == GEOT_REDUCE_MAX) return -INFINITY;
  else if constexpr (RED == GEOT_REDUCE_MIN) return INFINITY;
  else return 0.f;
}
template <int RED> __device__ __forceinline__ float slab_op(float x, float y) {
  if constexpr (RED == GEOT_REDUCE_MAX) return (y != y) ? y : (x < y ? y : x);
  else if constexpr (RED == GEOT_REDUCE_MIN) return (y != y) ? y : (y < x ? y : x);
  else return x + y;
}
template <int RED> __device__ __forceinline__ void slab_acc(f4_t &acc, const f4_t &m) {
  if constexpr (RED == GEOT_REDUCE_SUM || RED == GEOT_REDUCE_MEAN) acc += m;
  else {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = slab_op<RED>(acc[i], m[i]);
  }
}

// A unit's edges are read in the PLAN's order, their weights live in the caller's EDGE order: weight[e_perm[i]].  Read that way inside
// the row loop, every 4-byte weight drags a 64-byte line through the fabric each time the sweep comes back to its neighbourhood
// (a group's edges are a contiguous range of edge ids, its ~10 k weights a 40 KB block touched over the whole life of the group:
// ~15 GB of lines for 1.8 GB of weights at configs[3], and 1.1 ms of a 4.5 ms gws call at F = 128).  Given room in the workspace
// (geot_slab_workspace_bytes_staged) the call brings the weights into plan order FIRST - this kernel: e_perm streamed, the weights
// gathered while their group's block is hot in L2 (consecutive plan positions belong to one group), written out coalesced - and the
// row loop streams them and never reads e_perm.  Inside the persistent kernel (every unit its own group's weights at the start of
// its round) the same idea LOST: 4 edges per pass +0.3-1.0 ms, 16 per pass cost the row loop its registers (profiles/r05/
// slab_cases__lane_groups__staging_*_in_kernel.txt).
template <typename V>
__global__ __launch_bounds__(kThreads) void slab_stage_weights_kernel(const int32_t *__restrict__ e_perm, const V *__restrict__ weight,
                                                                      V *__restrict__ wst, int64_t nnz) {
  constexpr int kS = 8;
  const int64_t per_block = (int64_t)kThreads * kS;
  for (int64_t base = (int64_t)blockIdx.x * per_block; base < nnz; base += (int64_t)gridDim.x * per_block) {
    int pe[kS];
#pragma unroll
    for (int k = 0; k < kS; ++k) {
      const int64_t i = base + (int64_t)k * kThreads + threadIdx.x;
      pe[k] = i < nnz ? __builtin_nontemporal_load(e_perm + i) : -1;
    }
    V v[kS];
#pragma unroll
    for (int k = 0; k < kS; ++k)
      if (pe[k] >= 0) v[k] = weight[pe[k]];
#pragma unroll
    for (int k = 0; k < kS; ++k)
      if (pe[k] >= 0) __builtin_nontemporal_store(v[k], wst + base + (int64_t)k * kThreads + threadIdx.x);
  }
}

// The same for 2- and 4-byte weights, four consecutive plan positions per lane: e_perm arrives as one 16-byte load, the four gathered
// weights leave as one 8- / 16-byte store - a third of the instructions per element (the pass is issue-bound like the row loops:
// 0.59 ms for 1.4 GB at Reddit scale before).  e_perm and wst 16- / (4 x sizeof V)-byte aligned (the launcher checks wst).
template <typename V>
__global__ __launch_bounds__(kThreads) void slab_stage_weights_x4_kernel(const int32_t *__restrict__ e_perm, const V *__restrict__ weight,
                                                                         V *__restrict__ wst, int64_t nnz) {
  typedef int i4_t __attribute__((ext_vector_type(4)));
  typedef V v4_t __attribute__((ext_vector_type(4)));
  constexpr int kS = 2;
  const int64_t quads = nnz >> 2;
  const int64_t per_block = (int64_t)kThreads * kS;
  for (int64_t base = (int64_t)blockIdx.x * per_block; base < quads; base += (int64_t)gridDim.x * per_block) {
    i4_t pe[kS];
    bool ok[kS];
#pragma unroll
    for (int k = 0; k < kS; ++k) {
      const int64_t q = base + (int64_t)k * kThreads + threadIdx.x;
      ok[k] = q < quads;
      pe[k] = ok[k] ? __builtin_nontemporal_load(reinterpret_cast<const i4_t *>(e_perm) + q) : i4_t{0, 0, 0, 0};
    }
    v4_t v[kS];
#pragma unroll
    for (int k = 0; k < kS; ++k) {
      v[k][0] = weight[pe[k][0]];          // (lanes past the end re-read element 0: no branch around the loads)
      v[k][1] = weight[pe[k][1]];
      v[k][2] = weight[pe[k][2]];
      v[k][3] = weight[pe[k][3]];
    }
#pragma unroll
    for (int k = 0; k < kS; ++k)
      if (ok[k]) __builtin_nontemporal_store(v[k], reinterpret_cast<v4_t *>(wst) + base + (int64_t)k * kThreads + threadIdx.x);
  }
  if (blockIdx.x == 0 && (int64_t)threadIdx.x < (nnz & 3)) {
    const int64_t i = (quads << 2) + threadIdx.x;
    wst[i] = weight[e_perm[i]];
  }
}

// (A form of this pre-pass that takes a group's weight block through LDS - coalesced in, picked in plan order, coalesced out, the
// mirror image of slab_unstage_kernel - measured SLOWER than this plain gather: gws F=128 fp32 4.52 vs 4.00 ms per call,
// profiles/r05/slab_cases__weight_prepass_through_lds.txt; three barriers per group cost more than the L2 requests they save.)

// ---- loose lockstep inside an XCD (used by every source-blocked kernel) ---------------------------------------------------------
// The natural lockstep (same start, same work) drifts like a random walk: with ~6000 edges per group the waves of an XCD spread
// over ~+-3 % of the table (+-6 MB at 238 MB) - more than the 4 MiB L2.  So every wave publishes the step (round * slabs + slab) it
// is working on in a per-XCD array and does not START a slab more than `window` slabs ahead of the slowest registered wave of its
// XCD.  Only leaders ever wait, the slowest wave never does, a wave that is not resident is not registered, and every wait is
// bounded: no deadlock by construction.  Same-XCD visibility: plain stores go through to the XCD's L2, the polls bypass L1 (nt loads).
// Bounded in AGGREGATE too: a wait is ~1 ms at most, and a wave whose waits run out kSyncGiveUp times in a row stops keeping step for
// the rest of the launch (it withdraws its progress word, so nobody waits for it either).  That is the situation of a grid that does
// not have the chip to itself - another persistent grid of this process, of another process, or a replayed graph on a second stream
// holds the CUs its slowest waves would run on: lockstep has nothing to offer there, and without this rule thousands of steps x a
// timed-out wait each would look like a hang.  Timing only: the result never depends on the lockstep.  All members are wave-uniform.
struct SlabStep {
  int my_slot = -1;
  int *xprog = nullptr;
  int published = -1;
  int known_min = -1;                          // a lower bound of the slowest wave's step (steps only grow)
  int timeouts = 0;                            // consecutive waits that ran out

  __device__ __forceinline__ void enter(const SlabParams &p, int lane) {
    if (p.window < 0) return;
    const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;        // HW_REG_XCC_ID[3:0]
    int slot = 0;
    if (lane == 0) slot = atomicAdd(p.prog_cnt + xcc, 1) - kProgIdle;
    slot = __builtin_amdgcn_readfirstlane(slot);
    xprog = p.prog + xcc * kProgSlots;
    if (slot >= 0 && slot < kProgSlots) my_slot = slot;
  }
  __device__ __forceinline__ void at(const SlabParams &p, int lane, int step) {   // about to start `step`
    if (my_slot < 0 || step <= published) return;
    published = step;
    if (lane == 0) __builtin_nontemporal_store(step, xprog + my_slot);
    if (known_min + p.window >= step) return;  // the last poll already allows this step: no memory round trip
    bool ok = false;
    for (int tries = 0; tries < kSyncTries; ++tries) {
      int m = kProgIdle;
#pragma unroll
      for (int q = 0; q < kProgSlots / 64; ++q) {
        const int v = __builtin_nontemporal_load(xprog + q * 64 + lane);
        m = v < m ? v : m;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_xor(m, o, 64);
        m = other < m ? other : m;
      }
      known_min = m;
      if (m + p.window >= step || step - m > p.far) {
        ok = true;
        break;
      }
      __builtin_amdgcn_s_sleep(16);
    }
    if (ok) timeouts = 0;
    else if (++timeouts >= kSyncGiveUp) {
      if (lane == 0) __builtin_nontemporal_store(kProgIdle, xprog + my_slot);
      my_slot = -1;
    }
  }
  __device__ __forceinline__ void round_done(const SlabParams &p, int lane, int r) {   // never hold the others back
    if (p.window >= 0 && my_slot >= 0) {
      published = (r + 1) * p.n_slabs;
      if (lane == 0) __builtin_nontemporal_store(published, xprog + my_slot);
    }
  }
  __device__ __forceinline__ void leave(int lane) {
    if (my_slot >= 0 && lane == 0) __builtin_nontemporal_store(kProgIdle, xprog + my_slot);
  }
};

// A gathered row read of the wave-row forms: buffer_load with the table as the buffer (base + size in four SGPRs, made once per wave
// from kernel arguments), the lane's byte offset inside the row in voffset (loop-invariant) and the ROW's byte offset - wave-uniform,
// out of v_readlane - in soffset: no vector instruction per edge for the address (global_load wanted a 64-bit add per edge).  The
// table is below 4 GiB (geot_slab_spmm / geot_slab_sddmm refuse more): offsets and the record count fit 32 bits.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t slab_table_rsrc(const void *table, int64_t rows, int row_shift, int probe = 0) {
#ifndef GEOT_DEV_EXPERIMENTS
  probe = 0;                                  // (the timing probe that drops every row read exists in the development build only)
#endif
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(table), 0, (probe & 1) ? 0 : (int)(uint32_t)((uint64_t)rows << row_shift), 0x00020000);
}
template <typename RAW> __device__ __forceinline__ RAW slab_row_load(__amdgpu_buffer_rsrc_t table, uint32_t lane_bytes, uint32_t row_bytes_off) {
  if constexpr (sizeof(RAW) == 4) return __builtin_bit_cast(RAW, __builtin_amdgcn_raw_buffer_load_b32(table, lane_bytes, row_bytes_off, 0));
  else if constexpr (sizeof(RAW) == 8) return __builtin_bit_cast(RAW, __builtin_amdgcn_raw_buffer_load_b64(table, lane_bytes, row_bytes_off, 0));
  else return __builtin_bit_cast(RAW, __builtin_amdgcn_raw_buffer_load_b128(table, lane_bytes, row_bytes_off, 0));
}

// The word that picks one kernel of a gated pair (SlabParams::gate) is written by kernels EARLIER ON THE STREAM (a memset, then
// slab_nonfinite_kernel's atomicOr) and read by every wave of the pair, on all eight XCDs.  An XCD's L2 is not coherent with the
// others': a plain (or non-temporal) load may be served a line its own L2 kept from the LAST launch - between eager launches the
// runtime's kernel-boundary invalidate hides that, between the kernel nodes of a replayed HIP graph it does not (seen: replay n + 1 of
// a captured call read replay n's word on some XCDs, so those waves of BOTH kernels took the wrong turn).  An agent-scope atomic load
// (sc1: served at the memory side) is coherent by construction.
__device__ __forceinline__ int slab_gate_word(const int *gate) { return __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// WMODE: 0 none, 1 weight[e], 2 weight[e*H + h], 3 weight[h*nnz + e]
// WAVE_ROW: rows of 1 KiB - the whole wave is one unit, edge fields are read with v_readlane (scalar row bases)
//
// Per chunk of `lpr` edges of a unit (one edge per lane of the unit: source row, row in group, original edge id):
//   * the NEXT chunk's fields are loaded at the top, its weights (which need the edge ids) after the first batch
//     of rows, and staged into the other half of a double-buffered LDS array at the end - nothing the row loop
//     needs is ever waited for;
//   * rows are gathered kU at a time; the fields of the batch (row number, row in group, weight) are fetched
//     BEFORE the loads are issued; the accumulate loop's only LDS traffic is one 16-byte write + read per lane when
//     the row changes (the open row is kept in registers).
template <typename T, int WMODE, bool WAVE_ROW, int RED>
__global__ __launch_bounds__(kThreads) void seg_slab_kernel(SlabParams p) {
  constexpr float kIdent = RED == GEOT_REDUCE_MAX ? -INFINITY : (RED == GEOT_REDUCE_MIN ? INFINITY : 0.f);
  constexpr int VEC = SlabVec<T>::VEC, NV = SlabVec<T>::NV;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (p.gate && ((slab_gate_word(p.gate) != 0) != (p.gate_want != 0))) return;     // (the twin of a gated pair: 16-bit rows of 1 KiB)
  const geot_slab_plan &P = p.plan;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lpr = WAVE_ROW ? 64 : (1 << p.lpr_log2);
  const int G = 64 / lpr;                     // units per wave
  const int sub = WAVE_ROW ? 0 : (lane >> p.lpr_log2), c = lane & (lpr - 1);
  const int R = P.rows_per_group;
  const int hw = WMODE == 0 ? 0 : (WMODE == 1 ? 1 : p.H);
  // LDS: fp32 accumulators [4 waves][G units][R rows][lpr lanes][NV] float4 = 4 * R * NV KiB, then the staged
  // weights [4 waves][2 buffers][G units][hw][lpr edge slots] (fp32; head-major inside a unit's block: the row loop reads edge
  // b+u's weight at (a per-lane base: block + head) + b + u)
  float *accS = reinterpret_cast<float *>(smem) + ((size_t)(wave * G + sub) * R) * 4 * NV * lpr;
  float *wbase = reinterpret_cast<float *>(smem + (size_t)4 * R * 1024 * NV) + (size_t)wave * 2 * 64 * (hw > 0 ? hw : 1) +
                 (size_t)sub * lpr * (hw > 0 ? hw : 1);
  const int wbuf_stride = 64 * (hw > 0 ? hw : 1);
  const int64_t unit = ((int64_t)blockIdx.x * 4 + wave) * G + sub;
  const int64_t units = P.units;
  const char *src = static_cast<const char *>(p.src);
  const T *weight = static_cast<const T *>(p.weight);
  const bool wpo_rt = p.w_in_plan_order != 0;            // weights indexed by plan position (given so, or staged by the pre-pass)
  T *dst = static_cast<T *>(p.dst);
  const int h = WMODE >= 2 ? (c * VEC) / p.Fh : 0;
  typedef T t4_t __attribute__((ext_vector_type(4)));
#ifdef GEOT_DEV_EXPERIMENTS
  const bool ntp = p.nt_plan != 0;
#else
  constexpr bool ntp = false;                 // ("slab_nt" exists in the development build only: measured neutral)
#endif
  auto load_w4 = [&](int64_t pe) { // the 4 head weights of an edge (edge-major layout, H == 4) as floats: one 16- / 8-byte read
    const t4_t x = ntp ? __builtin_nontemporal_load(reinterpret_cast<const t4_t *>(weight + pe * 4)) : *reinterpret_cast<const t4_t *>(weight + pe * 4);
    return f4_t{(float)x[0], (float)x[1], (float)x[2], (float)x[3]};
  };
  const uint32_t src_rows = (uint32_t)p.src_rows;
  const int rsh = WAVE_ROW ? 10 : 4 + p.lpr_log2;   // log2(row bytes)
  const uint32_t c16 = (uint32_t)c * 16u;
  const __amdgpu_buffer_rsrc_t table = slab_table_rsrc(p.src, p.src_rows, rsh, p.probe);

  SlabStep lock;                               // the loose lockstep of the XCD's waves (see SlabStep)
  lock.enter(p, lane);

  // a unit's accumulator rows are touched by that unit's lanes only, each lane its own 16 bytes: plain LDS
  // read / write (ds_read_b128 / ds_write_b128), no atomics (LDS float atomics measured ~30x slower per byte)
  f4_t *accV = reinterpret_cast<f4_t *>(accS);

  for (int r = 0; r < p.rounds; ++r) {
    // serpentine over the size-sorted groups: no unit is always handed the larger group of its round
    const int64_t pos = (int64_t)r * units + ((r & 1) ? units - 1 - unit : unit);
    const bool has = pos < P.n_groups;
    int64_t e0 = has ? P.g_begin[pos] : 0;
    int len = has ? (int)(P.g_begin[pos + 1] - e0) : 0;
    int nv = has ? P.g_nv[pos] : 0;
    if constexpr (WAVE_ROW) { // one unit per wave: tell the compiler (scalar loop bounds, scalar row bases, scalar row switches)
      e0 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(e0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)e0);
      len = __builtin_amdgcn_readfirstlane(len);
      nv = __builtin_amdgcn_readfirstlane(nv);
    }
    for (int l = 0; l < R; ++l)
#pragma unroll
      for (int q = 0; q < NV; ++q) accV[((size_t)l * lpr + c) * NV + q] = f4_t{kIdent, kIdent, kIdent, kIdent};
    int maxlen = len;
    if constexpr (!WAVE_ROW) {
      for (int o = 32; o >= lpr; o >>= 1) {         // max over the wave's units (wave-uniform loop bound)
        const int other = __shfl_xor(maxlen, o, 64);
        maxlen = other > maxlen ? other : maxlen;
      }
    }
    f4_t acc[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) acc[q] = f4_t{kIdent, kIdent, kIdent, kIdent};
    int cur = 255;                                   // no open row

    // fields of the first chunk, its weights into buffer 0
    // (whole-wave rows: an edge's two fields travel as ONE word, (source row << 8) | row in group - 1-KiB rows in a table below
    // 4 GiB are < 2^22 rows, a group has at most 32, 255 = padding: one v_readlane per edge, the split is scalar arithmetic)
    int my_src = 0, my_dl = 255;
    uint32_t my_edge = 255;
    {
      const bool valid = c < len;
      my_src = valid ? P.e_src[e0 + c] : 0;
      if ((uint32_t)my_src >= src_rows) my_src = 0;   // (checked once per edge, here, not in the row loop)
      my_dl = valid ? (int)P.e_dl[e0 + c] : 255;
      my_edge = ((uint32_t)my_src << 8) | (uint32_t)my_dl;
      if constexpr (WMODE != 0) {
        const int64_t pe = valid ? (wpo_rt ? e0 + c : (int64_t)P.e_perm[e0 + c]) : 0;
        if constexpr (WMODE == 1) wbase[c] = valid ? (float)weight[pe] : 0.f;
        if constexpr (WMODE == 2) {
          if (p.H == 4) {
            const f4_t w4 = valid ? load_w4(pe) : f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) wbase[q * lpr + c] = w4[q];
          } else for (int q = 0; q < p.H; ++q) wbase[q * lpr + c] = valid ? (float)weight[pe * p.H + q] : 0.f;
        }
        if constexpr (WMODE == 3)
          for (int q = 0; q < p.H; ++q) wbase[q * lpr + c] = valid ? (float)weight[(int64_t)q * P.nnz + pe] : 0.f;
        __builtin_amdgcn_wave_barrier();
      }
    }

    // (the chunk loop exists twice, for weights in plan order and through e_perm - see seg_slab_wrow_kernel: with the choice made at
    // run time inside one body the compiler guards e_perm's in-flight entry with vmcnt(0) waits on both paths)
    auto chunks = [&](auto wpo_c) __attribute__((always_inline)) {
    constexpr bool wpo = decltype(wpo_c)::value;
    int k = 0;
    for (int off = 0; off < maxlen; off += lpr, ++k) {
      const float *wcur = wbase + (k & 1) * wbuf_stride + h * lpr;       // this lane's head
      float *wnext = wbase + ((k + 1) & 1) * wbuf_stride;
      // Whole-wave rows (a chunk = 64 edges): the next chunk's fields are loaded BEHIND the first batch's row gathers (below), not
      // here: vector-memory operations retire in issue order, so a stream read issued ahead of the gathers (an HBM miss, several times
      // a gather's L2 hit) is waited for by the chunk's first row.  Every lane loads, unconditionally (lanes behind their unit's end re-read its last edge and drop the value
      // when the chunk is handed over): a predicated load is a branch, and behind a branch the compiler waits for vmcnt(0).
      const bool nvalid = off + lpr + c < len;
      const int64_t ne = WAVE_ROW ? e0 + (nvalid ? off + lpr + c : (len > 0 ? len - 1 : 0)) : e0 + off + lpr + c;
      int n_src = 0, n_dl = 255;
      uint32_t n_pe32 = 0;                            // (e_perm's entry stays as loaded until the weight read uses it)
      // (the next chunk's weights are held AS LOADED until they are staged at the end of the chunk: converting a 16-bit weight here
      // would put a full wait - behind the batch of row loads just issued - into the first batch of every chunk)
      t4_t wn4;
#pragma unroll
      for (int q = 0; q < 4; ++q) wn4[q] = (T)0.f;
      T wn1 = (T)0.f;

      const int n_here = len - off;                  // edges of this unit in the chunk (<= 0: none)
      int n_max = maxlen - off;
      n_max = n_max < lpr ? n_max : lpr;
      auto next_fields = [&]() __attribute__((always_inline)) {
        if (WAVE_ROW || nvalid) {                    // (lane groups: predicated - their chunks are short, the clamp's arithmetic shows)
          n_src = ntp ? __builtin_nontemporal_load(P.e_src + ne) : P.e_src[ne];
          n_dl = (int)(ntp ? __builtin_nontemporal_load(P.e_dl + ne) : P.e_dl[ne]);
          if constexpr (WMODE != 0) {
            if (!wpo) n_pe32 = (uint32_t)(ntp ? __builtin_nontemporal_load(P.e_perm + ne) : P.e_perm[ne]);
            else {                                   // ... and its weights, when they sit in plan order
              if constexpr (WMODE == 1) wn1 = weight[ne];
              if constexpr (WMODE == 2) {
                if (p.H == 4) wn4 = ntp ? __builtin_nontemporal_load(reinterpret_cast<const t4_t *>(weight + ne * 4)) : *reinterpret_cast<const t4_t *>(weight + ne * 4);
              }
            }
          }
        }
      };
      auto batch = [&](const int b, auto first_c) __attribute__((always_inline)) {     // (the first batch is its own copy of the body)
        constexpr bool kFirst = decltype(first_c)::value;
        if constexpr (WMODE == 1 || WMODE == 2) {
          // the next chunk's weights, in edge order: their place comes out of e_perm (loaded behind the first batch) - read at the TOP
          // of the second batch, where everything outstanding is a batch old, not behind that batch's gathers
          if (!kFirst && !wpo && b == kU) {
            uint32_t pe = n_pe32;
            asm volatile("" : "+v"(pe));      // (keeps the address arithmetic - and the wait for e_perm's entry - HERE, not at its load)
            if constexpr (WMODE == 1) wn1 = weight[pe];
            else if (p.H == 4) wn4 = ntp ? __builtin_nontemporal_load(reinterpret_cast<const t4_t *>(weight + (size_t)pe * 4)) : *reinterpret_cast<const t4_t *>(weight + (size_t)pe * 4);
          }
        }
        if (p.window >= 0) {                         // which slab does this batch start in (the wave's first unit decides)
          const int first_row = WAVE_ROW ? (int)((uint32_t)__builtin_amdgcn_readlane(my_edge, b) >> 8) : __builtin_amdgcn_readlane(my_src, b);
          const int has_edge = __builtin_amdgcn_readfirstlane(n_here) > b;
          if (has_edge) lock.at(p, lane, r * p.n_slabs + (first_row >> p.slab_shift));
        }
        f4_t v[kU];
        int dls[kU];
        float ws[kU];
        // Slots behind the unit's last edge need no test here: their lane holds row 0 and dl = 255 (set where the fields were
        // loaded), the weight read stays inside the staging buffer, and what they add goes to the row that is never written
        // back.  The row's byte offset is a 32-bit shift (rows are 128 B .. 1 KiB, the table is below 4 GiB - geot_slab_spmm
        // checks): scalar base + 32-bit lane offset, no 64-bit multiply per edge.
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          if constexpr (WAVE_ROW) {
            const uint32_t edge = (uint32_t)__builtin_amdgcn_readlane(my_edge, b + u);
            dls[u] = (int)(edge & 255u);
            if constexpr (WMODE != 0) ws[u] = wcur[b + u];
            v[u] = slab_row_load<f4_t>(table, c16, (edge & ~255u) << 2);       // (row << 10)
          } else {
            const uint32_t row = (uint32_t)__shfl(my_src, b + u, lpr);
            dls[u] = __shfl(my_dl, b + u, lpr);
            if constexpr (WMODE != 0) ws[u] = wcur[b + u];
            v[u] = *reinterpret_cast<const f4_t *>(src + (size_t)((row << rsh) + c16));
          }
        }
        if constexpr (kFirst && WAVE_ROW) next_fields();     // behind this batch's gathers
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          if (__builtin_expect(dls[u] != cur, 0)) {  // the open row goes back to LDS, the new one comes out of it (rare: laid out of line)
            if (cur != 255) {
#pragma unroll
              for (int q = 0; q < NV; ++q) accV[((size_t)cur * lpr + c) * NV + q] = acc[q];
            }
            cur = dls[u];
#pragma unroll
            for (int q = 0; q < NV; ++q) acc[q] = cur != 255 ? accV[((size_t)cur * lpr + c) * NV + q] : f4_t{kIdent, kIdent, kIdent, kIdent};
          }
          // padding slots of a short unit (dl = 255): their sum goes to a row that is never written back, but a
          // max / min must not see their value at all
          if ((RED != GEOT_REDUCE_MAX && RED != GEOT_REDUCE_MIN) || dls[u] != 255) {
            f4_t m[NV];
            slab_unpack<T>(v[u], m);
#pragma unroll
            for (int q = 0; q < NV; ++q) {
              if constexpr (WMODE == 0) slab_acc<RED>(acc[q], m[q]);
              else slab_acc<RED>(acc[q], m[q] * ws[u]);
            }
          }
        }
      };
      // (lane groups: a chunk is lpr = 32 / 16 / 8 edges - two to four batches -, so the fields go first: behind the first batch they
      // would have half the chunk to arrive; measured 1 % slower on the weightless plans)
      if constexpr (!WAVE_ROW) next_fields();
      batch(0, std::true_type{});
      for (int b = kU; b < n_max; b += kU) batch(b, std::false_type{});
      if constexpr (WMODE == 1 || WMODE == 2) {
        // (a chunk of ONE batch - 128-byte rows, lpr = 8 - has no second batch to fetch the edge-order weights in: here, then)
        if (!wpo && n_max <= kU && nvalid) {
          if constexpr (WMODE == 1) wn1 = weight[n_pe32];
          else if (p.H == 4) wn4 = *reinterpret_cast<const t4_t *>(weight + (size_t)n_pe32 * 4);
        }
      }
      // stage the next chunk
      if constexpr (WMODE != 0) {
        if constexpr (WMODE == 1) wnext[c] = nvalid ? (float)wn1 : 0.f;
        if constexpr (WMODE == 2) {
          if (p.H == 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) wnext[q * lpr + c] = nvalid ? (float)wn4[q] : 0.f;
          } else {
            const int64_t n_pe = wpo ? ne : (int64_t)n_pe32;
            for (int q = 0; q < p.H; ++q) wnext[q * lpr + c] = nvalid ? (float)weight[n_pe * p.H + q] : 0.f;
          }
        }
        if constexpr (WMODE == 3) {
          const int64_t n_pe = wpo ? ne : (int64_t)n_pe32;
          for (int q = 0; q < p.H; ++q) wnext[q * lpr + c] = nvalid ? (float)weight[(int64_t)q * P.nnz + n_pe] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
      }
      my_src = (nvalid && (uint32_t)n_src < src_rows) ? n_src : 0;
      my_dl = nvalid ? n_dl : 255;
      my_edge = ((uint32_t)my_src << 8) | (uint32_t)my_dl;
    }
    };
    if (WMODE == 0 || wpo_rt) chunks(std::true_type{});
    else chunks(std::false_type{});
    if (cur != 255) {
#pragma unroll
      for (int q = 0; q < NV; ++q) accV[((size_t)cur * lpr + c) * NV + q] = acc[q];
    }
    lock.round_done(p, lane, r);
    // the group's rows: dst for whole rows, the carry buffer for the pieces of a split hub
    if (has) {
      const int64_t v0 = P.g_vrow0[pos];
      for (int l = 0; l < nv; ++l) {
        const int64_t t = P.v_out[v0 + l];
        f4_t row[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) row[q] = accV[((size_t)l * lpr + c) * NV + q];
        if constexpr (RED == GEOT_REDUCE_MEAN) {
          if (t >= 0) {                                        // pieces of a split row are divided after the combine
#pragma unroll
            for (int q = 0; q < NV; ++q) row[q] = row[q] / (float)P.v_total[v0 + l];
          }
        }
        if (t >= 0) {
          if (t < p.K) *reinterpret_cast<f4_t *>(dst + t * p.F + c * VEC) = slab_pack<T>(row);   // one rounding, here
        } else {
#pragma unroll
          for (int q = 0; q < NV; ++q) *reinterpret_cast<f4_t *>(p.carry + (-t - 1) * p.F + c * VEC + 4 * q) = row[q];   // fp32
        }
      }
    }
  }
  lock.leave(lane);
}

// Rows of 512 / 256 bytes: ONE ROW PER WAVE-INSTRUCTION, E = row elements / 64 per lane (8 or 4 bytes a lane) - the scalar path of
// seg_slab_kernel's 1-KiB form (group bounds, row numbers and row switches in SGPRs, 32-bit row offsets on a scalar base) for the
// narrower rows, which the lane-group form served with a ds_bpermute per field and per-lane row switches (round 4: multi-head
// weights only - bf16 H=4 x F=64 at Reddit scale 5.66 -> 5.24 ms; round 5: every weight mode and reduction).  A unit is a wave
// whatever the row width: the plan is built with units = waves (geot_slab_units_for) and R rows of 64 x E fp32 accumulators per
// group.  WMODE 0 none | 1 weight[e] | 2 weight[e*H + h] | 3 weight[h*nnz + e]; p.w_in_plan_order (WMODE 1 / 2): the weight
// array is indexed by PLAN position (a producer that emits its values in plan order - geot_slab_sddmm's staged results, a
// static weight permuted once): no read through the edge permutation.  RED: sum | mean | max | min for WMODE 0 / 1 (multi-head: sum).
template <typename T, int E> struct SlabRaw;                       // the lane's E elements as one load
template <> struct SlabRaw<float, 1> { typedef float type; };
template <> struct SlabRaw<float, 2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct SlabRaw<half_t, 2> { typedef uint32_t type; };
template <> struct SlabRaw<half_t, 4> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <> struct SlabRaw<bf16_t, 2> { typedef uint32_t type; };
template <> struct SlabRaw<bf16_t, 4> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <int E> struct SlabAcc { typedef float type __attribute__((ext_vector_type(E))); };
template <> struct SlabAcc<1> { typedef float type; };

template <typename T, int E> __device__ __forceinline__ void mhrow_unpack(const typename SlabRaw<T, E>::type &raw, float (&m)[E]) {
  if constexpr (sizeof(T) == 4) {
    if constexpr (E == 1) m[0] = raw;
    else {
      m[0] = raw[0];
      m[1] = raw[1];
    }
  } else {
    typedef T tE __attribute__((ext_vector_type(E)));
    const tE x = __builtin_bit_cast(tE, raw);
#pragma unroll
    for (int i = 0; i < E; ++i) m[i] = (float)x[i];
  }
}
template <typename T, int E> __device__ __forceinline__ typename SlabRaw<T, E>::type mhrow_pack(const float (&m)[E]) {
  if constexpr (sizeof(T) == 4) {
    if constexpr (E == 1) return m[0];
    else return typename SlabRaw<T, E>::type{m[0], m[1]};
  } else {
    typedef T tE __attribute__((ext_vector_type(E)));
    tE x;
#pragma unroll
    for (int i = 0; i < E; ++i) x[i] = (T)m[i];
    return __builtin_bit_cast(typename SlabRaw<T, E>::type, x);
  }
}

template <typename T, int WMODE, int E, int RED, int U>
__global__ __launch_bounds__(kThreads) void seg_slab_wrow_kernel(SlabParams p) {
  static_assert(WMODE >= 0 && WMODE <= 3 && (WMODE <= 1 || RED == GEOT_REDUCE_SUM), "multi-head weights sum");
  constexpr float kIdent = RED == GEOT_REDUCE_MAX ? -INFINITY : (RED == GEOT_REDUCE_MIN ? INFINITY : 0.f);
  constexpr bool kSel = RED == GEOT_REDUCE_MAX || RED == GEOT_REDUCE_MIN;
  typedef typename SlabRaw<T, E>::type raw_t;
  typedef typename SlabAcc<E>::type accv_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const geot_slab_plan &P = p.plan;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int R = P.rows_per_group;
  const int hw = WMODE <= 1 ? 1 : p.H;
  // LDS: fp32 accumulators [4 waves][R rows][64 lanes][E], then the staged weights [4 waves][2 buffers][hw][64 edge slots] -
  // HEAD-major inside a buffer: the row loop's read of edge b+u's weight is (a per-lane base: buffer + head) + a scalar batch offset
  // + an immediate, no address arithmetic per edge
  accv_t *accV = reinterpret_cast<accv_t *>(smem) + (size_t)wave * R * 64;
  float *wbase = reinterpret_cast<float *>(smem + (size_t)4 * R * 64 * E * sizeof(float)) + (size_t)wave * 2 * 64 * hw;
  const int wbuf_stride = 64 * hw;
  const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
  const int64_t units = P.units;
  const char *src = static_cast<const char *>(p.src);
  const T *weight = static_cast<const T *>(p.weight);
  T *dst = static_cast<T *>(p.dst);
  const int h = WMODE >= 2 ? (lane * E) / p.Fh : 0;
  const bool wpo_rt = p.w_in_plan_order != 0;           // weights indexed by plan position (given so, or staged by the pre-pass)
  const uint32_t src_rows = (uint32_t)p.src_rows;
  const int rsh = 4 + p.lpr_log2;                       // log2(row bytes)
  const uint32_t cB = (uint32_t)lane * (uint32_t)(E * sizeof(T));
  const __amdgpu_buffer_rsrc_t table = slab_table_rsrc(p.src, p.src_rows, rsh, p.probe);
  typedef T t4_t __attribute__((ext_vector_type(4)));
  auto load_w4 = [&](int64_t pe) {
    const t4_t x = *reinterpret_cast<const t4_t *>(weight + pe * 4);
    return f4_t{(float)x[0], (float)x[1], (float)x[2], (float)x[3]};
  };

  if (p.gate && ((slab_gate_word(p.gate) != 0) != (p.gate_want != 0))) return;   // (the other twin of a gated pair runs: see SlabParams::gate)
  SlabStep lock;                               // the loose lockstep of the XCD's waves (see SlabStep)
  lock.enter(p, lane);
  auto zero = [] {                                      // (the reduction's identity)
    accv_t z;
    if constexpr (E == 1) z = kIdent;
    else {
#pragma unroll
      for (int i = 0; i < E; ++i) z[i] = kIdent;
    }
    return z;
  };

  for (int r = 0; r < p.rounds; ++r) {
    const int64_t pos = (int64_t)r * units + ((r & 1) ? units - 1 - unit : unit);
    const bool has = pos < P.n_groups;
    int64_t e0 = has ? P.g_begin[pos] : 0;
    int len = has ? (int)(P.g_begin[pos + 1] - e0) : 0;
    int nv = has ? P.g_nv[pos] : 0;
    e0 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(e0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)e0);
    len = __builtin_amdgcn_readfirstlane(len);
    nv = __builtin_amdgcn_readfirstlane(nv);
    for (int l = 0; l < R; ++l) accV[(size_t)l * 64 + lane] = zero();
    accv_t acc = zero();
    int cur = 255;

    // An edge's two fields travel as ONE word, (source row << 8) | row in group (rows are >= 256 B in a table below 4 GiB: < 2^24
    // rows; a group has at most 32 rows, 255 = padding): one v_readlane per edge instead of two, the split is scalar arithmetic
    int my_src = 0, my_dl = 255;
    uint32_t my_edge = 255;
    {
      const bool valid = lane < len;
      my_src = valid ? P.e_src[e0 + lane] : 0;
      if ((uint32_t)my_src >= src_rows) my_src = 0;
      my_dl = valid ? (int)P.e_dl[e0 + lane] : 255;
      my_edge = ((uint32_t)my_src << 8) | (uint32_t)my_dl;
      if constexpr (WMODE != 0) {
        const int64_t pe = valid ? (wpo_rt ? e0 + lane : (int64_t)P.e_perm[e0 + lane]) : 0;
        if constexpr (WMODE == 1) wbase[lane] = valid ? (float)weight[pe] : 0.f;
        else if constexpr (WMODE == 2) {
          if (p.H == 4) {
            const f4_t w4 = valid ? load_w4(pe) : f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) wbase[q * 64 + lane] = w4[q];
          } else for (int q = 0; q < p.H; ++q) wbase[q * 64 + lane] = valid ? (float)weight[pe * p.H + q] : 0.f;
        } else {
          for (int q = 0; q < p.H; ++q) wbase[q * 64 + lane] = valid ? (float)weight[(int64_t)q * P.nnz + pe] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }

    // (the chunk loop exists twice, for weights in plan order and through e_perm: with the choice made at run time inside one body the
    // compiler guards e_perm's in-flight entry with vmcnt(0) waits on BOTH paths)
    auto chunks = [&](auto wpo_c) __attribute__((always_inline)) {
    constexpr bool wpo = decltype(wpo_c)::value;
    int k = 0;
    for (int off = 0; off < len; off += 64, ++k) {
      const float *wcur = wbase + (k & 1) * wbuf_stride + h * 64;          // this lane's head
      float *wnext = wbase + ((k + 1) & 1) * wbuf_stride;
      // The next chunk's fields are loaded BEHIND the first batch's row gathers (below), not here: vector-memory operations retire in
      // issue order, so a stream read issued ahead of the gathers (an HBM miss, several times a gather's L2 hit) is waited for by the
      // very first row of the chunk - one full miss per 64 edges on every wave's critical path.
      // They are loaded by EVERY lane, unconditionally (lanes behind the group's end re-read its last edge and drop the value when the
      // chunk is handed over): a predicated load is a branch to the compiler, and behind a branch it waits for vmcnt(0).
      const bool nvalid = off + 64 + lane < len;
      const int64_t ne = e0 + (nvalid ? off + 64 + lane : len - 1);
      int n_src = 0, n_dl = 255;
      uint32_t n_pe32 = 0;                               // (e_perm's entry stays as loaded until the weight read uses it)
      // (the next chunk's weights are held AS LOADED until they are staged at the end of the chunk: converting a 16-bit weight here
      // would put a full wait - behind the batch of row loads just issued - into the first batch of every chunk)
      t4_t wn4;
#pragma unroll
      for (int q = 0; q < 4; ++q) wn4[q] = (T)0.f;
      T wn1 = (T)0.f;
      int n_max = len - off;
      n_max = n_max < 64 ? n_max : 64;
      // (the first batch is its own copy of the body: there the next chunk's field loads are unconditional, and the compiler counts
      // them - a conditional load inside one shared body made every batch's last row wait for vmcnt(0))
      auto batch = [&](const int b, auto first_c) __attribute__((always_inline)) {
        constexpr bool kFirst = decltype(first_c)::value;
        if (p.window >= 0) lock.at(p, lane, r * p.n_slabs + (int)((uint32_t)__builtin_amdgcn_readlane(my_edge, b) >> (8 + p.slab_shift)));
        if constexpr (WMODE == 1 || WMODE == 2) {
          // the next chunk's weights, in edge order: their place comes out of e_perm (loaded behind the first batch) - read it at the
          // TOP of the second batch, where everything outstanding is a batch old, not behind that batch's gathers
          if (!kFirst && !wpo && b == U) {
            uint32_t pe = n_pe32;
            asm volatile("" : "+v"(pe));      // (keeps the address arithmetic - and with it the wait for e_perm's entry - HERE, not at its load)
            if constexpr (WMODE == 1) wn1 = weight[pe];
            else if (p.H == 4) wn4 = *reinterpret_cast<const t4_t *>(weight + (size_t)pe * 4);
          }
        }
        raw_t v[U];
        int dls[U];
        float ws[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {       // (slots behind the last edge: row 0, dl = 255, weight 0 - see seg_slab_kernel)
          const uint32_t edge = (uint32_t)__builtin_amdgcn_readlane(my_edge, b + u);
          dls[u] = (int)(edge & 255u);
          if constexpr (WMODE != 0) ws[u] = wcur[b + u];
          v[u] = slab_row_load<raw_t>(table, cB, (edge & ~255u) << (rsh - 8));
        }
        if constexpr (kFirst) {
          n_src = P.e_src[ne];
          n_dl = (int)P.e_dl[ne];
          if constexpr (WMODE != 0) {
            if (!wpo) n_pe32 = (uint32_t)P.e_perm[ne];
          }
        }
        if (kFirst && wpo) {                // the next chunk's weights, in plan order: with the fields
          if constexpr (WMODE == 1) wn1 = weight[ne];
          if constexpr (WMODE == 2) {
            if (p.H == 4) wn4 = *reinterpret_cast<const t4_t *>(weight + ne * 4);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (__builtin_expect(dls[u] != cur, 0)) {      // (rare: the switch is laid out of line, the common path falls through)
            if (cur != 255) accV[(size_t)cur * 64 + lane] = acc;
            cur = dls[u];
            acc = cur != 255 ? accV[(size_t)cur * 64 + lane] : zero();
          }
          if (!kSel || dls[u] != 255) {      // (a max / min must not see the padding slots' values; their sums go to a row never written back)
            float m[E];
            mhrow_unpack<T, E>(v[u], m);
            const float wu = WMODE != 0 ? ws[u] : 1.f;
            if constexpr (E == 1) acc = slab_op<RED>(acc, WMODE != 0 ? m[0] * wu : m[0]);
            else {
#pragma unroll
              for (int i = 0; i < E; ++i) acc[i] = slab_op<RED>(acc[i], WMODE != 0 ? m[i] * wu : m[i]);
            }
          }
        }
      };
      batch(0, std::true_type{});
      for (int b = U; b < n_max; b += U) batch(b, std::false_type{});
      if constexpr (WMODE == 1) wnext[lane] = nvalid ? (float)wn1 : 0.f;
      else if constexpr (WMODE == 2) {
        if (p.H == 4) {
#pragma unroll
          for (int q = 0; q < 4; ++q) wnext[q * 64 + lane] = nvalid ? (float)wn4[q] : 0.f;
        } else {
          const int64_t n_pe = wpo ? ne : (int64_t)n_pe32;
          for (int q = 0; q < p.H; ++q) wnext[q * 64 + lane] = nvalid ? (float)weight[n_pe * p.H + q] : 0.f;
        }
      } else if constexpr (WMODE == 3) {
        const int64_t n_pe = wpo ? ne : (int64_t)n_pe32;
        for (int q = 0; q < p.H; ++q) wnext[q * 64 + lane] = nvalid ? (float)weight[(int64_t)q * P.nnz + n_pe] : 0.f;
      }
      if constexpr (WMODE != 0) __builtin_amdgcn_wave_barrier();
      my_src = (nvalid && (uint32_t)n_src < src_rows) ? n_src : 0;
      my_dl = nvalid ? n_dl : 255;
      my_edge = ((uint32_t)my_src << 8) | (uint32_t)my_dl;
    }
    };
    if (WMODE == 0 || wpo_rt) chunks(std::true_type{});
    else chunks(std::false_type{});
    if (cur != 255) accV[(size_t)cur * 64 + lane] = acc;
    lock.round_done(p, lane, r);
    if (has) {
      const int64_t v0 = P.g_vrow0[pos];
      for (int l = 0; l < nv; ++l) {
        const int64_t t = P.v_out[v0 + l];
        const accv_t row = accV[(size_t)l * 64 + lane];
        float m[E];
        if constexpr (E == 1) m[0] = row;
        else {
#pragma unroll
          for (int i = 0; i < E; ++i) m[i] = row[i];
        }
        if constexpr (RED == GEOT_REDUCE_MEAN) {
          if (t >= 0) {                                        // (pieces of a split row are divided after the combine)
            const float tot = (float)P.v_total[v0 + l];
#pragma unroll
            for (int i = 0; i < E; ++i) m[i] = m[i] / tot;
          }
        }
        if (t >= 0) {
          if (t < p.K) *reinterpret_cast<raw_t *>(dst + t * p.F + lane * E) = mhrow_pack<T, E>(m);   // one rounding, here
        } else {
          *reinterpret_cast<accv_t *>(p.carry + (-t - 1) * p.F + lane * E) = row;                     // fp32
        }
      }
    }
  }
  lock.leave(lane);
}

#ifdef GEOT_DEV_EXPERIMENTS
// Rows of 512 bytes under multi-head weights, TWO ROWS PER WAVE-INSTRUCTION (an EXPERIMENT, option "slab_pair"; off by default): the
// plan of seg_slab_wrow_kernel (a unit = a wave, R rows per group, group bounds / row switches in SGPRs) read at 16 bytes a lane -
// lanes 0..31 gather the row of edge 2j, lanes 32..63 the row of edge 2j+1 OF THE SAME UNIT.  The idea: fp32 rows of 512 B run at
// 3.14 ms with 16 bytes a lane (lane groups) and 4.15 ms with 8 (one row per instruction), so halve the gather instructions of the
// 16-bit multi-head plans too, without the lane groups' per-lane row switches and uneven units.  The result (bf16 H=4 x F=64 at
// Reddit scale, weights in plan order): 4.63 ms against seg_slab_wrow_kernel's 4.50 - SLOWER, and the counters say why
// (profiles/r05/pmc_slab_probe/): 12.2 vector instructions per edge against 10.4 (eight unpacks and four packed FMAs per pair stay,
// the per-half row offsets cost a ds_bpermute + shift-or per pair, the row switches are tested twice), LDS instructions 1.9 per edge
// against 0.6, and neither kernel waits for its row gathers in the first place (with the gathers DROPPED, "slab_probe", both keep
// 97-98 % of their time).  Kept for rows where the balance differs; tests/test_gpu_round5.py runs it against the float64 sums.
// Each half keeps ITS open row's partial sum in registers; a half's row switch (a
// scalar test per edge, as before) ADDS its partial sum into the row's LDS accumulators - read-add-write by that half's lanes only,
// the halves one after the other (LDS operations of a wave execute in order), so two halves holding pieces of the same row never
// collide - and starts the new row from zero: the same LDS traffic as one row per instruction (which wrote the old row and read the
// new one).  A row's sum is (sum over its even-position edges) + (sum over its odd-position edges), in plan order inside each: fixed
// by the plan, not by timing.  WMODE 2 weight[e*H + h] | 3 weight[h*nnz + e]; p.w_in_plan_order as in seg_slab_wrow_kernel.
template <typename T, int WMODE>
__global__ __launch_bounds__(kThreads) void seg_slab_wpair_kernel(SlabParams p) {
  static_assert(WMODE == 2 || WMODE == 3, "multi-head weights");
  constexpr int VEC = SlabVec<T>::VEC, NV = SlabVec<T>::NV;
  constexpr int kI = 8;                                  // gather instructions in flight per lane = 16 edges a batch
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const geot_slab_plan &P = p.plan;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, c = lane & 31;
  const int R = P.rows_per_group;
  const int hw = p.H;
  // LDS: fp32 accumulators [4 waves][R rows][32 lanes][NV] float4 (= seg_slab_wrow_kernel's bytes), then the staged weights
  // [4 waves][2 buffers][hw][64 edge slots]
  f4_t *accV = reinterpret_cast<f4_t *>(smem) + (size_t)wave * R * 32 * NV;
  float *wbase = reinterpret_cast<float *>(smem + (size_t)4 * R * 32 * NV * sizeof(f4_t)) + (size_t)wave * 2 * 64 * hw;
  const int wbuf_stride = 64 * hw;
  const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
  const int64_t units = P.units;
  const T *weight = static_cast<const T *>(p.weight);
  T *dst = static_cast<T *>(p.dst);
  const int h = (c * VEC) / p.Fh;
  const bool wpo = p.w_in_plan_order != 0;
  const uint32_t src_rows = (uint32_t)p.src_rows;
  constexpr int rsh = 9;                                 // log2(row bytes)
  const uint32_t c16 = (uint32_t)c * 16u;
  const __amdgpu_buffer_rsrc_t table = slab_table_rsrc(p.src, p.src_rows, rsh, p.probe);
  typedef T t4_t __attribute__((ext_vector_type(4)));

  SlabStep lock;                               // the loose lockstep of the XCD's waves (see SlabStep)
  lock.enter(p, lane);
  const f4_t z4 = {0.f, 0.f, 0.f, 0.f};

  for (int r = 0; r < p.rounds; ++r) {
    const int64_t pos = (int64_t)r * units + ((r & 1) ? units - 1 - unit : unit);
    const bool has = pos < P.n_groups;
    int64_t e0 = has ? P.g_begin[pos] : 0;
    int len = has ? (int)(P.g_begin[pos + 1] - e0) : 0;
    int nv = has ? P.g_nv[pos] : 0;
    e0 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(e0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)e0);
    len = __builtin_amdgcn_readfirstlane(len);
    nv = __builtin_amdgcn_readfirstlane(nv);
    for (int l = half; l < R; l += 2)
#pragma unroll
      for (int q = 0; q < NV; ++q) accV[((size_t)l * 32 + c) * NV + q] = z4;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (a row zeroed by one half is added into by the other)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    f4_t acc[NV];                                         // this half's open row: lanes 0..31 the even-position edges', 32..63 the odd ones'
#pragma unroll
    for (int q = 0; q < NV; ++q) acc[q] = z4;
    int cur0 = 255, cur1 = 255;                           // the halves' open rows (wave-uniform)
    auto hand_in = [&](int which, int row) {              // half `which` adds its partial sum into LDS row `row` and starts from zero
      if (half == which) {
        if (row != 255) {
#pragma unroll
          for (int q = 0; q < NV; ++q) accV[((size_t)row * 32 + c) * NV + q] += acc[q];
        }
#pragma unroll
        for (int q = 0; q < NV; ++q) acc[q] = z4;
      }
      // The other half's lanes may add into the SAME LDS words next (both halves often hold pieces of one row): to the compiler those
      // are other threads, and without this it is free to read for both halves first and write twice (seen: a hub lost 18 % of its
      // pieces in the 16-bit instantiations).  The hardware runs a wave's LDS operations in order; the compiler must not reorder them.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    int my_src = 0, my_dl = 255;
    {
      const bool valid = lane < len;
      my_src = valid ? P.e_src[e0 + lane] : 0;
      if ((uint32_t)my_src >= src_rows) my_src = 0;
      my_dl = valid ? (int)P.e_dl[e0 + lane] : 255;
      const int64_t pe = valid ? (wpo ? e0 + lane : (int64_t)P.e_perm[e0 + lane]) : 0;
      if constexpr (WMODE == 2) {
        if (p.H == 4) {
          t4_t w4;
#pragma unroll
          for (int q = 0; q < 4; ++q) w4[q] = (T)0.f;
          if (valid) w4 = *reinterpret_cast<const t4_t *>(weight + pe * 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) wbase[q * 64 + lane] = (float)w4[q];
        } else for (int q = 0; q < p.H; ++q) wbase[q * 64 + lane] = valid ? (float)weight[pe * p.H + q] : 0.f;
      } else {
        for (int q = 0; q < p.H; ++q) wbase[q * 64 + lane] = valid ? (float)weight[(int64_t)q * P.nnz + pe] : 0.f;
      }
      __builtin_amdgcn_wave_barrier();
    }

    int k = 0;
    for (int off = 0; off < len; off += 64, ++k) {
      const float *wcur = wbase + (k & 1) * wbuf_stride + h * 64 + half;    // this lane's head, this half's edge of a pair
      float *wnext = wbase + ((k + 1) & 1) * wbuf_stride;
      const bool nvalid = off + 64 + lane < len;
      const int64_t ne = e0 + off + 64 + lane;
      int n_src = nvalid ? P.e_src[ne] : 0;
      if ((uint32_t)n_src >= src_rows) n_src = 0;
      const int n_dl = nvalid ? (int)P.e_dl[ne] : 255;
      const int64_t n_pe = nvalid ? (wpo ? ne : (int64_t)P.e_perm[ne]) : 0;
      t4_t wn4;                                           // (held as loaded until staged: see seg_slab_kernel)
#pragma unroll
      for (int q = 0; q < 4; ++q) wn4[q] = (T)0.f;
      int n_max = len - off;
      n_max = n_max < 64 ? n_max : 64;
      for (int b = 0; b < n_max; b += 2 * kI) {
        if (p.window >= 0) lock.at(p, lane, r * p.n_slabs + (__builtin_amdgcn_readlane(my_src, b) >> p.slab_shift));
        f4_t v[kI];
        int dl0[kI], dl1[kI];
        float ws[kI];
        const int pick = (b + half) << 2;
#pragma unroll
        for (int u = 0; u < kI; ++u) {      // (slots behind the last edge: row 0, dl = 255, weight 0 - see seg_slab_kernel)
          // this half's source row: lane (b + 2u + half)'s field through the LDS crossbar (one ds_bpermute, the pair offset in its
          // immediate) - two v_readlane + a select + moves cost five vector instructions per pair
          const uint32_t row = (uint32_t)__builtin_amdgcn_ds_bpermute(pick + 8 * u, my_src);
          dl0[u] = __builtin_amdgcn_readlane(my_dl, b + 2 * u);
          dl1[u] = __builtin_amdgcn_readlane(my_dl, b + 2 * u + 1);
          ws[u] = wcur[b + 2 * u];
          v[u] = slab_row_load<f4_t>(table, (row << rsh) + c16, 0);
        }
        if constexpr (WMODE == 2) {
          if (b == 0 && p.H == 4 && nvalid) wn4 = *reinterpret_cast<const t4_t *>(weight + n_pe * 4);
        }
#pragma unroll
        for (int u = 0; u < kI; ++u) {
          if (__builtin_expect(dl0[u] != cur0, 0)) {
            hand_in(0, cur0);
            cur0 = dl0[u];
          }
          if (__builtin_expect(dl1[u] != cur1, 0)) {
            hand_in(1, cur1);
            cur1 = dl1[u];
          }
          f4_t m[NV];
          slab_unpack<T>(v[u], m);
#pragma unroll
          for (int q = 0; q < NV; ++q) acc[q] += m[q] * ws[u];
        }
      }
      if constexpr (WMODE == 2) {
        if (p.H == 4) {
#pragma unroll
          for (int q = 0; q < 4; ++q) wnext[q * 64 + lane] = (float)wn4[q];
        } else for (int q = 0; q < p.H; ++q) wnext[q * 64 + lane] = nvalid ? (float)weight[n_pe * p.H + q] : 0.f;
      } else {
        for (int q = 0; q < p.H; ++q) wnext[q * 64 + lane] = nvalid ? (float)weight[(int64_t)q * P.nnz + n_pe] : 0.f;
      }
      __builtin_amdgcn_wave_barrier();
      my_src = n_src;
      my_dl = n_dl;
    }
    hand_in(0, cur0);
    hand_in(1, cur1);
    lock.round_done(p, lane, r);
    if (has) {                                            // the group's rows, two at a time (a half each)
      const int64_t v0 = P.g_vrow0[pos];
      for (int l0 = 0; l0 < nv; l0 += 2) {
        const int l = l0 + half;
        if (l < nv) {
          const int64_t t = P.v_out[v0 + l];
          f4_t row[NV];
#pragma unroll
          for (int q = 0; q < NV; ++q) row[q] = accV[((size_t)l * 32 + c) * NV + q];
          if (t >= 0) {
            if (t < p.K) *reinterpret_cast<f4_t *>(dst + t * p.F + c * VEC) = slab_pack<T>(row);      // one rounding, here
          } else {
#pragma unroll
            for (int q = 0; q < NV; ++q) *reinterpret_cast<f4_t *>(p.carry + (-t - 1) * p.F + c * VEC + 4 * q) = row[q];   // fp32
          }
        }
      }
    }
  }
  lock.leave(lane);
}
#endif // GEOT_DEV_EXPERIMENTS


// SDDMM over the same plan (d/dweight of gather_weight_scatter on a dense graph): out[e] = <m1[dst(e)], m2[src(e)]>.
// The unit's <= R rows of m1 (the dst side: shared by all edges of a row) sit in LDS where the forward kernel keeps its
// accumulators; the m2 rows are gathered slab by slab in the same loose lockstep.  The 8 dot products of a batch are
// reduced across the unit's lanes together (3 halving exchanges + log2(lpr/8) plain ones instead of 8 x log2(lpr)),
// and written to out[original edge id] by the 8 lanes that end up holding them.
template <typename T, bool WAVE_ROW>
__global__ __launch_bounds__(kThreads) void seg_slab_sddmm_kernel(SlabParams p) {
  constexpr int VEC = SlabVec<T>::VEC, NV = SlabVec<T>::NV;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const geot_slab_plan &P = p.plan;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lpr = WAVE_ROW ? 64 : (1 << p.lpr_log2);
  const int G = 64 / lpr;
  const int sub = WAVE_ROW ? 0 : (lane >> p.lpr_log2), c = lane & (lpr - 1);
  const int R = P.rows_per_group;
  f4_t *rowV = reinterpret_cast<f4_t *>(smem) + ((size_t)(wave * G + sub) * R) * lpr;
  const int64_t unit = ((int64_t)blockIdx.x * 4 + wave) * G + sub;
  const int64_t units = P.units;
  const char *m2 = static_cast<const char *>(p.src);
  const T *m1 = static_cast<const T *>(p.weight);           // (the dst-side matrix travels in the `weight` slot)
  T *out = static_cast<T *>(p.dst);
  const uint32_t src_rows = (uint32_t)p.src_rows;
  const int rsh = WAVE_ROW ? 10 : 4 + p.lpr_log2;   // log2(row bytes)
  const uint32_t c16 = (uint32_t)c * 16u;
  const __amdgpu_buffer_rsrc_t table = slab_table_rsrc(p.src, p.src_rows, rsh, p.probe);
  int lph_log2 = WAVE_ROW ? 6 : p.lpr_log2;          // lanes per head (H = 1, 2, 4, 8 on whole-wave rows; the launcher checks)
  for (int hh = p.H; hh > 1; hh >>= 1) --lph_log2;
  const int lph = 1 << lph_log2;

  SlabStep lock;                               // the loose lockstep of the XCD's waves (see SlabStep)
  lock.enter(p, lane);

  for (int r = 0; r < p.rounds; ++r) {
    const int64_t pos = (int64_t)r * units + ((r & 1) ? units - 1 - unit : unit);
    const bool has = pos < P.n_groups;
    int64_t e0 = has ? P.g_begin[pos] : 0;
    int len = has ? (int)(P.g_begin[pos + 1] - e0) : 0;
    int nv = has ? P.g_nv[pos] : 0;
    if constexpr (WAVE_ROW) { // (as in seg_slab_kernel: one unit per wave - scalar bounds, bases and row switches)
      e0 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(e0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)e0);
      len = __builtin_amdgcn_readfirstlane(len);
      nv = __builtin_amdgcn_readfirstlane(nv);
    }
    const int64_t v0 = has ? P.g_vrow0[pos] : 0;
    for (int l = 0; l < nv; ++l) {                       // the group's m1 rows (pieces of a split hub share their row)
      const int64_t row = P.v_row[v0 + l];
      rowV[(size_t)l * lpr + c] = (row >= 0 && row < p.K) ? *reinterpret_cast<const f4_t *>(m1 + row * p.F + c * VEC)   // (16 raw bytes)
                                                          : f4_t{0.f, 0.f, 0.f, 0.f};
    }
    int maxlen = len;
    if constexpr (!WAVE_ROW) {
      for (int o = 32; o >= lpr; o >>= 1) {
        const int other = __shfl_xor(maxlen, o, 64);
        maxlen = other > maxlen ? other : maxlen;
      }
    }
    int cur = 255;
    f4_t mrow[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) mrow[q] = f4_t{0.f, 0.f, 0.f, 0.f};
    int my_src = 0, my_dl = 255, my_pe = 0;
    {
      const bool valid = c < len;
      my_src = valid ? P.e_src[e0 + c] : 0;
      my_dl = valid ? (int)P.e_dl[e0 + c] : 255;
      my_pe = valid ? P.e_perm[e0 + c] : 0;
      if ((uint32_t)my_src >= src_rows) { my_src = 0; my_dl = 255; }   // out-of-range source: the dot is 0 (checked once per edge, here)
    }
    for (int off = 0; off < maxlen; off += lpr) {
      const bool nvalid = off + lpr + c < len;
      const int64_t ne = e0 + off + lpr + c;
      int n_src = nvalid ? P.e_src[ne] : 0;
      int n_dl = nvalid ? (int)P.e_dl[ne] : 255;
      const int n_pe = nvalid ? P.e_perm[ne] : 0;
      if ((uint32_t)n_src >= src_rows) { n_src = 0; n_dl = 255; }
      const int n_here = len - off;
      int n_max = maxlen - off;
      n_max = n_max < lpr ? n_max : lpr;
      for (int b = 0; b < n_max; b += kU) {
        if (p.window >= 0) {
          const int first_row = __builtin_amdgcn_readlane(my_src, b);
          const int has_edge = __builtin_amdgcn_readfirstlane(n_here) > b;
          if (has_edge) lock.at(p, lane, r * p.n_slabs + (first_row >> p.slab_shift));
        }
        f4_t v[kU];
        int dls[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {           // (slots behind the unit's last edge hold row 0 and dl = 255: the dot is 0; 32-bit row offsets)
          uint32_t row;
          if constexpr (WAVE_ROW) {
            row = (uint32_t)__builtin_amdgcn_readlane(my_src, b + u);
            dls[u] = __builtin_amdgcn_readlane(my_dl, b + u);
          } else {
            row = (uint32_t)__shfl(my_src, b + u, lpr);
            dls[u] = __shfl(my_dl, b + u, lpr);
          }
          if constexpr (WAVE_ROW) v[u] = slab_row_load<f4_t>(table, c16, row << rsh);
          else v[u] = *reinterpret_cast<const f4_t *>(m2 + (size_t)((row << rsh) + c16));
        }
        float pd[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          if (__builtin_expect(dls[u] != cur, 0)) {
            cur = dls[u];
            slab_unpack<T>(cur != 255 ? rowV[(size_t)cur * lpr + c] : f4_t{0.f, 0.f, 0.f, 0.f}, mrow);
          }
          f4_t x[NV];
          slab_unpack<T>(v[u], x);
          float dot = 0.f;
#pragma unroll
          for (int q = 0; q < NV; ++q) dot += x[q][0] * mrow[q][0] + x[q][1] * mrow[q][1] + x[q][2] * mrow[q][2] + x[q][3] * mrow[q][3];
          pd[u] = dot;
        }
        // 8 values x lpr lanes -> lane l < 8 of the unit holds the total of value 4*(l&1) + 2*((l>>1)&1) + ((l>>2)&1)
        const bool b0 = c & 1, b1 = c & 2, b2 = c & 4;
        float q[4], t2[2], s1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float keep = b0 ? pd[i + 4] : pd[i], give = b0 ? pd[i] : pd[i + 4];
          q[i] = keep + __shfl_xor(give, 1, 64);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float keep = b1 ? q[i + 2] : q[i], give = b1 ? q[i] : q[i + 2];
          t2[i] = keep + __shfl_xor(give, 2, 64);
        }
        {
          const float keep = b2 ? t2[1] : t2[0], give = b2 ? t2[0] : t2[1];
          s1 = keep + __shfl_xor(give, 4, 64);
        }
        // (multi-head rows [H, Fh] - the backward of mh_spmm: a dot product per head, reduced over the lph = lpr / H lanes of a head;
        //  the results of an edge are H neighbours: out[idx * H + h])
        for (int o = 8; o < lph; o <<= 1) s1 += __shfl_xor(s1, o, 64);
        const int j = 4 * (c & 1) + 2 * ((c >> 1) & 1) + ((c >> 2) & 1);
        const int pe = __shfl(my_pe, b + j, lpr);          // original edge id of the batch's j-th edge
        // staged form (geot_slab_sddmm_staged): the 8 results of a batch go to 8 consecutive plan positions - one 32-byte piece;
        // slab_unstage_kernel then brings them into edge order.  Straight to out[original edge id] every result is its own
        // partial write: 3.6 GB written for 0.46 GB of results at 115 M edges, 5.5 ms instead of 3.7 (profiles/r04/pmc_sddmm.txt)
        if ((c & (lph - 1)) < 8 && b + j < n_here)
          out[(p.w_in_plan_order ? e0 + off + b + j : (int64_t)pe) * p.H + (c >> lph_log2)] = (T)s1;
      }
      my_src = n_src;
      my_dl = n_dl;
      my_pe = n_pe;
    }
    lock.round_done(p, lane, r);
  }
  lock.leave(lane);
}

// The same SDDMM for rows of 512 / 256 bytes, ONE ROW PER WAVE-INSTRUCTION (E elements per lane; see seg_slab_wrow_kernel): the forward
// and the SDDMM of its backward share one plan, whose units are waves.  The dot products reduce over the 64 / H lanes of a head.
template <typename T, int E>
__global__ __launch_bounds__(kThreads) void seg_slab_sddmm_wrow_kernel(SlabParams p) {
  typedef typename SlabRaw<T, E>::type raw_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const geot_slab_plan &P = p.plan;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int R = P.rows_per_group;
  raw_t *rowV = reinterpret_cast<raw_t *>(smem) + (size_t)wave * R * 64;
  const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
  const int64_t units = P.units;
  const char *m2 = static_cast<const char *>(p.src);
  const T *m1 = static_cast<const T *>(p.weight);           // (the dst-side matrix travels in the `weight` slot)
  T *out = static_cast<T *>(p.dst);
  const uint32_t src_rows = (uint32_t)p.src_rows;
  const int rsh = 4 + p.lpr_log2;                            // log2(row bytes)
  const uint32_t cB = (uint32_t)lane * (uint32_t)(E * sizeof(T));
  const __amdgpu_buffer_rsrc_t table = slab_table_rsrc(p.src, p.src_rows, rsh, p.probe);
  int lph_log2 = 6;
  for (int hh = p.H; hh > 1; hh >>= 1) --lph_log2;
  const int lph = 1 << lph_log2;

  SlabStep lock;                               // the loose lockstep of the XCD's waves (see SlabStep)
  lock.enter(p, lane);
  auto raw_zero = [] {
    raw_t z;
    if constexpr (sizeof(raw_t) == 4) z = raw_t(0);
    else z = raw_t{0, 0};
    return z;
  };

  for (int r = 0; r < p.rounds; ++r) {
    const int64_t pos = (int64_t)r * units + ((r & 1) ? units - 1 - unit : unit);
    const bool has = pos < P.n_groups;
    int64_t e0 = has ? P.g_begin[pos] : 0;
    int len = has ? (int)(P.g_begin[pos + 1] - e0) : 0;
    int nv = has ? P.g_nv[pos] : 0;
    e0 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(e0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)e0);
    len = __builtin_amdgcn_readfirstlane(len);
    nv = __builtin_amdgcn_readfirstlane(nv);
    const int64_t v0 = has ? P.g_vrow0[pos] : 0;
    for (int l = 0; l < nv; ++l) {                       // the group's m1 rows (pieces of a split hub share their row)
      const int64_t row = P.v_row[v0 + l];
      rowV[(size_t)l * 64 + lane] = (row >= 0 && row < p.K) ? *reinterpret_cast<const raw_t *>(m1 + row * p.F + lane * E) : raw_zero();
    }
    int cur = 255;
    float mrow[E];
#pragma unroll
    for (int i = 0; i < E; ++i) mrow[i] = 0.f;
    int my_src = 0, my_dl = 255, my_pe = 0;
    {
      const bool valid = lane < len;
      my_src = valid ? P.e_src[e0 + lane] : 0;
      my_dl = valid ? (int)P.e_dl[e0 + lane] : 255;
      my_pe = valid ? P.e_perm[e0 + lane] : 0;
      if ((uint32_t)my_src >= src_rows) { my_src = 0; my_dl = 255; }   // out-of-range source: the dot is 0
    }
    for (int off = 0; off < len; off += 64) {
      const bool nvalid = off + 64 + lane < len;
      const int64_t ne = e0 + off + 64 + lane;
      int n_src = nvalid ? P.e_src[ne] : 0;
      int n_dl = nvalid ? (int)P.e_dl[ne] : 255;
      const int n_pe = nvalid ? P.e_perm[ne] : 0;
      if ((uint32_t)n_src >= src_rows) { n_src = 0; n_dl = 255; }
      const int n_here = len - off;
      const int n_max = n_here < 64 ? n_here : 64;
      for (int b = 0; b < n_max; b += kU) {
        if (p.window >= 0) lock.at(p, lane, r * p.n_slabs + (__builtin_amdgcn_readlane(my_src, b) >> p.slab_shift));
        raw_t v[kU];
        int dls[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {           // (slots behind the last edge hold row 0 and dl = 255: the dot is 0)
          const uint32_t row = (uint32_t)__builtin_amdgcn_readlane(my_src, b + u);
          dls[u] = __builtin_amdgcn_readlane(my_dl, b + u);
          v[u] = slab_row_load<raw_t>(table, cB, row << rsh);
        }
        float pd[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          if (__builtin_expect(dls[u] != cur, 0)) {
            cur = dls[u];
            if (cur != 255) mhrow_unpack<T, E>(rowV[(size_t)cur * 64 + lane], mrow);
            else {
#pragma unroll
              for (int i = 0; i < E; ++i) mrow[i] = 0.f;
            }
          }
          float x[E];
          mhrow_unpack<T, E>(v[u], x);
          float dot = 0.f;
#pragma unroll
          for (int i = 0; i < E; ++i) dot += x[i] * mrow[i];
          pd[u] = dot;
        }
        // 8 values x 64 lanes -> lane l with (l mod lph) < 8 holds the total of value 4*(l&1) + 2*((l>>1)&1) + ((l>>2)&1) of head l / lph
        const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
        float q[4], t2[2], s1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float keep = b0 ? pd[i + 4] : pd[i], give = b0 ? pd[i] : pd[i + 4];
          q[i] = keep + __shfl_xor(give, 1, 64);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float keep = b1 ? q[i + 2] : q[i], give = b1 ? q[i] : q[i + 2];
          t2[i] = keep + __shfl_xor(give, 2, 64);
        }
        {
          const float keep = b2 ? t2[1] : t2[0], give = b2 ? t2[0] : t2[1];
          s1 = keep + __shfl_xor(give, 4, 64);
        }
        for (int o = 8; o < lph; o <<= 1) s1 += __shfl_xor(s1, o, 64);
        const int j = 4 * (lane & 1) + 2 * ((lane >> 1) & 1) + ((lane >> 2) & 1);
        const int pe = __shfl(my_pe, b + j, 64);
        if ((lane & (lph - 1)) < 8 && b + j < n_here)
          out[(p.w_in_plan_order ? e0 + off + b + j : (int64_t)pe) * p.H + (lane >> lph_log2)] = (T)s1;
      }
      my_src = n_src;
      my_dl = n_dl;
      my_pe = n_pe;
    }
    lock.round_done(p, lane, r);
  }
  lock.leave(lane);
}

// a compile-time loop (the DPP controls below are instruction immediates) and the row broadcast: every lane of each 16-lane row
// receives lane C of ITS row (DPP row_newbcast, gfx90a+): a vector-ALU move, no LDS crossbar
template <class F, int... J>
__device__ __forceinline__ void slab_static_for(F &&f, std::integer_sequence<int, J...>) { (f(std::integral_constant<int, J>{}), ...); }
template <int C>
__device__ __forceinline__ uint32_t slab_row_bcast(uint32_t v) {
  static_assert(C >= 0 && C < 16, "a lane of the row");
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + C, 0xF, 0xF, false);
}

// The multi-head SDDMM of 16-bit plans on the MATRIX cores (round 5): out[e, h] = <m1[dst(e), h, :], m2[src(e), h, :]> is a
// contraction over FEATURES, and that is the K dimension of an MFMA with both operands in the layout memory already has them in:
// v_mfma_f32_16x16x32_{bf16,f16} takes A[16 rows][32 features] with lane l holding 8 consecutive features (16 bytes) of row l & 15
// - a piece of one of the group's <= 16 m1 rows, which stay in registers for the whole group - and B[32 features][16 edges] with lane
// l holding 8 consecutive features of edge l & 15 - a 16-byte piece of a gathered m2 row.  One instruction forms all 16 x 16
// (row, edge) products of a 32-feature slice; the one wanted per edge, D[dl(n)][n], lies in ONE lane's result registers and that lane
// stores it (round 6; round 5 had the edges on the M side and picked the products up through an LDS exchange), the other 15 are the
// price.  It is worth it because these kernels are bound by vector-instruction issue, not by their gathers (§3.1d
// of DESIGN.md: with the gathers dropped they keep 97 % of their time): the row-per-wave kernel spends ~20 wave-instructions per edge
// on unpacking two 16-bit rows and multiplying them, this one ~8, and the MFMAs themselves are 0.4 ms of work at Reddit scale.
// Nothing is contaminated by the unused products: a result element is the sum over its OWN row's and edge's features only.
// The m2 rows are brought in WHOLE (one edge per gather instruction, 8 bytes a lane - the row-per-wave kernel's requests: four whole
// 128-byte lines) and turned into A fragments through LDS: a tile's 16 rows are written to a per-wave image [16 edges][512 + 32
// bytes] (ds_write_b64, contiguous) and read back as the operand map wants them (ds_read_b128 at [edge n][64 c + 16 kb]; the
// 32-byte pad makes every 16-lane group of the read hit 64 distinct banks).  (Gathering the operand pieces DIRECTLY - four lanes
// per edge, 16 rows x 64 bytes per instruction - was built first and measured slower than the row-per-wave kernel, 7.33 vs 5.84 ms,
// though it needed only 3.73 ms with its gathers dropped: two 64-byte requests per 128-byte line, eight per edge instead of four.)
// MEASURED at Reddit scale, bf16 H=4 x F=64: **4.30 ms** with the results left in plan order against 5.84 (5.25 vs 6.81 into edge
// order), exact on integer data for every shape and both types (tests/test_gpu_round5.py); 4.18 ms with the gathers dropped - what
// bounds it now is the LDS write path (a tile's image is 8 KB at ~80 B/clk per CU) and the chain of waits inside a tile
// (profiles/r05/slab_cases__mh_sddmm_matrix_cores_ab.txt).  Plans cut into waves, R <= 16 rows per group, rows of 512 bytes, CPH
// 32-feature slices per head (H = 8 / CPH heads), results in the plan's edge order
// (staged); fp32 accumulation as everywhere, the sum order inside a head is the hardware's (32 features per step, CPH steps).
// Option "slab_sddmm_mfma" = 0: the row-per-wave kernel.  Two refinements measured afterwards (one box, interleaved): the H exchange reads
// made unconditional (behind a per-head branch each waited for its own LDS round trip): 4.31 -> 4.24 ms, kept; a tile's rows gathered ONE
// TILE AHEAD (into the same registers, while the previous tile goes through the matrix cores): 4.47 vs 4.28 ms - slower (130 registers
// instead of 108, and the gathers were not what a tile waits for), dropped.  Round 6: two rows per gather instruction with the next
// tile's rows in flight 4.24 -> 3.81 ms; the operands exchanged (no result exchange through LDS) 3.81 -> 3.39 ms
// (profiles/r06/slab_cases__mh_bf16_sddmm_*.txt).
template <typename T, int CPH, int ROWB>
__global__ __launch_bounds__(kThreads, 3) void seg_slab_sddmm_mfma_kernel(SlabParams p) {   // (3 waves per SIMD: the persistent grid's 3 workgroups per CU all resident)
  static_assert(ROWB == 1024 || ROWB == 512 || ROWB == 256, "rows of 1024, 512 or 256 bytes");
  // rows of 1 KiB: TWO PASSES over the group's edges, each the 512-byte form on one half of every row - the first half of the heads,
  // then the others (see seg_slab_spmm_mfma_kernel; a head never straddles the halves: at least two heads)
  constexpr int PASSES = ROWB == 1024 ? 2 : 1;
  constexpr int PBYTES = ROWB / PASSES;                  // bytes of a row a pass works on: 512 | 512 | 256
  constexpr int NCH = PBYTES / 64;                       // 32-feature slices of a pass: 8 | 8 | 4
  static_assert(sizeof(T) == 2 && NCH % CPH == 0, "16-bit rows, a head is whole 32-feature slices");
  constexpr int HP = NCH / CPH, H = HP * PASSES;         // heads of a pass, heads of a row
  constexpr int LOGB = ROWB == 1024 ? 10 : (ROWB == 512 ? 9 : 8);
  constexpr int KS = 512 / PBYTES;                       // 16-edge blocks (the N side of a product) per tile: 1 | 2
  constexpr int KT = 16 * KS;                            // edges per tile: 16 | 32 - eight gather instructions, 8 KB, either way
  constexpr int PH = 64 / KT;                            // tiles per 64-edge chunk: 4 | 2
  constexpr int LPR = PBYTES / 16, RPI = 64 / LPR;         // lanes per row of a gather instruction, rows per instruction: 32, 2 | 16, 4
  constexpr int kStride = PBYTES + 32;                     // bytes between the rows of a tile's image
  typedef T t8_t __attribute__((ext_vector_type(8)));
  typedef uint32_t raw2_t __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const geot_slab_plan &P = p.plan;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, kb = lane >> 4;
  // this wave's LDS: the tile image [KT][kStride]; a wave's LDS operations execute in order, the fences keep the compiler's
  unsigned char *img = smem + (size_t)wave * (KT * kStride);
  const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
  const int64_t units = P.units;
  const T *m1 = static_cast<const T *>(p.weight);
  T *out = static_cast<T *>(p.dst);
  const uint32_t src_rows = (uint32_t)p.src_rows;
  const __amdgpu_buffer_rsrc_t table = slab_table_rsrc(p.src, p.src_rows, LOGB, p.probe);
  typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
  const uint32_t cH = (uint32_t)(lane % LPR) * 16u;       // a lane's 16 bytes of its lane group's row
  const uint32_t po_idx = (uint32_t)(KT * (n >> 3) + RPI * (n & 7) + (kb * RPI) / 4) << 2;   // (the broadcast register's layout, see the gathers)
  t8_t zero8;
#pragma unroll
  for (int i = 0; i < 8; ++i) zero8[i] = (T)0.f;

  SlabStep lock;                               // the loose lockstep of the XCD's waves (see SlabStep)
  lock.enter(p, lane);
  auto wave_order = [] {                        // LDS words written by some lanes and read by others: keep the compiler's order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  for (int r = 0; r < p.rounds; ++r) {
    const int64_t pos = (int64_t)r * units + ((r & 1) ? units - 1 - unit : unit);
    const bool has = pos < P.n_groups;
    int64_t e0 = has ? P.g_begin[pos] : 0;
    int len = has ? (int)(P.g_begin[pos + 1] - e0) : 0;
    int nv = has ? P.g_nv[pos] : 0;
    e0 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(e0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)e0);
    len = __builtin_amdgcn_readfirstlane(len);
    nv = __builtin_amdgcn_readfirstlane(nv);
    const int64_t v0 = has ? P.g_vrow0[pos] : 0;
    for (int pass = 0; pass < PASSES; ++pass) {
    const uint32_t cP = cH + (uint32_t)pass * PBYTES;       // this lane's 16 bytes inside the table's row
    const int vr = r * PASSES + pass;           // the lockstep's round: a pass sweeps the slabs once
    t8_t bfrag[NCH];                            // the group's m1 rows as operand fragments, in registers for the whole group
    {
      const int64_t row = n < nv ? P.v_row[v0 + n] : -1;
      const bool ok = row >= 0 && row < p.K;
      const T *rp = m1 + (ok ? row : 0) * p.F + pass * (PBYTES / 2) + 8 * kb;
#pragma unroll
      for (int c = 0; c < NCH; ++c) bfrag[c] = ok ? *reinterpret_cast<const t8_t *>(rp + 32 * c) : zero8;
    }
    uint32_t my_edge = 255;                     // (source row << 8) | row in group; 255 = padding / out-of-range source
    {
      const bool valid = lane < len;
      const uint32_t s_ = valid ? (uint32_t)P.e_src[e0 + lane] : 0u;
      const uint32_t d_ = valid ? (uint32_t)P.e_dl[e0 + lane] : 255u;
      my_edge = s_ < src_rows ? ((s_ << 8) | d_) : 255u;
    }
    // Tiles of 16 | 32 edges, the tiles of a 64-edge chunk unrolled.  SEVERAL rows per gather instruction, 16 bytes a lane (512-byte
    // rows: lanes 0..31 the row of edge 2 j, lanes 32..63 that of edge 2 j + 1; 256-byte rows: 16 lanes an edge): a wave-wide load costs
    // the CU ~17-20 cycles whatever its width (tools/kexp5.hip) - one 512-byte row per instruction was this kernel's floor (4.18 of
    // its 4.30 ms with the gathers dropped, round 5).  The NEXT tile's rows are gathered as soon as the current tile is in the image:
    // in flight under its matrix work and its stores.  The per-lane row offsets: one crossbar read per TWO tiles lays their edges'
    // offsets out for DPP row broadcasts (lane 8 tt + j of row r: what row r's lanes fetch in instruction j of tile tt - see
    // seg_slab_spmm_mfma_kernel).
    const int ntiles = (len + KT - 1) / KT;
    uint32_t nx_src = 0, nx_dl = 255;
    bool nx_valid = false;
    u4_t rv[8];
    uint32_t po = 0;
    auto pair_up = [&](int gen) __attribute__((always_inline)) {
      po = (uint32_t)__builtin_amdgcn_ds_bpermute((int)po_idx + gen * (2 * KT * 4), (int)((my_edge & ~255u) << (LOGB - 8)));
    };
    auto gather = [&](auto tb_c) __attribute__((always_inline)) {
      constexpr int tb = decltype(tb_c)::value, k0 = ((tb / KT) & 1) * 8;
      if (p.window >= 0) lock.at(p, lane, vr * p.n_slabs + (int)((uint32_t)__builtin_amdgcn_readlane(my_edge, tb) >> (8 + p.slab_shift)));
      slab_static_for([&](auto j_c) __attribute__((always_inline)) {
        constexpr int j = decltype(j_c)::value;
        rv[j] = slab_row_load<u4_t>(table, cP + slab_row_bcast<k0 + j>(po), 0u);
      }, std::make_integer_sequence<int, 8>{});
    };
    auto tile = [&](int t, auto ph_c) __attribute__((always_inline)) {
      constexpr int ph = decltype(ph_c)::value, tb = ph * KT, tbn = ((ph + 1) % PH) * KT;
      const bool more = t + 1 < ntiles;
      if constexpr (ph == 0) {                  // the next chunk's fields, behind this tile's gathers (every lane loads: see seg_slab_wrow_kernel)
        const int c = t / PH;
        nx_valid = (c + 1) * 64 + lane < len;
        const int64_t ne = e0 + (nx_valid ? (c + 1) * 64 + lane : len - 1);
        nx_src = (uint32_t)P.e_src[ne];
        nx_dl = (uint32_t)P.e_dl[ne];
      }
      int dlm[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) dlm[ks] = (int)((uint32_t)__builtin_amdgcn_ds_bpermute((tb + 16 * ks + n) << 2, (int)my_edge) & 255u);
      if constexpr (ph == PH - 1) {             // the chunk ends with this tile: the next chunk's edges become this lane's
        my_edge = (nx_valid && nx_src < src_rows) ? ((nx_src << 8) | nx_dl) : 255u;
        pair_up(0);
      } else if constexpr ((ph & 1) == 1) pair_up((ph + 1) / 2);
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<u4_t *>(img + (RPI * j + lane / LPR) * kStride + cH) = rv[j];
      wave_order();
      if (more) gather(std::integral_constant<int, tbn>{});
      // The group's rows are the M side of the product and the tile's edges the N side (both operands have the same lane map, so this is
      // only the order of the arguments): lane 16 kb + n then holds D[row 4 kb + j][edge n], j = 0..3 - the four lanes n, 16 + n, 32 + n,
      // 48 + n own all sixteen (row, edge n) products between them, and edge n's one wanted product, row dl(n), sits in lane
      // 16 (dl >> 2) + n, element dl & 3.  That lane stores it: no exchange through LDS, nothing in the image for the next tile to wait
      // for but the fragment reads (the first form - edges as M, D written to LDS and picked up by lanes 0..15 - cost two more fences
      // and 2 H LDS operations per tile, and kept the image busy until the pick-up: 3.81 ms at configs[3]'s graph).
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        t8_t afrag[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) afrag[c] = *reinterpret_cast<const t8_t *>(img + (16 * ks + n) * kStride + 64 * c + 16 * kb);
        float vals[HP];
#pragma unroll
        for (int h = 0; h < HP; ++h) {
          f4_t d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int cc = 0; cc < CPH; ++cc) {
            if constexpr (__is_same(T, bf16_t)) d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfrag[h * CPH + cc], afrag[h * CPH + cc], d, 0, 0, 0);
            else d = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfrag[h * CPH + cc], afrag[h * CPH + cc], d, 0, 0, 0);
          }
          const float lo = (dlm[ks] & 1) ? d[1] : d[0], hi = (dlm[ks] & 1) ? d[3] : d[2];
          vals[h] = (dlm[ks] & 2) ? hi : lo;
        }
        const int e = t * KT + 16 * ks + n;
        const bool real = dlm[ks] != 255;                               // (padding / out-of-range source: a zero, written by the kb = 0 lane)
        if (e < len && (real ? (dlm[ks] >> 2) == kb : kb == 0)) {
          T *op = out + (e0 + e) * H + pass * HP;
          if constexpr (HP == 1) op[0] = (T)(real ? vals[0] : 0.f);
          else {
            typedef T tH_t __attribute__((ext_vector_type(HP)));
            tH_t pk;
#pragma unroll
            for (int h = 0; h < HP; ++h) pk[h] = (T)(real ? vals[h] : 0.f);
            *reinterpret_cast<tH_t *>(op) = pk;
          }
        }
      }
      wave_order();
    };
    pair_up(0);
    if (ntiles > 0) gather(std::integral_constant<int, 0>{});
    for (int t = 0; t < ntiles; t += PH) {
      tile(t, std::integral_constant<int, 0>{});
      if (t + 1 < ntiles) tile(t + 1, std::integral_constant<int, 1>{});
      if constexpr (PH == 4) {
        if (t + 2 < ntiles) tile(t + 2, std::integral_constant<int, 2>{});
        if (t + 3 < ntiles) tile(t + 3, std::integral_constant<int, 3>{});
      }
    }
    lock.round_done(p, lane, vr);
    }                                           // pass
  }
  lock.leave(lane);
}

// ---- the 16-bit multi-head SpMM on the MATRIX cores (round 6) ---------------------------------------------------------------------
// out[d, h, :] = sum_e w[e, h] x[s_e, h, :] contracts over EDGES, and an edge's row lies in memory along the FEATURES: the operand the
// MFMA wants k-major has to be transposed on the way - gfx950's ds_read_b64_tr_b16 does that for free out of a row-major LDS image.
// Per step of 16 edges of a group (<= 16 output rows) and per 16-feature block fb:
//     D_fb[16 rows x 16 features] += A_h[16 rows x 16 edges] * B_fb[16 edges x 16 features]        (v_mfma_f32_16x16x16_{bf16,f16})
//   A_h[m][k] = dl(k) == m ? w[k, h] : 0   - the selector of the edge's output row times its weight, built in registers from the
//               step's row-in-group bytes and weights (staged in LDS per 64-edge chunk): 2 packed masks per step + 2 ANDs per head;
//   B_fb[k][n] = x[src(k), 16 fb + n]      - the gathered rows, written whole to a per-wave image [edges][row bytes + 32]
//               (ds_write_b128, several edges per instruction as they were gathered) and read back transposed, one ds_read_b64_tr_b16
//               per block (image rows 4 kq .. 4 kq + 3: with the 32-byte pad the rows a 16-lane group reads lie on distinct banks);
//   D_fb stays in registers for the WHOLE group (16 | 8 blocks x 4 VGPRs): no LDS accumulators, no row switches, no unpacking, no
//               per-edge FMA stream - the ~20 wave-instructions per edge of seg_slab_wrow_kernel become ~9.
// The unused 15/16 of every product is the price (the MFMAs are ~0.75 ms of matrix-core time at Reddit scale, under the gathers).  IEEE
// isolation: a zero of A times an Inf / NaN of ANOTHER row's source would put a NaN into this row.  So the table is checked first
// (slab_nonfinite_kernel: one streamed read of x, ~0.02 ms for 119 MB) and the launch is GATED on the result: with a non-finite value
// anywhere in x this kernel returns at once and its vector-ALU twin (seg_slab_wrow_kernel, enqueued behind it with the opposite gate)
// does the call - same plan, same results as before.  Weights may be anything (an Inf weight only meets its own row's products).
// Order of additions inside a row: the hardware's (16 edges per step in plan order); fixed by the plan, not by timing.
// Plans cut into waves (units = waves: 3 workgroups per CU - image + staged weights are ~10 KB a wave), R <= 16, rows of 512 or 256
// bytes (1 KiB: two passes, at least two heads), heads 1 / 2 / 4 / 8 with feat % 16 == 0; WMODE 0 none | 1 weight[e] | 2 weight[e * H + h] | 3 weight[h * nnz + e];
// p.w_in_plan_order as in seg_slab_wrow_kernel.  Option "slab_spmm_mfma" = 0: the row-per-wave kernel.
template <typename T>
__global__ __launch_bounds__(kThreads) void slab_nonfinite_kernel(const uint32_t *__restrict__ x, int64_t n16, int *flag) {
  typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
  const u4_t *p = reinterpret_cast<const u4_t *>(x);
  // an element is Inf / NaN iff its exponent field is all ones: bf16 0x7F80, f16 0x7C00 (per halfword)
  constexpr uint32_t kExp = __is_same(T, bf16_t) ? 0x7F807F80u : 0x7C007C00u, kInc = __is_same(T, bf16_t) ? 0x00800080u : 0x04000400u;
  uint32_t bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n16; i += (int64_t)gridDim.x * kThreads) {
    const u4_t v = __builtin_nontemporal_load(p + i);
#pragma unroll
    for (int q = 0; q < 4; ++q) bad |= ((v[q] & kExp) + kInc) & 0x80008000u;   // (0x7F80 + 0x0080 = 0x8000: bit 15 / 31 of each half, no carry across)
  }
  if (__builtin_amdgcn_ballot_w64(bad != 0) != 0 && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// ROWB = 512: tiles of 16 edges (one K = 16 step), two rows per gather instruction.  ROWB = 256: tiles of 32 edges (two K = 16 steps
// over one image), four rows per gather instruction - the same eight instructions, 8 KB, in flight per tile and the same image size.
// (A PAIR of waves per group of 512-byte rows, each one half of the features with 4 workgroups per CU, was built and measured slower,
// 3.87 against 3.71 ms, and is gone: profiles/r06/slab_cases__mh_bf16_spmm_matrix_cores_ab__product_library.txt.)
template <typename T, int H, int WMODE, int ROWB, int RED = GEOT_REDUCE_SUM>
__global__ __launch_bounds__(kThreads, 3) void seg_slab_spmm_mfma_kernel(SlabParams p) {
  static_assert(sizeof(T) == 2 && (H == 1 || H == 2 || H == 4 || H == 8), "16-bit rows, 1 / 2 / 4 / 8 heads");
  static_assert(RED == GEOT_REDUCE_SUM || (RED == GEOT_REDUCE_MEAN && H == 1), "sums; the mean of a single-head aggregation");
  static_assert(WMODE >= 0 && WMODE <= 3 && (WMODE != 1 || H == 1), "one weight per edge = one head");
  static_assert(ROWB == 1024 || ROWB == 512 || ROWB == 256, "rows of 1024, 512 or 256 bytes");
  static_assert(ROWB != 1024 || (H >= 2 && RED == GEOT_REDUCE_SUM), "1-KiB rows: two passes of whole heads");
  // rows of 1 KiB: TWO PASSES over the group's edges, each the 512-byte form on one half of every row (the first H / 2 heads, then the
  // others) - 512 features x 16 rows of fp32 accumulators are 128 registers a lane, a pass holds 64.  The edge fields and weights are
  // walked twice (13 bytes an edge against 1 KiB of row), the rows' bytes are gathered once each as before.
  constexpr int PASSES = ROWB == 1024 ? 2 : 1;
  constexpr int PBYTES = ROWB / PASSES;                  // bytes of a row a pass works on: 512 | 512 | 256
  constexpr int HP = H / PASSES;                         // heads of a pass
  constexpr int LOGB = ROWB == 1024 ? 10 : (ROWB == 512 ? 9 : 8);
  constexpr int KS = 512 / PBYTES;                       // K = 16 steps per tile: 1 | 2
  constexpr int KT = 16 * KS;                            // edges per tile: 16 | 32
  constexpr int PH = 64 / KT;                            // tiles per 64-edge chunk: 4 | 2
  constexpr int LPR = PBYTES / 16;                         // lanes per row of a gather instruction (16 bytes a lane): 32 | 16
  constexpr int RPI = 64 / LPR;                          // rows per gather instruction: 2 | 4
  constexpr int NL = KT / RPI;                           // gather instructions per tile: 8
  constexpr int TPR = 16 / NL;                           // tiles whose row offsets one broadcast register holds: 2
  constexpr int NFB = PBYTES / 32;                         // 16-feature blocks of a row: 16 | 8
  constexpr int kStride = PBYTES + 32;                     // bytes between the rows of a tile's image
  constexpr int kImg = KT * kStride;                     // 8 704 | 9 216 bytes
  constexpr int kWave = kImg + 2 * H * 64 * 2 + 2 * 64;  // + weights [2 chunks][H][64] of T + rows-in-group [2][64] bytes
  constexpr int FB_PER_H = NFB / HP;                     // 16-feature blocks per head
  static_assert(NL == 8 && TPR == 2 && FB_PER_H >= 1, "eight gather instructions per tile; a head is whole 16-feature blocks");
  typedef short s4_t __attribute__((ext_vector_type(4)));
  typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
  typedef uint32_t raw2_t __attribute__((ext_vector_type(2)));
  typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s4_t *lds_s4_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (p.gate && ((slab_gate_word(p.gate) != 0) != (p.gate_want != 0))) return;
  const geot_slab_plan &P = p.plan;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, kq = lane >> 4;
  unsigned char *img = smem + (size_t)wave * kWave;
  uint16_t *wst = reinterpret_cast<uint16_t *>(img + kImg);          // [2][H][64]
  unsigned char *dlb = img + kImg + 2 * H * 64 * 2;                   // [2][64]
  float *imgf = reinterpret_cast<float *>(img);                       // (group end: eight D tiles as [16 rows][128] fp32: 8 KB of the image)
  const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
  const int64_t units = P.units;
  const T *weight = static_cast<const T *>(p.weight);
  T *dst = static_cast<T *>(p.dst);
  const bool wpo = p.w_in_plan_order != 0;
  const uint32_t src_rows = (uint32_t)p.src_rows;
  const __amdgpu_buffer_rsrc_t table = slab_table_rsrc(p.src, p.src_rows, LOGB, p.probe);
  const uint32_t cL = (uint32_t)(lane % LPR) * 16u;       // a lane's 16 bytes inside the row (table and image)
  // the tile's transposed read: lane 16 kq + 4 q + pp supplies the address of image row 4 kq + q, columns 4 pp .. 4 pp + 3 of the block;
  // lane 16 kq + n receives column n of those four rows = B[k = 4 kq .. 4 kq + 3][n]
  const uint32_t tr_off = (uint32_t)((4 * kq + ((lane >> 2) & 3)) * kStride + (lane & 3) * 8);
  constexpr uint32_t kOne = __is_same(T, bf16_t) ? 0x3F803F80u : 0x3C003C00u;   // (1, 1) in the storage type
  // the broadcast register's layout (see the gathers): lane 16 r + k reads the chunk's lane KT (k / NL) + RPI (k % NL) + r RPI / 4
  const uint32_t po_idx = (uint32_t)(KT * (m / NL) + RPI * (m % NL) + (kq * RPI) / 4) << 2;

  SlabStep lock;                               // the loose lockstep of the XCD's waves (see SlabStep)
  lock.enter(p, lane);
  auto wave_order = [] {                        // LDS words written by some lanes and read by others: keep the compiler's order (the
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // hardware runs a wave's LDS operations in order; no wait is emitted)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // one edge's weights as loaded (raw 16-bit words; never converted - they go to the matrix cores as they are)
  struct WRaw { uint16_t v[H]; };
  auto load_w = [&](int64_t pe) {
    WRaw w;
    const uint16_t *wp = reinterpret_cast<const uint16_t *>(weight);
    if constexpr (WMODE == 0) {
#pragma unroll
      for (int h = 0; h < H; ++h) w.v[h] = (uint16_t)(kOne & 0xFFFFu);
    } else if constexpr (WMODE == 1) w.v[0] = wp[pe];
    else if constexpr (WMODE == 2) {
      if constexpr (H == 1) w.v[0] = wp[pe];
      else if constexpr (H == 2) {
        const uint32_t x = *reinterpret_cast<const uint32_t *>(wp + pe * 2);
        w.v[0] = (uint16_t)x, w.v[1] = (uint16_t)(x >> 16);
      } else if constexpr (H == 4) {
        const raw2_t x = *reinterpret_cast<const raw2_t *>(wp + pe * 4);
        w.v[0] = (uint16_t)x[0], w.v[1] = (uint16_t)(x[0] >> 16), w.v[2] = (uint16_t)x[1], w.v[3] = (uint16_t)(x[1] >> 16);
      } else {
        const u4_t x = *reinterpret_cast<const u4_t *>(wp + pe * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) w.v[2 * q] = (uint16_t)x[q], w.v[2 * q + 1] = (uint16_t)(x[q] >> 16);
      }
    } else {
#pragma unroll
      for (int h = 0; h < H; ++h) w.v[h] = wp[(int64_t)h * P.nnz + pe];
    }
    return w;
  };
  auto stage = [&](int buf, const WRaw &w, bool valid, uint32_t edge) {     // this lane's edge of a chunk into the chunk's LDS arrays
#pragma unroll
    for (int h = 0; h < H; ++h) wst[(buf * H + h) * 64 + lane] = valid ? w.v[h] : (uint16_t)0;
    dlb[buf * 64 + lane] = (unsigned char)(edge & 255u);
  };

  for (int r = 0; r < p.rounds; ++r) {
    const int64_t pos = (int64_t)r * units + ((r & 1) ? units - 1 - unit : unit);
    const bool has = pos < P.n_groups;
    int64_t e0 = has ? P.g_begin[pos] : 0;
    int len = has ? (int)(P.g_begin[pos + 1] - e0) : 0;
    int nv = has ? P.g_nv[pos] : 0;
    e0 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(e0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)e0);
    len = __builtin_amdgcn_readfirstlane(len);
    nv = __builtin_amdgcn_readfirstlane(nv);
    for (int pass = 0; pass < PASSES; ++pass) {
    const int hbase = pass * HP;                // this pass's first head ...
    const uint32_t cP = cL + (uint32_t)pass * PBYTES;   // ... and this lane's 16 bytes inside the table's row
    const int vr = r * PASSES + pass;           // the lockstep's round: a pass sweeps the slabs once
    f4_t D[NFB];
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) D[fb] = f4_t{0.f, 0.f, 0.f, 0.f};
    uint32_t my_edge = 255;                     // (source row << 8) | row in group; 255 = padding / out-of-range source (contributes nothing)
    {
      const bool valid = lane < len;
      const uint32_t s_ = valid ? (uint32_t)P.e_src[e0 + lane] : 0u;
      const uint32_t d_ = valid ? (uint32_t)P.e_dl[e0 + lane] : 255u;
      my_edge = s_ < src_rows ? ((s_ << 8) | d_) : 255u;
      int64_t pe = 0;
      if constexpr (WMODE != 0) pe = valid ? (wpo ? e0 + lane : (int64_t)P.e_perm[e0 + lane]) : 0;
      stage(0, load_w(pe), valid, my_edge);
    }
    wave_order();
    const int ntiles = (len + KT - 1) / KT;
    uint32_t nx_src = 0, nx_dl = 255, nx_pe = 0;        // the next chunk's fields: loaded under the chunk's first tile, handed over under its last
    WRaw nx_w = load_w(0);
    bool nx_valid = false;
    uint32_t my_off = (my_edge & ~255u) << (LOGB - 8);  // this lane's edge: its source row's byte offset
    // SEVERAL rows per gather instruction, 16 bytes a lane (512-byte rows: lanes 0..31 the row of edge 2 j, lanes 32..63 that of edge
    // 2 j + 1; 256-byte rows: 16 lanes an edge, 4 j .. 4 j + 3): a wave-wide load costs the CU ~17-20 cycles whatever its width
    // (tools/kexp5.hip: dropped by the range check 16 / 18 cycles at 8 / 16 bytes a lane, from an L2-resident table 20 / 19) - at one
    // 512-byte row per instruction that alone is 3.8 ms for configs[3]'s 114.6 M edges, the floor every row-per-wave kernel of this file
    // sits on.
    // The per-lane row offsets: ONE crossbar read (ds_bpermute) lays the offsets of two tiles' edges out so that lane k of every 16-lane
    // row holds what that row's lanes need for one gather instruction (row r, lane k = 8 tt + j: the edge of tile tt, instruction j,
    // that row r's lanes fetch); the instruction's offsets are then a DPP row broadcast of lane k - vector-ALU work, folded into the add
    // of the lane's column.  (First form: one ds_bpermute per gather instruction, 8 of a tile's 38 LDS operations: 3.72 -> 3.45 ms.)
    u4_t rv[NL];
    uint32_t po = 0;
    auto pair_up = [&](int gen) __attribute__((always_inline)) {
      po = (uint32_t)__builtin_amdgcn_ds_bpermute((int)po_idx + gen * (TPR * KT * 4), (int)my_off);
    };
    auto gather = [&](auto tb_c) __attribute__((always_inline)) {         // a tile's rows (slots behind the last edge: row 0)
      constexpr int tb = decltype(tb_c)::value, k0 = ((tb / KT) % TPR) * NL;
      if (p.window >= 0) lock.at(p, lane, vr * p.n_slabs + (int)((uint32_t)__builtin_amdgcn_readlane(my_off, tb) >> (LOGB + p.slab_shift)));
      slab_static_for([&](auto j_c) __attribute__((always_inline)) {
        constexpr int j = decltype(j_c)::value;
        rv[j] = slab_row_load<u4_t>(table, cP + slab_row_bcast<k0 + j>(po), 0u);
      }, std::make_integer_sequence<int, NL>{});
    };
    auto tile = [&](int t, auto ph_c) __attribute__((always_inline)) {
      constexpr int ph = decltype(ph_c)::value, tb = ph * KT, tbn = ((ph + 1) % PH) * KT;
      const int c = t / PH, buf = c & 1;
      const bool more = t + 1 < ntiles;
      // the next chunk's fields (and its weights, where they lie in plan order), behind this tile's gathers: every lane loads (lanes
      // behind the group's end re-read its last edge and drop the value)
      if constexpr (ph == 0) {
        nx_valid = (c + 1) * 64 + lane < len;
        const int64_t ne = e0 + (nx_valid ? (c + 1) * 64 + lane : len - 1);
        nx_src = (uint32_t)P.e_src[ne];
        nx_dl = (uint32_t)P.e_dl[ne];
        if constexpr (WMODE != 0) {
          if (wpo) nx_w = load_w(ne);
          else nx_pe = (uint32_t)P.e_perm[ne];
        }
      }
      if constexpr (WMODE != 0 && PH == 4 && ph == 2) {
        if (!wpo) nx_w = load_w((int64_t)nx_pe);       // (through the permutation: its entry arrived two tiles ago)
      }
      // Everything that needs only the LDS crossbar / the chunk's staged arrays goes FIRST, ahead of the wait for this tile's rows: the
      // hand-over of the next chunk (its fields arrived tiles ago), the next tile's gather offsets, this tile's selector bytes and weights
      if constexpr (ph == PH - 1) {             // the chunk ends with this tile: the next chunk's edges become this lane's
        my_edge = (nx_valid && nx_src < src_rows) ? ((nx_src << 8) | nx_dl) : 255u;
        my_off = (my_edge & ~255u) << (LOGB - 8);
        stage(buf ^ 1, nx_w, nx_valid, my_edge);
        pair_up(0);
      } else if constexpr ((ph + 1) % TPR == 0) pair_up((ph + 1) / TPR);      // (the next tile opens the register's next generation)
      uint32_t dl4[KS];
      raw2_t wa[KS][HP];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        dl4[ks] = *reinterpret_cast<const uint32_t *>(dlb + buf * 64 + tb + 16 * ks + 4 * kq);
#pragma unroll
        for (int h = 0; h < HP; ++h) wa[ks][h] = *reinterpret_cast<const raw2_t *>(wst + (buf * H + hbase + h) * 64 + tb + 16 * ks + 4 * kq);
      }
      // tile t into the image (the previous tile's transposed reads are ahead of these writes in the wave's LDS queue)
#ifdef GEOT_DEV_EXPERIMENTS
      if (!(p.probe & 4))
#endif
      {
#pragma unroll
        for (int j = 0; j < NL; ++j) *reinterpret_cast<u4_t *>(img + (RPI * j + lane / LPR) * kStride + cL) = rv[j];
      }
#ifdef GEOT_DEV_EXPERIMENTS
      if (p.probe & 4) {                        // (knock-out experiment "slab_probe" bit 2: no image writes - the rows still have to arrive)
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < NL; ++j) acc |= rv[j][0] ^ rv[j][1] ^ rv[j][2] ^ rv[j][3];
        if (acc == 0x12345678u) img[lane] = 1;
      }
#endif
      wave_order();
      if (more) gather(std::integral_constant<int, tbn>{});              // in flight under this tile's matrix work
#ifdef GEOT_DEV_EXPERIMENTS
      const bool no_tr = (p.probe & 8) != 0;   // (knock-out bit 3: the MFMAs without their LDS operand)
#endif
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        // A: selector x weight.  This lane's four edges are the image rows 16 ks + 4 kq .. + 3 of the tile
        uint32_t mk[2];
        mk[0] = (((dl4[ks] & 255u) == (uint32_t)m) ? 0xFFFFu : 0u) | ((((dl4[ks] >> 8) & 255u) == (uint32_t)m) ? 0xFFFF0000u : 0u);
        mk[1] = ((((dl4[ks] >> 16) & 255u) == (uint32_t)m) ? 0xFFFFu : 0u) | (((dl4[ks] >> 24) == (uint32_t)m) ? 0xFFFF0000u : 0u);
        s4_t afrag[HP];
#pragma unroll
        for (int h = 0; h < HP; ++h) {
          const raw2_t a = {wa[ks][h][0] & mk[0], wa[ks][h][1] & mk[1]};
          afrag[h] = __builtin_bit_cast(s4_t, a);
        }
#ifdef GEOT_DEV_EXPERIMENTS
        if (p.probe & 2) {                      // (knock-out bit 1: no transposed reads, no MFMAs)
          D[0][0] += __builtin_bit_cast(float, (uint32_t)afrag[0][0]);
          continue;
        }
#endif
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) {
          const int hl = fb / FB_PER_H;
          s4_t b;
#ifdef GEOT_DEV_EXPERIMENTS
          if (no_tr) b = afrag[(hl + 1) % HP];
          else
#endif
          b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(img + tr_off + ks * 16 * kStride + fb * 32));
          if constexpr (__is_same(T, bf16_t)) D[fb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(afrag[hl], b, D[fb], 0, 0, 0);
          else D[fb] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h4_t, afrag[hl]), __builtin_bit_cast(h4_t, b), D[fb], 0, 0, 0);
        }
      }
      if constexpr (WMODE != 0 && PH == 2 && ph == 0) {
        if (!wpo) nx_w = load_w((int64_t)nx_pe);       // (two tiles a chunk: the permutation's entry arrived under this tile's matrix work)
      }
    };
    pair_up(0);
    if (ntiles > 0) gather(std::integral_constant<int, 0>{});
    for (int t = 0; t < ntiles; t += PH) {      // a chunk of 64 edges = PH tiles, unrolled: lane numbers of the crossbar reads are immediates
      tile(t, std::integral_constant<int, 0>{});
      if (t + 1 < ntiles) tile(t + 1, std::integral_constant<int, 1>{});
      if constexpr (PH == 4) {
        if (t + 2 < ntiles) tile(t + 2, std::integral_constant<int, 2>{});
        if (t + 3 < ntiles) tile(t + 3, std::integral_constant<int, 3>{});
      }
    }
    lock.round_done(p, lane, vr);
    // the group's rows out: D_fb holds (row 4 kq + j, feature 16 fb + m) in element j - through LDS into whole rows, eight blocks
    // (128 features, 16 rows x 512 bytes of fp32 = 8 KB of the image) at a time
    const int64_t v0 = has ? P.g_vrow0[pos] : 0;
    constexpr int PB = 8, PF = PB * 16, LR = PF / 4;           // blocks and floats of a pass, lanes that cover a row's part
#pragma unroll
    for (int hf = 0; hf < NFB / PB; ++hf) {
      wave_order();
#pragma unroll
      for (int fb = 0; fb < PB; ++fb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) imgf[(4 * kq + j) * PF + 16 * fb + m] = D[PB * hf + fb][j];
      }
      wave_order();
      if (has) {
        for (int l = lane / LR; l < nv; l += 64 / LR) {          // 64 / LR rows at a time
          const int64_t tg = P.v_out[v0 + l];
          const int colp = (lane % LR) * 4;
          const int col = pass * (PBYTES / 2) + PF * hf + colp;
          f4_t row = *reinterpret_cast<const f4_t *>(imgf + l * PF + colp);
          if (tg >= 0) {
            if constexpr (RED == GEOT_REDUCE_MEAN) {                    // (pieces of a split row are divided after the combine)
              const float tot = (float)P.v_total[v0 + l];
#pragma unroll
              for (int i = 0; i < 4; ++i) row[i] = row[i] / tot;
            }
            if (tg < p.K) {
              typedef T t4_t __attribute__((ext_vector_type(4)));
              const t4_t o = {(T)row[0], (T)row[1], (T)row[2], (T)row[3]};                 // one rounding, here
              *reinterpret_cast<t4_t *>(dst + tg * p.F + col) = o;
            }
          } else {
            *reinterpret_cast<f4_t *>(p.carry + (-tg - 1) * p.F + col) = row;             // fp32
          }
        }
      }
    }
    wave_order();
    }                                           // pass
  }
  lock.leave(lane);
}

// The vector-ALU twin of seg_slab_spmm_mfma_kernel for 16-bit rows of 1 KiB.  Plans of such rows are cut for the matrix-core kernel
// - up to 16 rows a group, whose 512 features x 16 rows of fp32 sums seg_slab_kernel's LDS cannot hold (it takes 4-5 rows: 21-26
// rounds, each a sweep of the table through every L2) - so a source table with an Inf / NaN in it (the gate, see
// slab_nonfinite_kernel) and the option "slab_spmm_mfma" = 0 are served by THIS kernel: the same plan, the same grid, no lockstep,
// nothing in flight - kSub = 4 rows of accumulators in LDS at a time, the group's edges walked once per kSub rows (their fields loaded 64
// at a time), one edge's row at a time.
// Correct and slow by design (the path a diverged model takes); out-of-range sources contribute nothing, as in the matrix-core kernel.
// WMODE 0 none | 2 weight[e * H + h] | 3 weight[h * nnz + e]; p.w_in_plan_order as everywhere.
constexpr int kTwin1kRows = 4;
template <typename T, int WMODE>
__global__ __launch_bounds__(kThreads) void seg_slab_twin1k_kernel(SlabParams p) {
  static_assert(sizeof(T) == 2 && (WMODE == 0 || WMODE == 2 || WMODE == 3), "16-bit rows of 1 KiB, multi-head weights or none");
  constexpr int kSub = kTwin1kRows;                        // rows of fp32 accumulators in LDS at a time: 4 x 2 KB a wave (8 rows, two workgroups a CU: 43 against 31 ms - fewer waves to hide the loads behind)
  typedef T t8_t __attribute__((ext_vector_type(8)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (p.gate && ((slab_gate_word(p.gate) != 0) != (p.gate_want != 0))) return;
  const geot_slab_plan &P = p.plan;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *acc = reinterpret_cast<float *>(smem) + (size_t)wave * kSub * 512;          // [kSub][512]
  const int64_t unit = (int64_t)blockIdx.x * 4 + wave, units = P.units;
  const T *weight = static_cast<const T *>(p.weight);
  const T *src = static_cast<const T *>(p.src);
  T *dst = static_cast<T *>(p.dst);
  const bool wpo = p.w_in_plan_order != 0;
  const int head = (lane * 8) / p.Fh;                      // the head of this lane's eight features (Fh = 512 / H >= 64)
  for (int r = 0; r < p.rounds; ++r) {
    const int64_t pos = (int64_t)r * units + ((r & 1) ? units - 1 - unit : unit);
    if (pos >= P.n_groups) continue;
    const int64_t e0 = P.g_begin[pos];
    const int len = (int)(P.g_begin[pos + 1] - e0), nv = P.g_nv[pos];
    const int64_t v0 = P.g_vrow0[pos];
    for (int lo = 0; lo < nv; lo += kSub) {
#pragma unroll
      for (int q = 0; q < kSub * 8; ++q) acc[q * 64 + lane] = 0.f;
      for (int c0 = 0; c0 < len; c0 += 64) {                    // a chunk of 64 edges: every lane loads one edge's fields ...
        const int mine = c0 + lane;
        const bool in = mine < len;
        const int my_dl = in ? (int)P.e_dl[e0 + mine] : 255;
        const uint32_t my_src = in ? (uint32_t)P.e_src[e0 + mine] : 0u;
        const uint32_t my_pe = (WMODE != 0 && in && !wpo) ? (uint32_t)P.e_perm[e0 + mine] : 0u;
        const int n_here = len - c0 < 64 ? len - c0 : 64;
        for (int j = 0; j < n_here; ++j) {                      // ... and the wave walks them one at a time
          const int dl = __builtin_amdgcn_readlane(my_dl, j);
          const uint32_t s_ = (uint32_t)__builtin_amdgcn_readlane((int)my_src, j);
          if (dl < lo || dl >= lo + kSub || s_ >= (uint32_t)p.src_rows) continue;      // (wave-uniform)
          float w = 1.f;
          if constexpr (WMODE != 0) {
            const int64_t pe = wpo ? e0 + c0 + j : (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)my_pe, j);
            w = (float)(WMODE == 2 ? weight[pe * p.H + head] : weight[(int64_t)head * P.nnz + pe]);
          }
          const t8_t x = *reinterpret_cast<const t8_t *>(src + (int64_t)s_ * 512 + lane * 8);
          float *a = acc + (size_t)(dl - lo) * 512 + lane * 8;
#pragma unroll
          for (int q = 0; q < 8; ++q) a[q] += w * (float)x[q];
        }
      }
      for (int l = lo; l < nv && l < lo + kSub; ++l) {
        const int64_t tg = P.v_out[v0 + l];
        const float *a = acc + (size_t)(l - lo) * 512 + lane * 8;
        if (tg >= 0) {
          if (tg < p.K) {
            t8_t o;
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] = (T)a[q];                              // one rounding, here
            *reinterpret_cast<t8_t *>(dst + tg * p.F + lane * 8) = o;
          }
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) p.carry[(-tg - 1) * p.F + lane * 8 + q] = a[q];   // fp32
        }
      }
    }
  }
}

// split hubs: dst[row] = sum of its carry slots, in slot order (one lane group per split row).  The pieces meet in FLOAT64 and are
// rounded once: a hub of 300 k edges is thousands of pieces, and with interleaved pieces (Phase A) they can all be nearly EQUAL
// (two distinct source rows: every piece samples both in proportion) - adding thousands of equal fp32 values to a growing fp32
// sum rounds the same way every time (2e-5 relative seen in the soak, over the 1e-5 bar); in float64 the drift is gone.
template <typename T, int RED>
__global__ __launch_bounds__(kThreads) void seg_slab_combine_kernel(SlabParams p) {
  constexpr int VEC = SlabVec<T>::VEC, NV = SlabVec<T>::NV;
  const geot_slab_plan &P = p.plan;
  const int lpr = 1 << p.lpr_log2;
  const int g = threadIdx.x >> p.lpr_log2, c = threadIdx.x & (lpr - 1);
  const int ng = kThreads >> p.lpr_log2;
  for (int64_t s = (int64_t)blockIdx.x * ng + g; s < P.n_split; s += (int64_t)gridDim.x * ng) {
    const int64_t row = P.c_row[s];
    const int64_t first = P.c_first[s];
    const int n = P.c_count[s];
    constexpr float kIdent = RED == GEOT_REDUCE_MAX ? -INFINITY : (RED == GEOT_REDUCE_MIN ? INFINITY : 0.f);
    f4_t acc[NV];
    if constexpr (RED == GEOT_REDUCE_SUM || RED == GEOT_REDUCE_MEAN) {
      double dacc[NV][4];
#pragma unroll
      for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) dacc[q][e] = 0.0;
      for (int i = 0; i < n; ++i)
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const f4_t m = *reinterpret_cast<const f4_t *>(p.carry + (first + i) * p.F + c * VEC + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) dacc[q][e] += (double)m[e];
        }
      const double inv = RED == GEOT_REDUCE_MEAN ? 1.0 / (double)P.c_total[s] : 1.0;
#pragma unroll
      for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[q][e] = (float)(RED == GEOT_REDUCE_MEAN ? dacc[q][e] * inv : dacc[q][e]);
    } else {
#pragma unroll
      for (int q = 0; q < NV; ++q) acc[q] = f4_t{kIdent, kIdent, kIdent, kIdent};
      for (int i = 0; i < n; ++i)
#pragma unroll
        for (int q = 0; q < NV; ++q) slab_acc<RED>(acc[q], *reinterpret_cast<const f4_t *>(p.carry + (first + i) * p.F + c * VEC + 4 * q));
    }
    if (row >= 0 && row < p.K) *reinterpret_cast<f4_t *>(static_cast<T *>(p.dst) + row * p.F + c * VEC) = slab_pack<T>(acc);
  }
}

// Second step of the staged SDDMM: the results of a group sit in the plan's order, staged[e0 .. e1); its edges are a CONTIGUOUS
// range of the dst-sorted list (a group is a run of whole dst rows) unless it holds a piece of a split hub.  One workgroup per
// group: the range goes through LDS into edge order and leaves as whole lines; groups with hub pieces (and groups longer than the
// tile) write their results one by one.
template <typename T>
__global__ __launch_bounds__(kThreads) void slab_unstage_kernel(const int64_t *__restrict__ g_begin, const int32_t *__restrict__ e_perm,
                                                                const T *__restrict__ staged, T *__restrict__ out, int64_t n_groups, int tile_elems) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T *tile = reinterpret_cast<T *>(smem);
  __shared__ int s_min, s_max;
  for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int64_t e0 = g_begin[g];
    const int len = (int)(g_begin[g + 1] - e0);
    if (threadIdx.x == 0) {
      s_min = 0x7fffffff;
      s_max = -1;
    }
    __syncthreads();
    int mn = 0x7fffffff, mx = -1;
    for (int i = threadIdx.x; i < len; i += kThreads) {
      const int e = e_perm[e0 + i];
      mn = e < mn ? e : mn;
      mx = e > mx ? e : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64);
      mn = a < mn ? a : mn;
      mx = b > mx ? b : mx;
    }
    if ((threadIdx.x & 63) == 0 && len > 0) {
      atomicMin(&s_min, mn);
      atomicMax(&s_max, mx);
    }
    __syncthreads();
    const int base = s_min;
    const bool whole = len > 0 && len <= tile_elems && s_max - base + 1 == len;   // (a permutation: len distinct ids in a span of len = the range)
    if (whole) {
      for (int i = threadIdx.x; i < len; i += kThreads) tile[e_perm[e0 + i] - base] = __builtin_nontemporal_load(staged + e0 + i);
      __syncthreads();
      for (int i = threadIdx.x; i < len; i += kThreads) __builtin_nontemporal_store(tile[i], out + base + i);
    } else {
      for (int i = threadIdx.x; i < len; i += kThreads) out[e_perm[e0 + i]] = staged[e0 + i];
    }
    __syncthreads();
  }
}

// The mirror image of slab_unstage_kernel, for per-call multi-head WEIGHTS in the caller's edge order (weight_mode 2, heads x element
// size = 8 bytes: four 16-bit heads): a group's weights - a contiguous range of the edge-order array unless the group holds a piece
// of a split hub - come in as whole lines, go through LDS and leave in the plan's order (staged[e0 + i] = weight[e_perm[e0 + i]]).
// Read through the permutation inside the persistent kernel every one of those 8-byte reads is a 64-byte sector of its own from
// beyond the L2 (the gathers turn an XCD's 4 MiB over every few microseconds): 1.2 ms at configs[3]'s graph; this pre-pass takes
// 0.76 ms (rocprofv3) - bf16 H=4 x F=64 with per-call weights 4.58 -> 4.21 ms.  1 024 threads a workgroup, every load of a phase
// issued before the first is used (a loop around a non-temporal load is not unrolled by itself: one load in flight per thread);
// the pieces of split hubs (their edges interleave inside the hub's range) are read directly.  Measured and NOT adopted: the same
// for 16-byte weights (four fp32 heads, a 128 KB tile: 7.27 against 7.20 ms through the permutation), this shape for the unstage
// kernel (1.25 against 0.95 ms), and both moves without a tile, positions dealt to the XCDs in contiguous ranges so that one L2
// sees all of a group (stage 0.97, unstage 1.33 ms); and this kernel for ONE weight per edge in place of the plain gather pre-pass
// (fp32 3.80 against 3.88 ms, bf16 2.75 against 2.59) - profiles/r06/slab_cases__mh_weights_staged_a_group_at_a_time.txt.
constexpr int kStageThreads = 1024, kStageIters = 8, kStageTileBytes = 64 * 1024;   // tiles of at most 8 192 edges
template <typename T>
__global__ __launch_bounds__(kStageThreads) void slab_stage_kernel(const int64_t *__restrict__ g_begin, const int32_t *__restrict__ e_perm,
                                                                   const T *__restrict__ weight, T *__restrict__ staged, int64_t n_groups) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T *tile = reinterpret_cast<T *>(smem);
  // (a tile of kStageIters x kStageThreads elements: 64 KB of 8-byte weights, two workgroups a CU; 128 KB of 16-byte ones, one)
  for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int64_t e0 = g_begin[g];
    const int len = (int)(g_begin[g + 1] - e0);
    if (len <= kStageIters * kStageThreads) {
      int32_t pe[kStageIters];
      T v[kStageIters];
#pragma unroll
      for (int k = 0; k < kStageIters; ++k) {
        const int i = threadIdx.x + k * kStageThreads;
        if (i < len) {
          pe[k] = e_perm[e0 + i];
          v[k] = __builtin_nontemporal_load(weight + e0 + i);
        }
      }
#pragma unroll
      for (int k = 0; k < kStageIters; ++k) {
        const int i = threadIdx.x + k * kStageThreads;
        if (i < len) tile[i] = v[k];
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < kStageIters; ++k) {
        const int i = threadIdx.x + k * kStageThreads;
        if (i < len) {
          const int64_t at = (int64_t)pe[k] - e0;
          const T w = (at >= 0 && at < len) ? tile[at] : weight[pe[k]];      // (a hub's piece: its edges lie elsewhere)
          __builtin_nontemporal_store(w, staged + e0 + i);
        }
      }
    } else {
      for (int i = threadIdx.x; i < len; i += kStageThreads) staged[e0 + i] = weight[e_perm[e0 + i]];
    }
    __syncthreads();
  }
}

extern "C" int g_slab_turn;
// The persistent grids are sized for the whole chip and keep step per XCD: two of them at once take each other's CUs and
// lockstep has nothing to offer (the kernels stay correct and bounded - SlabStep gives up - but both run slower than one
// after the other).  So the launches of this process take turns per device, whichever stream, thread or entry point (the
// torch plugin, ctypes, a C caller) they come from: a launch waits, on ITS stream, for the event of the previous one.
// An optimisation, not a safety net: a stream that is being captured skips it (an event recorded outside a capture cannot
// be waited for inside one; replayed graphs on two streams may overlap), and so does another process on the same GPU.
// (fills - the output's rows without edges, the lockstep's words, the gate word - go through geot_internal_fill: a kernel, not a
// hipMemsetAsync node; see csrc/seg_reduce.hip)
static hipError_t slab_fill(void *p, size_t bytes, uint32_t word, hipStream_t st) {
  return geot_internal_fill(p, bytes, word, st) == GEOT_OK ? hipSuccess : hipErrorUnknown;
}

struct SlabTurn {
  std::mutex mu;
  std::map<int, hipEvent_t> last; // per device
  template <typename Launch> int take(hipStream_t st, Launch launch) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) (void)hipGetLastError();
    if (!g_slab_turn || cs != hipStreamCaptureStatusNone) return launch();
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
      (void)hipGetLastError();
      return launch();
    }
    hipEvent_t &e = last[dev];
    if (e) {
      if (hipStreamWaitEvent(st, e, 0) != hipSuccess) (void)hipGetLastError();
    } else if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      e = nullptr;
      return launch();
    }
    const int rc = launch();
    if (hipEventRecord(e, st) != hipSuccess) (void)hipGetLastError();
    return rc;
  }
};
SlabTurn g_turn;

} // namespace

// (the mean exists for one head only: inside a template the other instantiations are never formed)
template <typename T, int H, int W, int ROWB>
static void slab_launch_spmm_mfma(int reduce, dim3 grid, dim3 blk, size_t lds, hipStream_t st, const SlabParams &pm) {
  if constexpr (ROWB == 1024 && H < 2) {
    (void)reduce, (void)grid, (void)blk, (void)lds, (void)st, (void)pm;      // (never dispatched: 1-KiB rows are two passes of whole heads)
  } else {
    if constexpr (H == 1 && W != 3) {
      if (reduce == GEOT_REDUCE_MEAN) {
        hipLaunchKernelGGL((seg_slab_spmm_mfma_kernel<T, H, W, ROWB, GEOT_REDUCE_MEAN>), grid, blk, lds, st, pm);
        return;
      }
    }
    hipLaunchKernelGGL((seg_slab_spmm_mfma_kernel<T, H, W, ROWB, GEOT_REDUCE_SUM>), grid, blk, lds, st, pm);
  }
}

extern "C" {

constexpr size_t kSyncBytes = (size_t)(8 * kProgSlots + 64) * sizeof(int); // progress words + slot counters
// "slab_window": how many slabs a wave may run ahead of the slowest wave of its XCD; -2 = the rule, -1 = no synchronisation.
// The rule (round-4 sweeps, profiles/r04/sweep_slab_*.txt, slab_window_*.txt, configs[3]'s graph), for 2-MiB slabs: 2 everywhere
// (512-B rows without weights 3.14 vs 3.17 ms) except a per-edge weight on rows below 1 KiB, where 1 is better (fp32 F=128 4.50 vs
// 4.75 ms, bf16 F=128 3.20 vs 3.36 ms) - on dense graphs: from 4 uses of a source row per XCD and round; 1-MiB slabs (multi-head weights: the host's slab_bytes_rule): 3
int g_slab_window = -2;

// "slab_far": the lockstep exists so that the waves of an XCD read the SAME slab at about the same time.  A wave whose slab is far
// from the slowest wave's shares nothing with it whatever it does - the case of a graph with LOCALITY (sources near their
// destinations: every group lives in its own few slabs, the groups in flight are spread over the whole table), where waiting
// for the slowest wave serialised the chip (Reddit scale, sources within +-2000 rows: 48.7 ms against 6.2 ms per edge, r03).
// Such a wave does not wait.  On graphs without locality all waves sweep the table together and never get that far apart.
int g_slab_far = 12;
// Switches of measured-and-rejected variants: variables in the DEVELOPMENT build (-DGEOT_DEV_EXPERIMENTS -> geot_amd/libgeot_hip_dev.so,
// what tools/ and the A/B tests load), constants in the product - their kernels are not instantiated there and geot_set_option refuses
// the names.
#ifdef GEOT_DEV_EXPERIMENTS
#define GEOT_DEV_SWITCH int
#else
#define GEOT_DEV_SWITCH static constexpr int
#endif
GEOT_DEV_SWITCH g_slab_nt = 0;      // "slab_nt": experiment, see SlabParams::nt_plan
GEOT_DEV_SWITCH g_slab_unroll = 8;  // "slab_unroll": 8 | 16 row loads in flight per lane of the row-per-wave kernel (sums)
GEOT_DEV_SWITCH g_slab_tight = 1;   // "slab_tight": 1 = the window of 1 slab for per-call weights on dense graphs (the round-4 rule), 0 = always 2
GEOT_DEV_SWITCH g_slab_stage = 1;   // "slab_stage": 1 = edge-order weights are staged into plan order by a pre-pass when the workspace has room, 0 = read through e_perm, 2 = multi-head weights too
int g_slab_turn = 1;    // "slab_turn": 1 = the persistent grids of this process take turns on a device (see SlabTurn), 0 = launch freely

int g_slab_blocks = 3;  // workgroups per CU of the persistent grid ("slab_blocks"; 160 KB of LDS per CU): measured 2 -> 3: -18 %, 4: same

// The persistent grid is sized for the device the process runs on: CUs and LDS per CU are read once (a partitioned
// MI355X - CPX / DPX - or a CU-masked process reports fewer CUs; the build container has no device at all: MI355X's
// numbers).  The density rule (slab_worthwhile in the host layer) was measured on the full 256-CU chip: on anything else
// geot_slab_full_chip() is 0 and the automatic routing keeps the per-edge kernels.
struct SlabDevice { int cus; size_t lds; };
static const SlabDevice &slab_device() {
  static const SlabDevice d = [] {
    SlabDevice r{256, (size_t)160 * 1024};
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) {
      r.cus = p.multiProcessorCount;
      if (p.maxSharedMemoryPerMultiProcessor >= 64 * 1024) r.lds = p.maxSharedMemoryPerMultiProcessor;
    } else {
      (void)hipGetLastError();
    }
    return r;
  }();
  return d;
}
int geot_slab_full_chip(void) { return slab_device().cus == 256 && slab_device().lds >= (size_t)160 * 1024; }

int geot_slab_units(void) { return slab_device().cus * g_slab_blocks * 4; } // waves of the persistent grid: CUs x workgroups x 4

// nv: float4 accumulators per lane (1: fp32 storage, 2: 16-bit storage - 8 elements per 16-byte lane)
static size_t slab_lds_bytes(int rows_per_group, int weight_mode, int64_t heads, int nv = 1) {
  const size_t hw = weight_mode <= 1 ? 1 : (size_t)heads;
  return (size_t)4 * rows_per_group * 1024 * nv + (size_t)4 * 2 * 64 * hw * sizeof(float); // + double-buffered weights
}

// the workgroups of a CU share its 160 KB of LDS: R * nv KiB of accumulators per wave + the staged weights
int geot_slab_rows_per_group_dtype(int weight_mode, int64_t heads, int dtype) {
  const int nv = dtype == GEOT_F32 ? 1 : 2;
  int r = 16;
  const size_t budget = g_slab_blocks <= 2 ? 64 * 1024 : (slab_device().lds - 4 * 1024) / g_slab_blocks / 1024 * 1024;
  while (r > 1 && slab_lds_bytes(r, weight_mode, heads, nv) > budget) --r;
  return r;
}
int geot_slab_rows_per_group(int weight_mode, int64_t heads) { return geot_slab_rows_per_group_dtype(weight_mode, heads, GEOT_F32); }

// WHICH FORM RUNS A PLAN is decided by the plan's `units`.  Rows of 1 KiB: a unit is a wave (seg_slab_kernel's WAVE_ROW form).  Rows
// of 512 / 256 bytes have two forms: lane groups of rowbytes / 16 lanes (units = waves x 1024 / rowbytes; seg_slab_kernel: 16 bytes
// per lane, 2 / 4 rows per wave-instruction) and one row per wave-instruction (units = waves; seg_slab_wrow_kernel: 8 / 4 bytes per
// lane, scalar row bases and row switches).  Measured at configs[3]'s graph (profiles/r05/slab_cases__row_per_wave_for_every_weight_
// mode__*.txt against round 4's lane groups): the row-per-wave form wins under MULTI-HEAD weights (bf16 H=4 x F=64 5.66 -> 5.24 ms:
// per-edge field shuffles and per-lane head selects go away) and LOSES everywhere else - gs F=128 fp32 4.15 vs 3.14 ms, gws 4.94 vs
// 4.50, the SDDMM 6.86 vs 4.52: half the bytes per load instruction, and a weightless lane-group kernel already reads at the L2's
// 18.7 TB/s.  So geot_slab_units_for keeps round 4's split; option "slab_wrow_all" = 1 makes every plan of such rows row-per-wave.
static bool slab_wrow(int64_t rowbytes) { return rowbytes == 512 || rowbytes == 256; }
GEOT_DEV_SWITCH g_slab_wrow_all = 0;
int g_slab_spmm_mfma = 1;    // "slab_spmm_mfma": 16-bit SpMM over wave-cut plans of 256- / 512-byte / 1-KiB rows on the matrix cores (seg_slab_spmm_mfma_kernel,
                             // gated on a finite source table); 0 = the row-per-wave kernel
int g_slab_sddmm_mfma = 1;   // "slab_sddmm_mfma": 16-bit SDDMM over wave-cut plans of 256- / 512-byte / 1-KiB rows on the matrix cores (seg_slab_sddmm_mfma_kernel:
                             // 4.30 vs 5.84 ms at Reddit scale); 0 = the row-per-wave kernel
GEOT_DEV_SWITCH g_slab_probe = 0;   // "slab_probe": see SlabParams::probe (results are wrong by design)
GEOT_DEV_SWITCH g_slab_pair = 0;    // "slab_pair": 1 = multi-head plans over 512-byte rows run seg_slab_wpair_kernel (two rows per instruction).  Measured
                        // SLOWER than seg_slab_wrow_kernel (bf16 H=4 x F=64, weights in plan order: 4.63 vs 4.50 ms), so off: see the kernel's header
static bool slab_wants_wrow(int weight_mode, int64_t rowbytes) {
  return slab_wrow(rowbytes) && (g_slab_wrow_all || weight_mode == 2 || weight_mode == 3 || weight_mode == 5);
}
int geot_slab_units_for(int weight_mode, int64_t rowbytes) {
  if (slab_wants_wrow(weight_mode, rowbytes) || rowbytes >= 1024 || rowbytes < 16) return geot_slab_units();
  return geot_slab_units() * (int)(1024 / rowbytes);
}
int geot_slab_rows_per_group_shape(int weight_mode, int64_t heads, int dtype, int64_t rowbytes) {
  // 16-bit multi-head plans over rows of 1 KiB: the 16 rows of a matrix-core operand (seg_slab_spmm_mfma_kernel /
  // seg_slab_sddmm_mfma_kernel run them in two passes; bf16 H=8 x F=64 at configs[3]'s graph: 6 rounds instead of the 26 that the 4
  // rows seg_slab_kernel's LDS holds would need - 6.95 against 11.45 ms; seg_slab_twin1k_kernel serves what those two cannot)
  if (g_slab_spmm_mfma && g_slab_sddmm_mfma && rowbytes == 1024 && dtype != GEOT_F32 && (heads == 2 || heads == 4 || heads == 8) &&
      (weight_mode == 2 || weight_mode == 3 || weight_mode == 5))
    return 16;
  if (!slab_wants_wrow(weight_mode, rowbytes)) return geot_slab_rows_per_group_dtype(weight_mode, heads, dtype);
  const size_t row = (size_t)(rowbytes / (dtype == GEOT_F32 ? 4 : 2)) * sizeof(float);   // a group row's accumulators
  const size_t budget = g_slab_blocks <= 2 ? 64 * 1024 : (slab_device().lds - 4 * 1024) / g_slab_blocks / 1024 * 1024;
  const size_t hw = weight_mode == 0 ? 0 : ((weight_mode == 1 || weight_mode == 4) ? 1 : (size_t)heads);
  int r = 32;
  while (r > 1 && (size_t)4 * r * row + (size_t)4 * 2 * 64 * hw * sizeof(float) > budget) --r;
  // 16-bit rows: at most the 16 rows of a matrix-core operand, so that the plan can be run by seg_slab_spmm_mfma_kernel /
  // seg_slab_sddmm_mfma_kernel (512-byte rows end below 16 by the LDS bound above; 256-byte rows would get 24)
  if (dtype != GEOT_F32 && r > 16) r = 16;
  return r;
}

// Scratch of geot_slab_spmm: control words, progress words, carry rows - and, optionally, room for the weights in plan order
// (geot_slab_workspace_bytes_staged: nnz x heads elements more): given that room, a call with EDGE-order weights (modes 1 / 2)
// stages them inside the kernel (stage_weights) instead of reading them through the permutation.
size_t geot_slab_workspace_bytes(const geot_slab_plan *plan, int64_t feat_total) {
  if (!plan) return 0;
  return ((size_t)(plan->n_carry > 0 ? plan->n_carry : 1) * (size_t)feat_total * sizeof(float) + 256 + kSyncBytes + 255) & ~(size_t)255;
}
size_t geot_slab_workspace_bytes_staged(const geot_slab_plan *plan, int64_t feat_total, int weight_mode, int64_t heads, int dtype) {
  if (!plan) return 0;
  const size_t base = geot_slab_workspace_bytes(plan, feat_total);
  // (one weight per edge, or 8 bytes of heads an edge in edge-major layout: what the pre-passes of geot_slab_spmm serve; "slab_stage"
  //  = 2, development build: every multi-head layout through the plain gather)
  const size_t wb = (size_t)(weight_mode == 1 ? 1 : heads) * (dtype == GEOT_F32 ? 4 : 2);
  if (weight_mode != 1 && !(weight_mode == 2 && (wb == 8 || (wb == 16 && dtype != GEOT_F32) || g_slab_stage == 2))) return base;
  return base + (size_t)plan->nnz * wb;
}

int geot_slab_spmm(const geot_slab_plan *plan, const void *weight, int weight_mode, const void *src, void *dst,
                   int64_t heads, int64_t feat, int64_t src_rows, int64_t out_rows, int dtype, int reduce, void *workspace,
                   size_t workspace_bytes, void *stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  // mode 4 = mode 1, mode 5 = mode 2 (edge-major [nnz, heads]) with weight[] already in the PLAN's edge order
  const bool w_in_plan_order = weight_mode == 4 || weight_mode == 5;
  const int mode_given = weight_mode;
  if (weight_mode == 4) weight_mode = 1;
  if (weight_mode == 5) weight_mode = 2;
  if (!plan || !src || !dst) return geot_internal_fail(GEOT_EINVAL, "slab_spmm: null pointer");
  if (dtype != GEOT_F32 && dtype != GEOT_F16 && dtype != GEOT_BF16) return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_spmm: float32, float16 or bfloat16");
  const int tsize = dtype == GEOT_F32 ? 4 : 2, vec = 16 / tsize, nv = vec / 4;
  if (heads < 1 || feat < 1 || out_rows < 0) return geot_internal_fail(GEOT_EINVAL, "slab_spmm: bad sizes");
  if (reduce != GEOT_REDUCE_SUM && reduce != GEOT_REDUCE_MEAN && reduce != GEOT_REDUCE_MAX && reduce != GEOT_REDUCE_MIN)
    return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_spmm: reduce must be sum, mean, max or min");
  if (reduce != GEOT_REDUCE_SUM && weight_mode >= 2) return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_spmm: multi-head weights reduce with sum only");
  if (reduce == GEOT_REDUCE_MEAN && (!plan->v_total || (plan->n_split > 0 && !plan->c_total)))
    return geot_internal_fail(GEOT_EINVAL, "slab_spmm: mean needs the plan's edge counts (v_total, c_total)");
  if (weight_mode < 0 || weight_mode > 3 || (weight_mode != 0 && !weight))
    return geot_internal_fail(GEOT_EINVAL, "slab_spmm: weight_mode 0..5 (and a weight pointer for 1..5)");
  const int64_t F = heads * feat;
  const int64_t rowbytes = F * tsize;
  int lpr_log2 = -1;
  for (int l = 3; l <= 6; ++l)
    if (rowbytes == ((int64_t)16 << l)) lpr_log2 = l;
  if (lpr_log2 < 0) return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_spmm: rows of 128, 256, 512 or 1024 bytes only");
  if (feat % vec != 0 && weight_mode >= 2) return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_spmm: feat per head must be a multiple of 16 bytes");
  if (src_rows < 0 || (uint64_t)src_rows * (uint64_t)rowbytes >= ((uint64_t)1 << 32))
    return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_spmm: source table below 4 GiB (32-bit row offsets; the kernel is for tables a slab sweep can cover)");
  if ((((uintptr_t)src) | ((uintptr_t)dst) | ((uintptr_t)workspace)) & 15) return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_spmm: 16-byte aligned operands");
  // rows of 512 / 256 bytes: the plan's units say which form it was cut for - lane groups (waves x 1024 / rowbytes) or one row per
  // wave-instruction (waves); a plan with another unit count (tests, callers with their own grids) follows geot_slab_units_for's rule
  bool mhrow = false;
  if (slab_wrow(rowbytes)) {
    if (plan->units == (int64_t)geot_slab_units() * (64 >> lpr_log2)) mhrow = false;
    else if (plan->units == (int64_t)geot_slab_units()) mhrow = true;
    else mhrow = slab_wants_wrow(mode_given, rowbytes);
  }
  const int per_wave = mhrow ? 1 : (64 >> lpr_log2);
  const int64_t waves = plan->units / per_wave;
  const int64_t cu_waves = (int64_t)4 * slab_device().cus;
  if (plan->units % per_wave != 0 || waves % 4 != 0 || waves < 4 || waves > cu_waves * 4)
    return geot_internal_fail(GEOT_EINVAL, "slab_spmm: the plan's unit count is not a whole number of 4-wave workgroups, at most 4 per CU of this device "
                                           "(geot_slab_units_for)");
  if (plan->rows_per_group < 1 || plan->rows_per_group > 32) return geot_internal_fail(GEOT_EINVAL, "slab_spmm: rows_per_group 1..32");
  const int el = (int)(F / 64);                                    // (mhrow) elements per lane
  if (mhrow && weight_mode >= 2 && (feat % el != 0 || F % 64 != 0)) return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_spmm: feat per head must be a multiple of row elements / 64");
  if (heads > 16) return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_spmm: at most 16 heads");
  const size_t need = geot_slab_workspace_bytes(plan, F);
  if (!workspace || workspace_bytes < need) return geot_internal_fail(GEOT_EWORKSPACE, "slab_spmm: workspace too small");
  if (out_rows == 0) return GEOT_OK;

  SlabParams p;
  p.plan = *plan;
  p.weight = weight;
  p.src = src;
  p.dst = dst;
  // workspace: [256 B control words of the tile kernels | progress words + slot counters | carry rows]
  p.prog = reinterpret_cast<int *>(static_cast<char *>(workspace) + 256);
  p.prog_cnt = p.prog + 8 * kProgSlots;
  p.carry = reinterpret_cast<float *>(static_cast<char *>(workspace) + 256 + kSyncBytes);
  p.slab_shift = plan->slab_shift;
  p.n_slabs = plan->n_slabs;
  // (slabs of <= 1 MiB - the host's choice under multi-head weights - keep step within 3 slabs)
  const bool small_slabs = ((int64_t)rowbytes << plan->slab_shift) <= ((int64_t)1 << 20);
  // (uses of a source row per XCD and round: the window of 1 pays on dense graphs - Reddit scale, 12 uses: 4.50 vs 4.75 ms - and
  // costs 3 % at 2.3 uses, profiles/r04/bench_slab_density_rule.txt: 7.60 vs 7.36 ms)
  const int64_t rounds_ = (plan->n_groups + plan->units - 1) / (plan->units > 0 ? plan->units : 1);
  const double uses = (double)plan->nnz / (double)(rounds_ > 0 ? rounds_ : 1) / 8.0 / (double)(src_rows > 0 ? src_rows : 1);
  const bool staged_w = g_slab_stage && weight_mode == 1 && workspace_bytes >= geot_slab_workspace_bytes_staged(plan, F, weight_mode, heads, dtype);
  const int tight = (weight_mode == 1 && !w_in_plan_order && !staged_w && lpr_log2 < 6 && uses >= 4.0 && g_slab_tight) ? 1 : 2;   // (a weight in plan order streams: 3.52 vs 3.56 ms at 2)
  p.window = (plan->slab_shift > 0 && plan->n_slabs > 1) ? (g_slab_window == -2 ? (small_slabs ? 3 : tight) : g_slab_window) : -1;
  p.far = g_slab_far;
  p.nt_plan = g_slab_nt;
  p.probe = g_slab_probe;
  p.gate = nullptr;
  p.gate_want = 0;
  p.w_in_plan_order = w_in_plan_order ? 1 : 0;
  // room behind the carry rows for the weights in plan order -> the kernel stages them itself (geot_slab_workspace_bytes_staged)
  void *wstage = nullptr;
  const int64_t wbytes = (weight_mode == 1 ? 1 : heads) * tsize;               // one edge's weights
  // (measured at configs[3]'s graph, profiles/r05/slab_cases__lane_groups__weights_staged_by_a_prepass.txt: one weight per edge - gws
  //  F=128 fp32 4.50 -> 3.98 ms, F=64 3.40 -> 2.74, bf16 F=128 3.19 -> 2.79; four heads of weights: the pre-pass moves 4 GB and costs more
  //  than the permuted reads - fp32 7.18 -> 7.51, bf16 5.24 -> 5.61: multi-head weights are staged only on request, "slab_stage" = 2)
  //  Round 6: four 16-bit heads (8 bytes an edge) go through LDS a GROUP at a time instead (slab_stage_kernel) - bf16 4.58 -> 4.21 ms.
  //  Eight 16-bit heads (16 bytes an edge) likewise, since their consumer went to the matrix cores (bf16 H=8 x F=64 9.16 -> see
  //  profiles/r06/slab_cases__rows_of_1_kib_16_bit.txt); four fp32 heads - the same 16 bytes - measured no gain and stay as they were.
  if (!w_in_plan_order && (weight_mode == 1 || (weight_mode == 2 && (wbytes == 8 || (wbytes == 16 && tsize == 2) || g_slab_stage == 2))) && g_slab_stage && plan->n_groups > 0 &&
      (wbytes == 2 || wbytes == 4 || wbytes == 8 || wbytes == 16) && (((uintptr_t)weight) & (wbytes - 1)) == 0 &&
      workspace_bytes >= geot_slab_workspace_bytes_staged(plan, F, weight_mode, heads, dtype))
    wstage = static_cast<char *>(workspace) + need;
  p.src_rows = src_rows;
  p.K = out_rows;
  p.F = F;
  p.H = (int)heads;
  p.Fh = (int)feat;
  p.rowbytes = (uint32_t)rowbytes;
  p.lpr_log2 = lpr_log2;
  p.rounds = (int)((plan->n_groups + plan->units - 1) / plan->units);

  hipError_t e = slab_fill(dst, (size_t)out_rows * (size_t)rowbytes, 0u, st); // rows without edges
  if (e != hipSuccess) return geot_internal_fail(GEOT_ELAUNCH, hipGetErrorString(e));
  if (p.window >= 0) {
    e = slab_fill(p.prog, kSyncBytes, (uint32_t)kProgIdle, st);                              // every word = kProgIdle
    if (e != hipSuccess) return geot_internal_fail(GEOT_ELAUNCH, hipGetErrorString(e));
  }
  if (wstage) {   // the weights into plan order first; the row loop then streams them (p.w_in_plan_order)
    const int64_t per_block = (int64_t)kThreads * 8;
    int64_t sblocks = (plan->nnz + per_block - 1) / per_block;
    if (sblocks > (int64_t)slab_device().cus * 16) sblocks = (int64_t)slab_device().cus * 16;
    const dim3 sgrid((unsigned)(sblocks > 0 ? sblocks : 1)), sblk(kThreads);
    const bool x4 = (((uintptr_t)wstage | (uintptr_t)plan->e_perm) & 15) == 0 && plan->nnz >= 4;
    static const hipError_t big_tile = hipFuncSetAttribute(reinterpret_cast<const void *>(&slab_stage_kernel<f4_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStageTileBytes);
    if (weight_mode == 2 && wbytes == 8 && g_slab_stage != 2) {            // a group at a time through LDS, two workgroups per CU
      const unsigned gb = (unsigned)(plan->n_groups < (int64_t)slab_device().cus * 2 ? plan->n_groups : (int64_t)slab_device().cus * 2);
      hipLaunchKernelGGL((slab_stage_kernel<uint64_t>), dim3(gb), dim3(kStageThreads), kStageTileBytes, st, plan->g_begin, plan->e_perm,
                         static_cast<const uint64_t *>(weight), static_cast<uint64_t *>(wstage), plan->n_groups);
    } else if (weight_mode == 2 && wbytes == 16 && tsize == 2 && g_slab_stage != 2 && big_tile == hipSuccess) {   // ... 128 KB tiles, one per CU
      const unsigned gb = (unsigned)(plan->n_groups < (int64_t)slab_device().cus ? plan->n_groups : (int64_t)slab_device().cus);
      hipLaunchKernelGGL((slab_stage_kernel<f4_t>), dim3(gb), dim3(kStageThreads), 2 * kStageTileBytes, st, plan->g_begin, plan->e_perm,
                         static_cast<const f4_t *>(weight), static_cast<f4_t *>(wstage), plan->n_groups);
    } else
    if (wbytes == 2 && x4) hipLaunchKernelGGL((slab_stage_weights_x4_kernel<uint16_t>), sgrid, sblk, 0, st, plan->e_perm, static_cast<const uint16_t *>(weight), static_cast<uint16_t *>(wstage), plan->nnz);
    else if (wbytes == 4 && x4) hipLaunchKernelGGL((slab_stage_weights_x4_kernel<uint32_t>), sgrid, sblk, 0, st, plan->e_perm, static_cast<const uint32_t *>(weight), static_cast<uint32_t *>(wstage), plan->nnz);
    else if (wbytes == 2) hipLaunchKernelGGL((slab_stage_weights_kernel<uint16_t>), sgrid, sblk, 0, st, plan->e_perm, static_cast<const uint16_t *>(weight), static_cast<uint16_t *>(wstage), plan->nnz);
    else if (wbytes == 4) hipLaunchKernelGGL((slab_stage_weights_kernel<uint32_t>), sgrid, sblk, 0, st, plan->e_perm, static_cast<const uint32_t *>(weight), static_cast<uint32_t *>(wstage), plan->nnz);
    else if (wbytes == 8) hipLaunchKernelGGL((slab_stage_weights_kernel<uint64_t>), sgrid, sblk, 0, st, plan->e_perm, static_cast<const uint64_t *>(weight), static_cast<uint64_t *>(wstage), plan->nnz);
    else hipLaunchKernelGGL((slab_stage_weights_kernel<f4_t>), sgrid, sblk, 0, st, plan->e_perm, static_cast<const f4_t *>(weight), static_cast<f4_t *>(wstage), plan->nnz);
    p.weight = wstage;
    p.w_in_plan_order = 1;
  }
  if (plan->n_groups > 0) {
    const size_t hw_lds = weight_mode == 0 ? 0 : (weight_mode == 1 ? 1 : (size_t)heads);
    const size_t lds = mhrow ? (size_t)4 * plan->rows_per_group * 64 * el * sizeof(float) + (size_t)4 * 2 * 64 * hw_lds * sizeof(float)
                             : slab_lds_bytes(plan->rows_per_group, weight_mode, heads, nv);
    // (16-bit multi-head plans over rows of 1 KiB are cut for the matrix-core kernel - up to 16 rows a group, which seg_slab_kernel's
    //  LDS cannot hold: seg_slab_twin1k_kernel stands in for it, see there)
    const bool big1k = lds > 64 * 1024 && rowbytes == 1024 && tsize == 2 && plan->rows_per_group <= 16 && heads >= 2 && reduce == GEOT_REDUCE_SUM &&
                       (weight_mode == 0 || weight_mode == 2 || weight_mode == 3) && plan->units == (int64_t)geot_slab_units();
    if (lds > 64 * 1024 && !big1k) return geot_internal_fail(GEOT_EINVAL, "slab_spmm: rows_per_group exceeds the LDS budget (geot_slab_rows_per_group_shape)");
    const dim3 grid((unsigned)(waves / 4)), blk(kThreads);
    const bool wave_row = lpr_log2 == 6;
    int64_t cblocks = (plan->n_split + (kThreads >> lpr_log2) - 1) / (kThreads >> lpr_log2);
    if (cblocks > 1024) cblocks = 1024;
    const dim3 cgrid((unsigned)(cblocks > 0 ? cblocks : 1));
    const bool combine = plan->n_split > 0;
#define GEOT_SLAB_LAUNCH(T_, W, RED_)                                                                         \
  do {                                                                                                        \
    geot_internal_note_kernel((std::string("seg_slab_kernel<") + slab_tname<T_>() + ", " #W ", " + (wave_row ? "true" : "false") + ", " + std::to_string((int)RED_) + ">").c_str()); \
    if (wave_row) hipLaunchKernelGGL((seg_slab_kernel<T_, W, true, RED_>), grid, blk, lds, st, p);            \
    else hipLaunchKernelGGL((seg_slab_kernel<T_, W, false, RED_>), grid, blk, lds, st, p);                    \
    if (combine) hipLaunchKernelGGL((seg_slab_combine_kernel<T_, RED_>), cgrid, blk, 0, st, p);               \
  } while (0)
#define GEOT_SLAB_RED(T_, W)                                                                                  \
  switch (reduce) {                                                                                           \
  case GEOT_REDUCE_MEAN: GEOT_SLAB_LAUNCH(T_, W, GEOT_REDUCE_MEAN); break;                                    \
  case GEOT_REDUCE_MAX: GEOT_SLAB_LAUNCH(T_, W, GEOT_REDUCE_MAX); break;                                      \
  case GEOT_REDUCE_MIN: GEOT_SLAB_LAUNCH(T_, W, GEOT_REDUCE_MIN); break;                                      \
  default: GEOT_SLAB_LAUNCH(T_, W, GEOT_REDUCE_SUM); break;                                                   \
  }
#define GEOT_SLAB_MODE(T_)                                                                                    \
  switch (weight_mode) {                                                                                      \
  case 0: GEOT_SLAB_RED(T_, 0) break;                                                                         \
  case 1: GEOT_SLAB_RED(T_, 1) break;                                                                         \
  case 2: GEOT_SLAB_LAUNCH(T_, 2, GEOT_REDUCE_SUM); break;                                                    \
  default: GEOT_SLAB_LAUNCH(T_, 3, GEOT_REDUCE_SUM); break;                                                   \
  }
#define GEOT_SLAB_WROW_U(T_, W, E_, RED_, U_)                                                                 \
  do {                                                                                                        \
    geot_internal_note_kernel((std::string("seg_slab_wrow_kernel<") + slab_tname<T_>() + ", " #W ", " #E_ ", " + std::to_string((int)RED_) + ", " #U_ ">").c_str()); \
    hipLaunchKernelGGL((seg_slab_wrow_kernel<T_, W, E_, RED_, U_>), grid, blk, lds, st, p);                  \
    if (combine) hipLaunchKernelGGL((seg_slab_combine_kernel<T_, RED_>), cgrid, blk, 0, st, p);               \
  } while (0)
  // (16 loads in flight per lane: an experiment switch, "slab_unroll"; instantiated for the sums only)
#ifdef GEOT_DEV_EXPERIMENTS
#define GEOT_SLAB_WROW(T_, W, E_, RED_)                                                                       \
  do {                                                                                                        \
    if (RED_ == GEOT_REDUCE_SUM && g_slab_unroll == 16) GEOT_SLAB_WROW_U(T_, W, E_, GEOT_REDUCE_SUM, 16);     \
    else GEOT_SLAB_WROW_U(T_, W, E_, RED_, 8);                                                                \
  } while (0)
#else
#define GEOT_SLAB_WROW(T_, W, E_, RED_) GEOT_SLAB_WROW_U(T_, W, E_, RED_, 8)
#endif
#define GEOT_SLAB_WROW_RED(T_, W, E_)                                                                         \
  switch (reduce) {                                                                                           \
  case GEOT_REDUCE_MEAN: GEOT_SLAB_WROW(T_, W, E_, GEOT_REDUCE_MEAN); break;                                  \
  case GEOT_REDUCE_MAX: GEOT_SLAB_WROW(T_, W, E_, GEOT_REDUCE_MAX); break;                                    \
  case GEOT_REDUCE_MIN: GEOT_SLAB_WROW(T_, W, E_, GEOT_REDUCE_MIN); break;                                    \
  default: GEOT_SLAB_WROW(T_, W, E_, GEOT_REDUCE_SUM); break;                                                 \
  }
#define GEOT_SLAB_WROW_MODE(T_, E_)                                                                           \
  switch (weight_mode) {                                                                                      \
  case 0: GEOT_SLAB_WROW_RED(T_, 0, E_) break;                                                                \
  case 1: GEOT_SLAB_WROW_RED(T_, 1, E_) break;                                                                \
  case 2: GEOT_SLAB_WROW(T_, 2, E_, GEOT_REDUCE_SUM); break;                                                  \
  default: GEOT_SLAB_WROW(T_, 3, E_, GEOT_REDUCE_SUM); break;                                                 \
  }
#ifdef GEOT_DEV_EXPERIMENTS
#define GEOT_SLAB_WPAIR(T_)                                                                                   \
  do {                                                                                                        \
    geot_internal_note_kernel((std::string("seg_slab_wpair_kernel<") + slab_tname<T_>() + (weight_mode == 2 ? ", 2>" : ", 3>")).c_str()); \
    if (weight_mode == 2) hipLaunchKernelGGL((seg_slab_wpair_kernel<T_, 2>), grid, blk, lds, st, p);          \
    else hipLaunchKernelGGL((seg_slab_wpair_kernel<T_, 3>), grid, blk, lds, st, p);                           \
    if (combine) hipLaunchKernelGGL((seg_slab_combine_kernel<T_, GEOT_REDUCE_SUM>), cgrid, blk, 0, st, p);    \
  } while (0)
#endif
    // 16-bit plans cut into waves over 512- / 256-byte rows, sums: the matrix-core kernel, GATED on a finite source table (its header), with the
    // row-per-wave kernel enqueued behind it under the opposite gate.  Same persistent grid, same plan.
    const bool mfma = g_slab_spmm_mfma && tsize == 2 && plan->rows_per_group <= 16 &&
                      ((mhrow && (rowbytes == 512 || rowbytes == 256) && (reduce == GEOT_REDUCE_SUM || (reduce == GEOT_REDUCE_MEAN && heads == 1))) ||
                       (rowbytes == 1024 && heads >= 2 && reduce == GEOT_REDUCE_SUM && plan->units == (int64_t)geot_slab_units())) &&
                      (heads == 1 || heads == 2 || heads == 4 || heads == 8) && feat % 16 == 0 && (weight_mode != 1 || heads == 1) &&
                      (weight_mode == 0 || (((uintptr_t)p.weight) & (uintptr_t)(weight_mode == 2 ? heads * 2 - 1 : 1)) == 0);
    const int rc = g_turn.take(st, [&]() -> int {
      if (mfma) {
        int *flag = p.prog_cnt + 32;                                   // (a word of the scratch's control block nobody else uses)
        hipError_t me = slab_fill(flag, sizeof(int), 0u, st);
        if (me != hipSuccess) return geot_internal_fail(GEOT_ELAUNCH, hipGetErrorString(me));
        SlabParams pm = p;
        pm.gate = flag;
        pm.gate_want = 0;
        p.gate = flag;                                                 // the vector-ALU twin below runs iff the table holds an Inf / NaN
        p.gate_want = 1;
        const int64_t n16 = src_rows * rowbytes / 16;
        const int wm = weight_mode == 1 ? 1 : weight_mode;             // (one weight per edge: H == 1)
#define GEOT_SLAB_SPMM_MFMA_W(T_, H_, W_)                                                                     \
        do {                                                                                                  \
          const size_t mlds = (size_t)4 * ((rowbytes >= 512 ? 16 * (512 + 32) : 32 * (256 + 32)) + 2 * H_ * 64 * 2 + 2 * 64);   /* per wave: image + staged weights + rows in group */ \
          geot_internal_note_kernel((std::string("seg_slab_spmm_mfma_kernel<") + slab_tname<T_>() + ", " #H_ ", " #W_ ", " + std::to_string(rowbytes) + ">").c_str()); \
          if (reduce == GEOT_REDUCE_MEAN)                                                                     \
            geot_internal_note_kernel((std::string("seg_slab_spmm_mfma_kernel<") + slab_tname<T_>() + ", " #H_ ", " #W_ ", " + std::to_string(rowbytes) + ", mean>").c_str()); \
          if (rowbytes == 512) slab_launch_spmm_mfma<T_, H_, W_, 512>(reduce, grid, blk, mlds, st, pm);       \
          else if (rowbytes == 1024) slab_launch_spmm_mfma<T_, H_, W_, 1024>(reduce, grid, blk, mlds, st, pm); \
          else slab_launch_spmm_mfma<T_, H_, W_, 256>(reduce, grid, blk, mlds, st, pm);                       \
        } while (0)
#define GEOT_SLAB_SPMM_MFMA_H(T_, H_)                                                                         \
        do {                                                                                                  \
          if (wm == 0) GEOT_SLAB_SPMM_MFMA_W(T_, H_, 0);                                                      \
          else if (wm == 3) GEOT_SLAB_SPMM_MFMA_W(T_, H_, 3);                                                 \
          else GEOT_SLAB_SPMM_MFMA_W(T_, H_, 2);                                                              \
        } while (0)
#define GEOT_SLAB_SPMM_MFMA(T_)                                                                               \
        do {                                                                                                  \
          hipLaunchKernelGGL((slab_nonfinite_kernel<T_>), dim3((unsigned)(slab_device().cus * 8)), blk, 0, st, static_cast<const uint32_t *>(src), n16, flag); \
          if (heads == 1) GEOT_SLAB_SPMM_MFMA_H(T_, 1);                                                       \
          else if (heads == 2) GEOT_SLAB_SPMM_MFMA_H(T_, 2);                                                  \
          else if (heads == 4) GEOT_SLAB_SPMM_MFMA_H(T_, 4);                                                  \
          else GEOT_SLAB_SPMM_MFMA_H(T_, 8);                                                                  \
        } while (0)
        if (dtype == GEOT_F16) GEOT_SLAB_SPMM_MFMA(half_t);
        else GEOT_SLAB_SPMM_MFMA(bf16_t);
#undef GEOT_SLAB_SPMM_MFMA
#undef GEOT_SLAB_SPMM_MFMA_H
#undef GEOT_SLAB_SPMM_MFMA_W
      }
      const std::string mfma_name = mfma ? std::string(geot_last_kernel()) : std::string();
#ifdef GEOT_DEV_EXPERIMENTS
      if (mhrow && g_slab_pair && weight_mode >= 2 && rowbytes == 512 && feat % vec == 0) {
        if (dtype == GEOT_F32) { GEOT_SLAB_WPAIR(float); }
        else if (dtype == GEOT_F16) { GEOT_SLAB_WPAIR(half_t); }
        else { GEOT_SLAB_WPAIR(bf16_t); }
      } else
#endif
      if (mhrow) {
        if (dtype == GEOT_F32) { if (el == 2) { GEOT_SLAB_WROW_MODE(float, 2) } else { GEOT_SLAB_WROW_MODE(float, 1) } }
        else if (dtype == GEOT_F16) { if (el == 4) { GEOT_SLAB_WROW_MODE(half_t, 4) } else { GEOT_SLAB_WROW_MODE(half_t, 2) } }
        else { if (el == 4) { GEOT_SLAB_WROW_MODE(bf16_t, 4) } else { GEOT_SLAB_WROW_MODE(bf16_t, 2) } }
      }
      else if (big1k) {
        const size_t tlds = (size_t)4 * kTwin1kRows * 512 * sizeof(float);   // four rows of fp32 sums a wave
#define GEOT_SLAB_TWIN1K(T_)                                                                                  \
        do {                                                                                                  \
          geot_internal_note_kernel((std::string("seg_slab_twin1k_kernel<") + slab_tname<T_>() + ", " + std::to_string(weight_mode) + ">").c_str()); \
          if (weight_mode == 0) hipLaunchKernelGGL((seg_slab_twin1k_kernel<T_, 0>), grid, blk, tlds, st, p);  \
          else if (weight_mode == 2) hipLaunchKernelGGL((seg_slab_twin1k_kernel<T_, 2>), grid, blk, tlds, st, p); \
          else hipLaunchKernelGGL((seg_slab_twin1k_kernel<T_, 3>), grid, blk, tlds, st, p);                   \
          if (combine) hipLaunchKernelGGL((seg_slab_combine_kernel<T_, GEOT_REDUCE_SUM>), cgrid, blk, 0, st, p); \
        } while (0)
        if (dtype == GEOT_F16) GEOT_SLAB_TWIN1K(half_t);
        else GEOT_SLAB_TWIN1K(bf16_t);
#undef GEOT_SLAB_TWIN1K
      }
      else if (dtype == GEOT_F32) { GEOT_SLAB_MODE(float) }
      else if (dtype == GEOT_F16) { GEOT_SLAB_MODE(half_t) }
      else { GEOT_SLAB_MODE(bf16_t) }
      if (mfma) geot_internal_note_kernel(mfma_name.c_str());         // (the twin behind it returns at once unless the table holds an Inf / NaN)
      const hipError_t le = hipGetLastError();
      return le == hipSuccess ? GEOT_OK : geot_internal_fail(GEOT_ELAUNCH, hipGetErrorString(le));
    });
#ifdef GEOT_DEV_EXPERIMENTS
#undef GEOT_SLAB_WPAIR
#endif
#undef GEOT_SLAB_WROW_MODE
#undef GEOT_SLAB_WROW_RED
#undef GEOT_SLAB_WROW
#undef GEOT_SLAB_WROW_U
#undef GEOT_SLAB_MODE
#undef GEOT_SLAB_RED
#undef GEOT_SLAB_LAUNCH
    if (rc != GEOT_OK) return rc;
  }
  return GEOT_OK;
}

// staging != NULL: the persistent kernel writes its results in the plan's edge order into `staging`; `unstage`: a second kernel brings
// them into original edge order in `out` (otherwise plan order IS the result and `out` is not touched)
// out[i, :] = weight[e_perm[i], :] - per-edge values (heads per edge) from the caller's edge order into the plan's: what weight modes 4 / 5
// read.  heads x element size of 2 / 4 / 8 / 16 bytes (GEOT_EUNSUPPORTED otherwise: the caller permutes by other means).
int geot_slab_to_plan_order(const geot_slab_plan *plan, const void *weight, void *out, int64_t heads, int dtype, void *stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!plan || !weight || !out || heads < 1) return geot_internal_fail(GEOT_EINVAL, "slab_to_plan_order: bad arguments");
  if (dtype != GEOT_F32 && dtype != GEOT_F16 && dtype != GEOT_BF16) return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_to_plan_order: float32, float16 or bfloat16");
  const int64_t wbytes = heads * (dtype == GEOT_F32 ? 4 : 2);
  if ((wbytes != 2 && wbytes != 4 && wbytes != 8 && wbytes != 16) || ((((uintptr_t)weight) | ((uintptr_t)out)) & (uintptr_t)(wbytes - 1)))
    return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_to_plan_order: heads x element size of 2, 4, 8 or 16 bytes, aligned");
  if (plan->nnz == 0) return GEOT_OK;
  const int64_t per_block = (int64_t)kThreads * 8;
  int64_t sblocks = (plan->nnz + per_block - 1) / per_block;
  if (sblocks > (int64_t)slab_device().cus * 16) sblocks = (int64_t)slab_device().cus * 16;
  const dim3 sgrid((unsigned)sblocks), sblk(kThreads);
  const bool x4 = (((uintptr_t)out | (uintptr_t)plan->e_perm) & 15) == 0 && plan->nnz >= 4;
  if (wbytes == 2 && x4) hipLaunchKernelGGL((slab_stage_weights_x4_kernel<uint16_t>), sgrid, sblk, 0, st, plan->e_perm, static_cast<const uint16_t *>(weight), static_cast<uint16_t *>(out), plan->nnz);
  else if (wbytes == 4 && x4) hipLaunchKernelGGL((slab_stage_weights_x4_kernel<uint32_t>), sgrid, sblk, 0, st, plan->e_perm, static_cast<const uint32_t *>(weight), static_cast<uint32_t *>(out), plan->nnz);
  else if (wbytes == 2) hipLaunchKernelGGL((slab_stage_weights_kernel<uint16_t>), sgrid, sblk, 0, st, plan->e_perm, static_cast<const uint16_t *>(weight), static_cast<uint16_t *>(out), plan->nnz);
  else if (wbytes == 4) hipLaunchKernelGGL((slab_stage_weights_kernel<uint32_t>), sgrid, sblk, 0, st, plan->e_perm, static_cast<const uint32_t *>(weight), static_cast<uint32_t *>(out), plan->nnz);
  else if (wbytes == 8) hipLaunchKernelGGL((slab_stage_weights_kernel<uint64_t>), sgrid, sblk, 0, st, plan->e_perm, static_cast<const uint64_t *>(weight), static_cast<uint64_t *>(out), plan->nnz);
  else hipLaunchKernelGGL((slab_stage_weights_kernel<f4_t>), sgrid, sblk, 0, st, plan->e_perm, static_cast<const f4_t *>(weight), static_cast<f4_t *>(out), plan->nnz);
  const hipError_t le = hipGetLastError();
  return le == hipSuccess ? GEOT_OK : geot_internal_fail(GEOT_ELAUNCH, hipGetErrorString(le));
}

static int slab_sddmm_impl(const geot_slab_plan *plan, const void *mat_1, const void *mat_2, void *out, void *staging, int64_t heads,
                           int64_t feat, int64_t rows_1, int64_t rows_2, int dtype, void *workspace, size_t workspace_bytes, void *stream,
                           bool unstage = true) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!plan || !mat_1 || !mat_2 || (!out && (unstage || !staging))) return geot_internal_fail(GEOT_EINVAL, "slab_sddmm: null pointer");
  if (dtype != GEOT_F32 && dtype != GEOT_F16 && dtype != GEOT_BF16) return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_sddmm: float32, float16 or bfloat16");
  if (!plan->v_row) return geot_internal_fail(GEOT_EINVAL, "slab_sddmm: the plan has no v_row table");
  if (heads < 1 || feat < 1) return geot_internal_fail(GEOT_EINVAL, "slab_sddmm: bad sizes");
  const int tsize = dtype == GEOT_F32 ? 4 : 2;
  const int64_t F = heads * feat;
  const int64_t rowbytes = F * tsize;
  int lpr_log2 = -1;
  for (int l = 4; l <= 6; ++l)
    if (rowbytes == ((int64_t)16 << l)) lpr_log2 = l;
  if (lpr_log2 < 0) return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_sddmm: rows of 256, 512 or 1024 bytes only");
  // a dot product per head is reduced over the 64 / heads lanes of the head: heads 1, 2, 4 or 8, at least 8 lanes each
  if (heads != 1 && heads != 2 && heads != 4 && heads != 8) return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_sddmm: 1, 2, 4 or 8 heads");
  if ((((uintptr_t)mat_1) | ((uintptr_t)mat_2) | ((uintptr_t)workspace)) & 15) return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_sddmm: 16-byte aligned operands");
  if (rows_2 < 0 || (uint64_t)rows_2 * (uint64_t)rowbytes >= ((uint64_t)1 << 32))
    return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_sddmm: mat_2 below 4 GiB (32-bit row offsets)");
  // the form follows the plan's units, as in geot_slab_spmm: lane groups (the plan of a single-weight forward) or one row per
  // wave-instruction (a multi-head plan; "slab_wrow_all"); the multi-head SDDMM needs whole-wave rows
  bool wrow = false;
  if (slab_wrow(rowbytes)) {
    if (plan->units == (int64_t)geot_slab_units() * (64 >> lpr_log2)) wrow = false;
    else if (plan->units == (int64_t)geot_slab_units()) wrow = true;
    else wrow = slab_wants_wrow(heads > 1 ? 2 : 1, rowbytes);
  }
  if (heads > 1 && lpr_log2 < 6 && !wrow)
    return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_mh_sddmm: needs a plan cut into waves (geot_slab_units_for with weight_mode 2)");
  const int per_wave = (wrow || lpr_log2 == 6) ? 1 : (64 >> lpr_log2);
  const int64_t waves = plan->units / per_wave;
  const int64_t cu_waves = (int64_t)4 * slab_device().cus;
  if (plan->units % per_wave != 0 || waves % 4 != 0 || waves < 4 || waves > cu_waves * 4)
    return geot_internal_fail(GEOT_EINVAL, "slab_sddmm: the plan's unit count is not a whole number of 4-wave workgroups, at most 4 per CU of this device");
  if (plan->rows_per_group < 1 || plan->rows_per_group > 32) return geot_internal_fail(GEOT_EINVAL, "slab_sddmm: rows_per_group 1..32");
  if (!workspace || workspace_bytes < 256 + kSyncBytes) return geot_internal_fail(GEOT_EWORKSPACE, "slab_sddmm: workspace too small");
  const size_t lds = (size_t)4 * plan->rows_per_group * (size_t)(wrow ? rowbytes : 1024);   // the group's mat_1 rows, storage type (per wave: R rows x its units)
  if (lds > 64 * 1024) return geot_internal_fail(GEOT_EINVAL, "slab_sddmm: rows_per_group exceeds the LDS budget (geot_slab_rows_per_group_shape)");
  const bool staged = staging != nullptr;
  const int64_t ebytes = heads * tsize;                    // one edge's results
  if (staged && unstage && ebytes != 2 && ebytes != 4 && ebytes != 8 && ebytes != 16)
    return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_sddmm_staged: heads x element size of 2, 4, 8 or 16 bytes");
  if (plan->n_groups == 0) return GEOT_OK;
  SlabParams p;
  p.plan = *plan;
  p.weight = mat_1;
  p.src = mat_2;
  p.dst = staged ? staging : out;
  p.carry = nullptr;
  p.far = g_slab_far;
  p.nt_plan = g_slab_nt;
  p.probe = g_slab_probe;
  p.gate = nullptr;
  p.gate_want = 0;
  p.w_in_plan_order = staged ? 1 : 0;
  p.src_rows = rows_2;
  p.K = rows_1;
  p.F = F;
  p.H = (int)heads;
  p.Fh = (int)feat;
  p.rowbytes = (uint32_t)rowbytes;
  p.lpr_log2 = lpr_log2;
  p.rounds = (int)((plan->n_groups + plan->units - 1) / plan->units);
  p.prog = reinterpret_cast<int *>(static_cast<char *>(workspace) + 256);
  p.prog_cnt = p.prog + 8 * kProgSlots;
  p.slab_shift = plan->slab_shift;
  p.n_slabs = plan->n_slabs;
  p.window = (plan->slab_shift > 0 && plan->n_slabs > 1) ? (g_slab_window == -2 ? 2 : g_slab_window) : -1;
  hipError_t e = hipSuccess;
  if (p.window >= 0) {
    e = slab_fill(p.prog, kSyncBytes, (uint32_t)kProgIdle, st);
    if (e != hipSuccess) return geot_internal_fail(GEOT_ELAUNCH, hipGetErrorString(e));
  }
  const dim3 grid((unsigned)(waves / 4)), blk(kThreads);
  const int el = (int)(F / 64);                            // (rows of 512 / 256 bytes) elements per lane
  // 16-bit plans cut into waves over 512- / 256-byte rows, results in plan order: the matrix-core kernel (seg_slab_sddmm_mfma_kernel) when the group's
  // rows fit one 16-column operand and a head is a whole number of 32-feature slices (H = 1 / 2 / 4 / 8)
  const int nch = (int)(F / 32);
  const bool mfma_1k = rowbytes == 1024 && (heads == 2 || heads == 4 || heads == 8) && plan->units == (int64_t)geot_slab_units();   // (two passes of whole heads)
  if (g_slab_sddmm_mfma && staged && tsize == 2 && ((wrow && (rowbytes == 512 || rowbytes == 256)) || mfma_1k) && plan->rows_per_group <= 16 &&
      (heads == 1 || heads == 2 || heads == 4 || heads == 8) && nch % (int)heads == 0 && (((uintptr_t)mat_1 | (uintptr_t)staging) & 15) == 0) {
    const size_t xlds = (size_t)4 * (rowbytes >= 512 ? 16 * (512 + 32) : 32 * (256 + 32));   // per wave: the tile image
    const int cph = nch / (int)heads;
    const int rc = g_turn.take(st, [&]() -> int {
#define GEOT_SLAB_MFMA_B(T_, C_)                                                                               \
      do {                                                                                                     \
        if (rowbytes == 512) hipLaunchKernelGGL((seg_slab_sddmm_mfma_kernel<T_, C_, 512>), grid, blk, xlds, st, p); \
        else if (rowbytes == 1024) { if constexpr (C_ >= 2) hipLaunchKernelGGL((seg_slab_sddmm_mfma_kernel<T_, (C_ >= 2 ? C_ : 2), 1024>), grid, blk, xlds, st, p); } \
        else hipLaunchKernelGGL((seg_slab_sddmm_mfma_kernel<T_, C_, 256>), grid, blk, xlds, st, p);           \
      } while (0)
#define GEOT_SLAB_MFMA(T_)                                                                                     \
      do {                                                                                                     \
        geot_internal_note_kernel((std::string("seg_slab_sddmm_mfma_kernel<") + slab_tname<T_>() + ", " + std::to_string(cph) + ", " + std::to_string(rowbytes) + ">").c_str()); \
        if (cph == 8 && rowbytes == 1024) hipLaunchKernelGGL((seg_slab_sddmm_mfma_kernel<T_, 8, 1024>), grid, blk, xlds, st, p);   /* (two heads over 1-KiB rows) */ \
        else if (cph == 8) hipLaunchKernelGGL((seg_slab_sddmm_mfma_kernel<T_, 8, 512>), grid, blk, xlds, st, p);   /* (one head over 512-byte rows) */ \
        else if (cph == 4) GEOT_SLAB_MFMA_B(T_, 4);                                                            \
        else if (cph == 2) GEOT_SLAB_MFMA_B(T_, 2);                                                            \
        else GEOT_SLAB_MFMA_B(T_, 1);                                                                          \
      } while (0)
      if (dtype == GEOT_F16) GEOT_SLAB_MFMA(half_t);
      else GEOT_SLAB_MFMA(bf16_t);
#undef GEOT_SLAB_MFMA
#undef GEOT_SLAB_MFMA_B
      const hipError_t le = hipGetLastError();
      return le == hipSuccess ? GEOT_OK : geot_internal_fail(GEOT_ELAUNCH, hipGetErrorString(le));
    });
    if (rc != GEOT_OK || !unstage) return rc;
  } else {
  // (a plan cut for the matrix-core kernels - 16 rows of 1 KiB a group - has no vector-ALU SDDMM: the group's mat_1 rows would take
  //  64 KB of LDS a workgroup, and the persistent grid's workgroups must all be resident)
  if (lds * (size_t)g_slab_blocks > slab_device().lds)
    return geot_internal_fail(GEOT_EUNSUPPORTED, "slab_sddmm: this plan's groups (geot_slab_rows_per_group_shape: cut for the matrix-core kernel) need the "
                                                 "staged 16-bit multi-head form (geot_slab_mh_sddmm with slab_sddmm_mfma = 1)");
#define GEOT_SLAB_SDDMM(T_, E2_, E1_)                                                                        \
  do {                                                                                                        \
    geot_internal_note_kernel((std::string(wrow ? "seg_slab_sddmm_wrow_kernel<" : "seg_slab_sddmm_kernel<") + slab_tname<T_>() + ">").c_str()); \
    if (lpr_log2 == 6) hipLaunchKernelGGL((seg_slab_sddmm_kernel<T_, true>), grid, blk, lds, st, p);          \
    else if (!wrow) hipLaunchKernelGGL((seg_slab_sddmm_kernel<T_, false>), grid, blk, lds, st, p);            \
    else if (el == E2_) hipLaunchKernelGGL((seg_slab_sddmm_wrow_kernel<T_, E2_>), grid, blk, lds, st, p);     \
    else hipLaunchKernelGGL((seg_slab_sddmm_wrow_kernel<T_, E1_>), grid, blk, lds, st, p);                    \
  } while (0)
  const int rc = g_turn.take(st, [&]() -> int {
    if (dtype == GEOT_F32) GEOT_SLAB_SDDMM(float, 2, 1);
    else if (dtype == GEOT_F16) GEOT_SLAB_SDDMM(half_t, 4, 2);
    else GEOT_SLAB_SDDMM(bf16_t, 4, 2);
    const hipError_t le = hipGetLastError();
    return le == hipSuccess ? GEOT_OK : geot_internal_fail(GEOT_ELAUNCH, hipGetErrorString(le));
  });
#undef GEOT_SLAB_SDDMM
  if (rc != GEOT_OK || !staged || !unstage) return rc;
  }
  int64_t ublocks = plan->n_groups < (int64_t)slab_device().cus * 8 ? plan->n_groups : (int64_t)slab_device().cus * 8;
  constexpr int kTileBytes = 48 * 1024;                    // (a group of configs[3]'s plans: ~6 000 results)
#define GEOT_UNSTAGE(ET_)                                                                                     \
  hipLaunchKernelGGL((slab_unstage_kernel<ET_>), dim3((unsigned)ublocks), dim3(kThreads), kTileBytes, st, plan->g_begin, plan->e_perm, \
                     static_cast<const ET_ *>(staging), static_cast<ET_ *>(out), plan->n_groups, kTileBytes / (int)sizeof(ET_))
  if (ebytes == 2) GEOT_UNSTAGE(uint16_t);
  else if (ebytes == 4) GEOT_UNSTAGE(float);
  else if (ebytes == 8) GEOT_UNSTAGE(uint64_t);
  else GEOT_UNSTAGE(f4_t);
#undef GEOT_UNSTAGE
  const hipError_t ue = hipGetLastError();
  return ue == hipSuccess ? GEOT_OK : geot_internal_fail(GEOT_ELAUNCH, hipGetErrorString(ue));
}

int geot_slab_sddmm(const geot_slab_plan *plan, const void *mat_1, const void *mat_2, void *out, int64_t feat, int64_t rows_1,
                    int64_t rows_2, int dtype, void *workspace, size_t workspace_bytes, void *stream) {
  return slab_sddmm_impl(plan, mat_1, mat_2, out, nullptr, 1, feat, rows_1, rows_2, dtype, workspace, workspace_bytes, stream);
}

int geot_slab_sddmm_staged(const geot_slab_plan *plan, const void *mat_1, const void *mat_2, void *out, void *staging, int64_t feat,
                           int64_t rows_1, int64_t rows_2, int dtype, void *workspace, size_t workspace_bytes, void *stream) {
  if (!staging) return geot_internal_fail(GEOT_EINVAL, "slab_sddmm_staged: null staging buffer");
  return slab_sddmm_impl(plan, mat_1, mat_2, out, staging, 1, feat, rows_1, rows_2, dtype, workspace, workspace_bytes, stream);
}

// Multi-head form (d/dweight of geot_mh_spmm over the plan of its forward): out(e, h) = <mat_1[dst(e), h, :], mat_2[src(e), h, :]>,
// out[e * heads + h] (edge-major, the layout the source-blocked forward reads).  out == NULL: the results stay in `staging`, in the
// PLAN's edge order - what geot_slab_spmm's weight_mode 5 reads back without a permutation (attention: SDDMM -> per-row softmax in
// plan order -> SpMM).  heads 1 / 2 / 4 / 8, heads x element size <= 16 bytes.
int geot_slab_mh_sddmm(const geot_slab_plan *plan, const void *mat_1, const void *mat_2, void *out, void *staging, int64_t heads, int64_t feat,
                       int64_t rows_1, int64_t rows_2, int dtype, void *workspace, size_t workspace_bytes, void *stream) {
  if (!staging && !out) return geot_internal_fail(GEOT_EINVAL, "slab_mh_sddmm: null output");
  if (!out)     // plan order is the result: the persistent kernel writes it, no second step
    return slab_sddmm_impl(plan, mat_1, mat_2, nullptr, staging, heads, feat, rows_1, rows_2, dtype, workspace, workspace_bytes, stream, false);
  return slab_sddmm_impl(plan, mat_1, mat_2, out, staging, heads, feat, rows_1, rows_2, dtype, workspace, workspace_bytes, stream);
}

int geot_internal_slab_option(const char *name, int value) {       // 1 = a name of this file (applied if the value is in range), 0 = not
  if (!name) return 0;
  const std::string n(name);
  if (n == "slab_window") g_slab_window = value;
  else if (n == "slab_turn") g_slab_turn = value != 0;
  else if (n == "slab_far") { if (value >= 0) g_slab_far = value; }
  else if (n == "slab_sddmm_mfma") g_slab_sddmm_mfma = value != 0;
  else if (n == "slab_spmm_mfma") g_slab_spmm_mfma = value != 0;
  else if (n == "slab_blocks") { if (value >= 1 && value <= 4) g_slab_blocks = value; }
#ifdef GEOT_DEV_EXPERIMENTS
  else if (n == "slab_nt") g_slab_nt = value != 0;
  else if (n == "slab_unroll") { if (value == 8 || value == 16) g_slab_unroll = value; }
  else if (n == "slab_tight") g_slab_tight = value != 0;
  else if (n == "slab_stage") { if (value >= 0 && value <= 2) g_slab_stage = value; }
  else if (n == "slab_wrow_all") g_slab_wrow_all = value != 0;
  else if (n == "slab_pair") g_slab_pair = value != 0;
  else if (n == "slab_probe") g_slab_probe = value;          // bit 0: drop the row reads; bits 1..3 (matrix-core SpMM only): skip the MFMAs / the image writes / the transposed reads
#endif
  else return 0;
  return 1;
}

} // extern "C"
