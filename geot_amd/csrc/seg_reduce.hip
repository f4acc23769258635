// seg_reduce.hip -- MI355X (gfx950 / CDNA4) segment-reduction kernels behind include/geot_hip.h.
//
// Kernels in this file (details, measurements and the roofline of each: DESIGN.md section 3):
//   seg_tile_kernel     index_scatter / gather_scatter / gather_weight_scatter / mh_spmm, sorted index, any
//                       reduction, fp32 / fp64 / half / bfloat16 (fp32 accumulate); also the atomic flush
//                       used for an unsorted index.  The reference has four copies of one control flow
//                       (csrc/cuda/index_scatter_kernel.cuh:135-201, gather_scatter_kernel.cuh:118-186,
//                       gather_weight_scatter_kernel.cuh:118-185, mh_spmm_kernel.cuh:28-213) that flush
//                       every run with atomicAdd into a zeroed dst; this is a different algorithm: no
//                       global atomics, every dst row written exactly once, deterministic.
//   seg_lane_kernel     fp32 rows of 1..8 values: a lane owns E consecutive edges, one segmented scan per wave.
//   seg_narrow_kernel   the same rows, lane-per-edge segmented shuffle scan per 64 edges (fallback / option).
//   seg_fixup_kernel    second launch: finishes the runs that straddle tiles, zero-fills large gaps.
//   seg_wsum_kernel     few-key inputs only: pre-reduces the carries of whole 64-tile windows for the fix-up.
//   index_probe_kernel  index[-1] and the number of descents of an index in one pass (row rule + sorted=False).
//   seg_lds_bin_kernel  unsorted index with an output that fits in LDS: LDS-binned atomics.
//   sddmm_coo_kernel, gather_rows_kernel, csr_expand_kernel, coo_*_kernel   backward / CSR helpers.
//   (seg_slab.hip, the other translation unit of the library: seg_slab_kernel / seg_slab_sddmm_kernel, the
//    source-blocked persistent kernels for dense graphs.)
//
// Sorted path in one paragraph:
//   * edge-balanced tiles: block b owns edges [b*TE, (b+1)*TE) whatever the segment lengths are;
//   * 256 threads = 4 wave64; a "lane group" of LPR lanes owns one row at a time, 16 B per lane, so a
//     wave instruction moves 64/LPR whole rows; each group walks CG consecutive edges with U row loads
//     in flight, running sum in registers; run starts come from wave ballots made while the keys are
//     staged in LDS; interior runs are stored straight to dst; a group's first and last run go to LDS;
//   * after one barrier the tile's LDS partials are merged in edge order: complete runs are stored, a
//     run continuing FROM the previous tile goes to carry[tile][0], one continuing INTO the next tile
//     to carry[tile][1];
//   * seg_fixup_kernel writes dst[k] = tail + head (+ whole follow-on tiles of a hub) with plain stores;
//   * empty keys are zero-filled by whoever sees the key jump: dst needs no memset pass.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "geot_hip.h"
#include "geot_hip_dev.h"
#include "internal.h"

// Build: one storage type's instantiations of the tile kernel per object.  seg_reduce_{f32,f64,f16,bf16}.hip include this file
// with GEOT_SEG_PART = 1..4 and keep run_segment_op<T> of their type (geot_seg::run_part_*); this file by itself (part 0)
// holds the state, the C ABI and the small kernels.  The objects compile side by side (one of them alone took 1 min 40 s).
// GEOT_HEADLINE_ONLY (tools/isa.sh: seconds) keeps everything in one object.
#ifndef GEOT_SEG_PART
#define GEOT_SEG_PART 0
#endif
#if defined(GEOT_HEADLINE_ONLY)
#define GEOT_SEG_SPLIT 0
#else
#define GEOT_SEG_SPLIT 1
#endif

namespace {

constexpr int kThreads = 256;
constexpr int kU = 8;          // row loads in flight per lane (16 where the plan rule says so)
constexpr int kNarrowMaxF = 8; // fp32 rows up to this width use the narrow-row kernels (seg_lane_kernel / seg_narrow_kernel)
constexpr int kGapInline = 16; // gaps up to this many rows are zeroed by the lane group itself
constexpr int kMinLprLog2 = 2; // lane groups are at least 4 lanes wide (1-2 lane groups measured slower:
                               // 512 LDS partials per tile make the merge the bottleneck)
constexpr int64_t kNoKey = -2; // key of the padding edges behind the end of the edge list (-1 = before edge 0)

struct SegParams {
  const int64_t *dst_index;
  const int64_t *src_index;
  const void *weight;
  const void *src;
  void *dst;
  void *carry;              // [num_tiles, 2, F] slot 0: head partial (run continues from the
                            //   previous tile); slot 1: tail partial (run starts here, continues)
  int64_t *meta;            // [num_tiles]      first_key*4 + head_continues + 2*single_key_tile
  int64_t *ccnt;            // [num_tiles, 2]   edge counts of the two carry slots (mean only)
  unsigned long long *ctrl; // [0] large-gap count, [1] fix-up ticket; zero between calls
  int64_t *gap_list;        // pairs (first_row, n_rows)
  int64_t gap_cap;
  int64_t nnz, F, K, src_rows;
  int64_t H, Fh;            // mh_spmm: heads, features per head (F = H*Fh)
  uint32_t rowbytes;        // F * sizeof(T)
  int lpr_log2;             // lanes per row, log2
  int cg;                   // edges per lane-group sub-chunk (multiple of 16)
  int xcd_swizzle;          // gather modes: contiguous tile ranges per XCD
  int nt_keys;              // non-temporal key loads (with nt row loads)
  // long-run inputs (few keys, hub segments): per 64-tile window, the sum of the carries that join a
  // chain when the whole window is one run (seg_wsum_kernel); null when the launcher did not ask for it
  void *wsum;               // [windows][F] accumulators
  int64_t *wcnt;            // [windows] edge counts (mean)
  int *wflag;               // [windows] 1 = every tile of the window is `single`
  // row-rule read-back published by the kernel itself (geot_publish_word): the first thread of the first workgroup
  // copies *pub_src (the caller's index[-1]) to pub_dst[0] in pinned host memory, then pub_seq to pub_dst[1]
  const int64_t *pub_src;
  int64_t *pub_dst;
  int64_t pub_seq;
  // descent guard (see seg_fixup_kernel): the staging loops of the sorted kernels ballot "key < previous key" for free
  // and raise ctrl[kCtrlDescent]; the fix-up kernel then repairs the call in place.  mode: 0 index_scatter, 1 gather,
  // 2 gather + weight[e], 3 / 4 multi-head weights edge- / head-major (the repair pass re-reads the operands).
  int mode;
  int64_t *alarm;           // pinned host word (or null): set when a call had to be repaired (geot_set_alarm_word)
  // in-kernel hand-off of the tile carries (see "hand-off" in seg_tile_kernel): per-tile flag words, this call's tag
  unsigned long long *flags; // [num_tiles]
  unsigned long long epoch;
  int handoff;              // 1: the tile kernel finishes the straddling runs itself, the second launch only tidies up
  int ho_tries;             // polls of a predecessor's flag before a run is left to the second launch
};

// control words at the head of the workspace (zero between calls)
enum { kCtrlGaps = 0, kCtrlTicket = 1, kCtrlDescent = 2, kCtrlZeroed = 3, kCtrlGo = 4, kCtrlChunk = 5, kCtrlDone = 6, kCtrlDeferred = 7 };

// wave-uniform: some lane saw its key below its predecessor's -> the index is NOT ascending, whatever the caller or the
// host layer's remembered facts said.  One plain store; the fix-up kernel (next launch) reads it.
__device__ __forceinline__ void raise_descent(const SegParams &p, bool desc) {
  if (__ballot(desc) != 0ull && (threadIdx.x & 63) == 0) p.ctrl[kCtrlDescent] = 1ull;
}

// one 8-byte device word to the host while the kernel runs (fine-grained pinned memory): value first, fence, then
// the sequence number the host is spinning on
__device__ __forceinline__ void publish_word(const SegParams &p) {
  if (p.pub_dst && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    const int64_t v = *p.pub_src;
    __hip_atomic_store(p.pub_dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __hip_atomic_store(p.pub_dst + 1, p.pub_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---- write-through ("sc1") accesses for the in-kernel hand-off of the tile carries -------------------------------------
// A carry row written by one workgroup is read by another one - possibly on another XCD - while the kernel runs.  Plain
// stores stay in the writing XCD's L2; `sc1` stores write through and `sc1` loads bypass the reader's L1
// (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility": every handed-off byte stored and
// loaded `sc1`, each storing wave drained with s_waitcnt vmcnt(0), the workgroup's barrier, then ONE lane's flag store;
// the consumer loads only after its poll of the flag has matched).  16-byte forms; the load waits for its data in the same
// asm statement, so the compiler never sees a register that is still in flight.
typedef float ho_f4 __attribute__((ext_vector_type(4)));
// HAZARD: a VMEM store of more than 8 bytes reads its data registers a few cycles after issue, and on gfx940+ a VALU write to
// those registers needs 2 wait states behind it.  The compiler inserts them for its own stores; it cannot see into an asm
// statement - and did, for the 16-bit mean kernel (two stores per lane), compute the second store's address INTO the first
// store's data registers in the very next instruction: elements 0-1 of the lane's piece went out wrong.  Hence the s_nop here.
__device__ __forceinline__ void ho_store16(float *p, const ho_f4 &v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void ho_load16x4(const float *p0, const float *p1, const float *p2, const float *p3, ho_f4 &a, ho_f4 &b, ho_f4 &c,
                                            ho_f4 &d) {
  asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
               "global_load_dwordx4 %1, %5, off sc1\n\t"
               "global_load_dwordx4 %2, %6, off sc1\n\t"
               "global_load_dwordx4 %3, %7, off sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
               : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
               : "memory");
}

// Storage types: float, double, and the 16-bit types with fp32 accumulation (the reference's CPU path
// accumulates half/bfloat16 in an fp32 buffer and converts once at the end,
// csrc/cpu/index_scatter_cpu.cpp:78-86,114-116).
typedef _Float16 half_t;
typedef __bf16 bf16_t;
template <typename T> struct AccOf { typedef T type; };
template <> struct AccOf<half_t> { typedef float type; };
template <> struct AccOf<bf16_t> { typedef float type; };

template <typename T, int VEC> struct VecOf { typedef T type __attribute__((ext_vector_type(VEC))); };
template <typename T> struct VecOf<T, 1> { typedef T type; };

// load VEC storage elements (one 16-B access at full width), widen to the accumulator type
template <typename T, int VEC, bool NT>
__device__ __forceinline__ void load_vec(const void *p, typename AccOf<T>::type (&v)[VEC]) {
  using V = typename VecOf<T, VEC>::type;
  using A = typename AccOf<T>::type;
  V x;
  if constexpr (NT) x = __builtin_nontemporal_load(reinterpret_cast<const V *>(p));
  else x = *reinterpret_cast<const V *>(p);
  if constexpr (VEC == 1) v[0] = (A)x;
  else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = (A)x[i];
  }
}

// narrow to the storage type (one rounding, at the very end) and store
template <typename T, int VEC, bool NT = false>
__device__ __forceinline__ void store_vec(T *p, const typename AccOf<T>::type (&v)[VEC]) {
  using V = typename VecOf<T, VEC>::type;
  V x;
  if constexpr (VEC == 1) x = (T)v[0];
  else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) x[i] = (T)v[i];
  }
  if constexpr (NT) __builtin_nontemporal_store(x, reinterpret_cast<V *>(p));
  else *reinterpret_cast<V *>(p) = x;
}

// Reductions of the reference's CPU path (csrc/cpu/index_scatter_cpu.cpp:124-134; init / update /
// write of ATen/native/cpu/ReduceUtils.h).  Codes follow csrc/reducetype.h:3.
enum { RED_MAX = 0, RED_MEAN = 1, RED_MIN = 2, RED_SUM = 3, RED_PROD = 4 };

template <typename T, int RED> __device__ __forceinline__ T red_ident() {
  if constexpr (RED == RED_PROD) return T(1);
  else if constexpr (RED == RED_MAX) return -INFINITY;
  else if constexpr (RED == RED_MIN) return INFINITY;
  else return T(0);
}

// ATen's _max/_min propagate NaN: isnan(y) ? y : max(x, y)
template <typename T, int RED> __device__ __forceinline__ T red_op(T x, T y) {
  if constexpr (RED == RED_PROD) return x * y;
  else if constexpr (RED == RED_MAX) return (y != y) ? y : (x < y ? y : x);
  else if constexpr (RED == RED_MIN) return (y != y) ? y : (y < x ? y : x);
  else return x + y;
}

// LDS carve-up, shared by kernel and launcher.  te (edges per tile) is a multiple of 64.
struct SmemLayout {
  int te;
  size_t off_keys; // int64 [te + 2] : [0] key of edge ts-1, [1+i] key of edge ts+i, [te+1] of ts+te
  size_t off_off;  // int64 [te]     : byte offset of the gathered src row (gather modes)
  size_t off_pk;   // int64 [2*ng]   : keys of the LDS partials
  size_t off_mask; // u64   [te/64]  : bit i = "edge i starts a new run" (wave ballots)
  size_t off_p;    // T     [2*ng][FB]
  size_t off_w;    // T     [te*hw]  : edge weights, edge-major
  size_t off_cnt;  // int   [2*ng]   : edge counts of the partials (mean)
  size_t bytes;
};

__host__ __device__ inline SmemLayout smem_layout(int lpr_log2, int cg, int vec, int tsize,
                                                  bool gather, int hw) {
  SmemLayout L;
  const int ng = kThreads >> lpr_log2;
  L.te = ng * cg;
  size_t o = 0;
  L.off_keys = o; o += sizeof(int64_t) * (size_t)(L.te + 2);
  L.off_off = o;  if (gather) o += sizeof(int64_t) * (size_t)L.te;
  L.off_pk = o;   o += sizeof(int64_t) * (size_t)(2 * ng);
  L.off_mask = o; o += sizeof(uint64_t) * (size_t)(L.te / 64);
  o = (o + 15) & ~(size_t)15;
  L.off_p = o;    o += (size_t)2 * kThreads * vec * tsize; // 2*ng*FB, FB = lpr*vec
  L.off_w = o;    o += (size_t)tsize * L.te * hw;
  o = (o + 7) & ~(size_t)7;
  L.off_cnt = o;  o += sizeof(int) * (size_t)(2 * ng);
  L.bytes = (o + 15) & ~(size_t)15;
  return L;
}

// WMODE: 0 none, 1 weight[e], 2 weight[e*H + h], 3 weight[h*nnz + e]
// NT   : bit 0 non-temporal row loads, bit 1 non-temporal dst stores
// RAG ("ragged"): rows that are NOT whole vectors (F = 601 / 602 floats; rows and bases then sit on 4- or 8-byte boundaries).  gfx950
// serves 16-byte global accesses at 4-byte alignment at ~0.9 of the aligned rate, 4-byte accesses at half of it (tools/kexp3.hip,
// profiles/r04/kexp3_misaligned_vectors.txt), so such rows keep the full-width lanes: the row's LAST lane owns fewer than VEC
// elements; it works on the columns [F - VEC, F) instead - a whole vector that ends with the row, overlapping its neighbour's
// columns - and every global store of it skips the first `shift` = VEC - (its own elements) entries.  Nothing is ever read or
// written outside a row.  Its LDS partials and accumulators simply carry the overlap along.  No in-kernel hand-off for such rows.
template <typename T, int VEC, bool GATHER, int WMODE, bool ATOMIC, int NT, int RED, int U, bool RAG>
__device__ __forceinline__ void seg_tile_body(const SegParams &p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using A = typename AccOf<T>::type; // accumulator / LDS / carry type
  constexpr bool NTL = (NT & 1) != 0, NTS = (NT & 2) != 0;
  const int lpr = 1 << p.lpr_log2;
  const int ng = kThreads >> p.lpr_log2;
  const int cg = p.cg;
  const int hw = WMODE == 0 ? 0 : (WMODE == 1 ? 1 : (int)p.H);
  const SmemLayout L = smem_layout(p.lpr_log2, cg, VEC, (int)sizeof(A), GATHER, hw);
  const int te = L.te;
  const int FB = lpr * VEC;

  int64_t *keysL = reinterpret_cast<int64_t *>(smem + L.off_keys);
  int64_t *offL = reinterpret_cast<int64_t *>(smem + L.off_off);
  int64_t *pkL = reinterpret_cast<int64_t *>(smem + L.off_pk);
  unsigned long long *maskL = reinterpret_cast<unsigned long long *>(smem + L.off_mask);
  const unsigned char *mask8L = smem + L.off_mask;
  A *pL = reinterpret_cast<A *>(smem + L.off_p);
  A *wL = reinterpret_cast<A *>(smem + L.off_w);
  int *cntL = reinterpret_cast<int *>(smem + L.off_cnt);
  constexpr bool MEAN = RED == RED_MEAN;
  // the instantiations that can finish their straddling runs in-kernel ("hand-off", at the end of the kernel): streamed
  // rows, fp32 accumulators, whole 16-byte pieces per lane, no per-run counts
  constexpr bool kHandoff = !RAG && !GATHER && WMODE == 0 && !ATOMIC && std::is_same<A, float>::value && VEC % 4 == 0;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  publish_word(p);
  // Gather modes: blocks are dealt round-robin over the 8 XCDs (block b and b+8 share an L2), so give
  // every XCD a CONTIGUOUS range of tiles - neighbouring dst rows of a graph with locality gather
  // overlapping src rows, which then hit in that XCD's L2.  Pure placement: any mapping is correct.
  int64_t tile = blockIdx.x;
  if constexpr (GATHER) {
    if (p.xcd_swizzle) {
      const int64_t nb = gridDim.x, per = nb / 8;
      if (tile < per * 8) tile = (tile % 8) * per + tile / 8; // bijective on [0, 8*per); the tail keeps its id
    }
  }
  const int64_t ts = tile * (int64_t)te;
  const int64_t rem = p.nnz - ts;
  const int n = rem < (int64_t)te ? (int)rem : te; // valid edges in this tile, >= 1

  const T *__restrict__ weight = static_cast<const T *>(p.weight);
  T *__restrict__ dst = static_cast<T *>(p.dst);
  const int64_t F = p.F;
  const int64_t K = p.K;
  const uint32_t rb = p.rowbytes;

  const int g = tid >> p.lpr_log2; // lane group in block
  const int c = tid & (lpr - 1);   // lane in group
  const int64_t f0 = (int64_t)blockIdx.y * FB + (int64_t)c * VEC;
  const bool active = f0 < F;      // false only in the last feature block of a ragged F
  // (RAG) the row's last lane: `shift` leading entries of its vector belong to its neighbour
  const int shift = (RAG && active && f0 + VEC > F) ? (int)(f0 + VEC - F) : 0;
  const int64_t f0c = active ? f0 - shift : 0; // loads of inactive lanes are clamped, never predicated
  const int gs = g * cg;

  // streamed operand: tile base is wave-uniform, the lane adds a 32-bit byte offset
  const char *tbase = static_cast<const char *>(p.src) + (GATHER ? 0 : ts * (int64_t)rb);
  const uint32_t fbytes = (uint32_t)(f0c * (int64_t)sizeof(T));
  int hh = 0;
  if constexpr (WMODE >= 2) {
    hh = (int)(f0c / p.Fh);
    if (hh >= (int)p.H) hh = (int)p.H - 1;
  }

  A v[U][VEC];
  auto load_batch = [&](int b) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int r = gs + b + u;
      r = r < n ? r : n - 1; // padding rows re-read the last valid row; their sum is discarded
      if constexpr (GATHER) load_vec<T, VEC, NTL>(tbase + (offL[r] + fbytes), v[u]);
      else load_vec<T, VEC, NTL>(tbase + ((uint32_t)r * rb + fbytes), v[u]);
    }
  };

  if constexpr (!GATHER) load_batch(0); // rows do not depend on the keys: get them moving first

  // ---- stage keys, run-start bitmasks (wave ballot), gather offsets, weights --------------------
  unsigned long long descents = 0ull;
  for (int i0 = __builtin_amdgcn_readfirstlane(tid >> 6) * 64; i0 < te; i0 += kThreads) { // (i0: wave-uniform, scalar loop)
    const int i = i0 + lane;
    const int64_t ge = ts + i;
    int64_t k = kNoKey;
    if (ge < p.nnz) {
      if (NTL && p.nt_keys) k = __builtin_nontemporal_load(p.dst_index + ge); // keys are read once as well
      else k = p.dst_index[ge];
    }
    int64_t kp = __shfl_up(k, 1, 64);
    if (lane == 0) kp = ge > 0 ? (ge - 1 < p.nnz ? p.dst_index[ge - 1] : kNoKey) : -1;
    keysL[1 + i] = k;
    if (i == 0) keysL[0] = kp;
    const unsigned long long m = __ballot(k != kp);
    if (lane == 0) maskL[i0 >> 6] = m;
    // descent guard (see repair_call): a REAL edge (not the padding behind the list, not edge 0) whose key lies below its
    // predecessor's, at least ONE of the two inside [0, K).  Keys outside that range are ignored by every kernel - the speculative
    // `index - lo` of geot_amd/sharding.py makes leading negatives and trailing keys >= K, both ascending: no alarm; a descent
    // between two ignored keys is harmless: no alarm - but a descent THROUGH an ignored key ([1, K+3, 1], [1, -5, 1]: row 1 would
    // be written by two separate runs, the second overwriting the first) has one in-range side and is caught.
    // Only a wave-uniform bit is kept here (an SGPR: the row loads in flight leave no VGPR to spare); the flag is stored once,
    // behind the loop.
    if constexpr (!ATOMIC) descents |= __ballot(ge > 0 && ge < p.nnz && k < kp && ((uint64_t)k < (uint64_t)p.K || (uint64_t)kp < (uint64_t)p.K));
    if constexpr (GATHER) {
      int64_t row = ge < p.nnz ? p.src_index[ge] : 0;
      if ((uint64_t)row >= (uint64_t)p.src_rows) row = 0; // out-of-range gather index: memory-safe
      offL[i] = row * (int64_t)rb;
    }
    if constexpr (WMODE == 1) wL[i] = ge < p.nnz ? (A)weight[ge] : A(0);
  }
  if constexpr (WMODE == 2) {
    const int64_t base = ts * p.H, lim = p.nnz * p.H;
    for (int j = tid; j < te * hw; j += kThreads) wL[j] = base + j < lim ? (A)weight[base + j] : A(0);
  }
  if constexpr (WMODE == 3) {
    for (int j = tid; j < te * hw; j += kThreads) {
      const int h = j / te, i = j - h * te;
      wL[i * hw + h] = ts + i < p.nnz ? (A)weight[(int64_t)h * p.nnz + ts + i] : A(0);
    }
  }
  if (tid == 0) keysL[te + 1] = ts + te < p.nnz ? p.dst_index[ts + te] : kNoKey;
  if (descents != 0ull && lane == 0) p.ctrl[kCtrlDescent] = 1ull;
  __syncthreads();

  if constexpr (!ATOMIC) {
    if (tid == 0 && blockIdx.y == 0) {
      const int64_t kf = keysL[1], kl = keysL[n], kn = keysL[n + 1];
      const int64_t head = kf == keysL[0];
      const int64_t single = head && kl == kf && kn == kf;
      const int64_t mword = (uint64_t)kf < (uint64_t)K ? (kf * 4 + head + 2 * single) : 0;
      if (kHandoff && p.handoff) __hip_atomic_store(&p.meta[tile], mword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (write-through)
      else p.meta[tile] = mword;
    }
  }

  T *dstf = dst + f0c;
  auto put = [&](T *q, const A (&val)[VEC]) { // a dst row's VEC entries of this lane (RAG: the row's last lane stores only its own)
    if (!RAG || shift == 0) store_vec<T, VEC, NTS>(q, val);
    else
      for (int i = shift; i < VEC; ++i) q[i] = (T)val[i];
  };
  auto gapfill = [&](int64_t lo, int64_t hi) {
    if (hi <= lo || lo < 0 || hi > K) return; // also rejects the padding key and unsorted input
    const int64_t cnt = hi - lo;
    if (cnt <= kGapInline) {
      if (active) {
        A z[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          A t = A(0);
          asm volatile("" : "+v"(t)); // (made here, on the rare path: a hoisted zero vector would sit in VEC registers across the whole walk)
          z[i] = t;
        }
        for (int64_t r = lo; r < hi; ++r) put(dstf + r * F, z);
      }
    } else if (c == 0 && blockIdx.y == 0) {
      const unsigned long long slot = atomicAdd(&p.ctrl[0], 1ull);
      if ((int64_t)slot < p.gap_cap) {
        p.gap_list[2 * slot] = lo;
        p.gap_list[2 * slot + 1] = cnt;
      }
    }
  };

  A acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = red_ident<A, RED>();
  int64_t cur = keysL[1 + gs];
  bool first = true;
  int cnt = 0; // edges in the current run (mean)
  if constexpr (!ATOMIC) {
    const int64_t kprev = keysL[gs];
    if (cur > kprev + 1) gapfill(kprev + 1, cur);
    // rows behind the last key (a caller may ask for more rows than index[-1]+1, e.g. the
    // backward pass wants src.shape[0] rows): zero-filled by the last tile
    if (tile == (int64_t)gridDim.x - 1 && g == 0) gapfill(keysL[n] + 1, K); // (tile is the remapped id)
  }
  if constexpr (GATHER) load_batch(0);

  for (int b = 0;;) {
    unsigned m8 = mask8L[(gs + b) >> 3];
    if constexpr (U == 16) m8 |= (unsigned)mask8L[((gs + b) >> 3) + 1] << 8;
    if (b == 0) m8 &= ~1u; // the group's first edge opens its first run, it does not end one
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (m8 & (1u << u)) {
        // the run of `cur` ends in front of local edge gs+b+u
        const int64_t knew = keysL[1 + gs + b + u];
        if constexpr (ATOMIC) {
          if constexpr (sizeof(T) >= 4) { // float atomics exist for fp32 / fp64 only
            if (active && (uint64_t)cur < (uint64_t)K) {
#pragma unroll
              for (int i = 0; i < VEC; ++i)
                if (!RAG || i >= shift) atomicAdd(dstf + cur * F + i, acc[i]);
            }
          }
        } else {
          if (first) {
            store_vec<A, VEC>(pL + (size_t)(2 * g) * FB + c * VEC, acc);
            if (c == 0) {
              pkL[2 * g] = cur;
              if constexpr (MEAN) cntL[2 * g] = cnt;
            }
            first = false;
          } else if (active && (uint64_t)cur < (uint64_t)K) {
            if constexpr (MEAN) {
#pragma unroll
              for (int i = 0; i < VEC; ++i) acc[i] = acc[i] / A(cnt);
            }
            put(dstf + cur * F, acc);
          }
          if (knew > cur + 1) gapfill(cur + 1, knew);
        }
        cur = knew;
        cnt = 0;
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = red_ident<A, RED>();
      }
      if constexpr (MEAN) ++cnt;
      if constexpr (WMODE == 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = red_op<A, RED>(acc[i], v[u][i]);
      } else {
        const A w = WMODE == 1 ? wL[gs + b + u] : wL[(gs + b + u) * hw + hh];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = red_op<A, RED>(acc[i], v[u][i] * w);
      }
    }
    b += U;
    if (b >= cg) break;
    load_batch(b);
  }

  // the group's last run
  if constexpr (ATOMIC) {
    if constexpr (sizeof(T) >= 4) {
      if (active && (uint64_t)cur < (uint64_t)K) {
#pragma unroll
        for (int i = 0; i < VEC; ++i)
          if (!RAG || i >= shift) atomicAdd(dstf + cur * F + i, acc[i]);
      }
    }
    return;
  } else {
    // a group with a single run fills its second slot with the identity under the same key, so that
    // every slot is valid and "same run" is plain key equality in the merge below
    const int slot = 2 * g + (first ? 0 : 1);
    store_vec<A, VEC>(pL + (size_t)slot * FB + c * VEC, acc);
    if (first) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = red_ident<A, RED>();
      store_vec<A, VEC>(pL + (size_t)(slot + 1) * FB + c * VEC, acc);
    }
    if (c == 0) {
      pkL[slot] = cur;
      if constexpr (MEAN) cntL[slot] = cnt;
      if (first) {
        pkL[slot + 1] = cur;
        if constexpr (MEAN) cntL[slot + 1] = 0;
      }
    }
  }

  // ---- merge the tile's LDS partials in edge order ---------------------------------------------
  __syncthreads();
  const int64_t kprev_tile = keysL[0];
  const int64_t knext_tile = keysL[n + 1];
  const int ne = 2 * ng;
  // hand-off: does a run that came in from the previous tile END in this tile?  (lane group 0 learns it at partial 0)
  bool ho_pending = false;
  int64_t ho_key = -1;
  int64_t ho_cnt = 0; // (mean: the edges of that run inside this tile)
  A ho_head[VEC];
#pragma unroll
  for (int q = 0; q < VEC; ++q) ho_head[q] = red_ident<A, RED>();
  for (int i = g; i < ne; i += ng) {
    const int64_t k = pkL[i];
    if (i > 0 && pkL[i - 1] == k) continue; // not the first partial of its run
    A sum[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) sum[q] = pL[(size_t)i * FB + c * VEC + q];
    bool at_end = true; // the merged run reaches the last edge of the tile
    int64_t csum = MEAN ? cntL[i] : 0;
    int j = i + 1;
    if (j < ne) {
      if (pkL[j] != k) at_end = false; // short runs (the common case) stop at the first probe
      else {
        if constexpr (MEAN) csum += cntL[j];
#pragma unroll
        for (int q = 0; q < VEC; ++q) sum[q] = red_op<A, RED>(sum[q], pL[(size_t)j * FB + c * VEC + q]);
        // a long run (hub segment): walk it in batches of kMB slots whose LDS reads are independent,
        // one round trip per batch instead of one per slot; partials are still added in edge order
        constexpr int kMB = VEC > 4 ? 4 : 8; // (8 elements per lane: half the batch, the same registers)
        for (++j; j < ne; j += kMB) {
          int64_t kk[kMB];
          A val[kMB][VEC];
          int cc[kMB];
#pragma unroll
          for (int t = 0; t < kMB; ++t) {
            const int jj = j + t < ne ? j + t : ne - 1;
            kk[t] = pkL[jj];
            if constexpr (MEAN) cc[t] = cntL[jj];
#pragma unroll
            for (int q = 0; q < VEC; ++q) val[t][q] = pL[(size_t)jj * FB + c * VEC + q];
          }
          bool stop = false;
#pragma unroll
          for (int t = 0; t < kMB; ++t) {
            const bool in = j + t < ne;
            if (in && kk[t] != k) stop = true;
            if (in && !stop) {
              if constexpr (MEAN) csum += cc[t];
#pragma unroll
              for (int q = 0; q < VEC; ++q) sum[q] = red_op<A, RED>(sum[q], val[t][q]);
            }
          }
          if (stop) { at_end = false; break; }
        }
      }
    }
    const int cslot_id = (i == 0 && k == kprev_tile) ? 0 : ((at_end && k == knext_tile && k != kNoKey) ? 1 : -1); // (padding is no run)
    if constexpr (MEAN) {
      if (cslot_id >= 0 && c == 0 && blockIdx.y == 0) {
        if (kHandoff && p.handoff) // (read by the tile in which the run ends, while this kernel runs: written through, like the carry rows)
          __hip_atomic_store(&p.ccnt[tile * 2 + cslot_id], csum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
          p.ccnt[tile * 2 + cslot_id] = csum;
      }
    }
    if constexpr (kHandoff) {
      if (p.handoff && cslot_id == 0) {
        ho_pending = !(at_end && k == knext_tile); // (a run that also leaves the tile passes through: the tile where it ends takes it)
        if constexpr (MEAN && sizeof(T) == 4) { // (fp32 storage; the 16-bit kernels have registers to spare)
          // (the mean kernel's merge loop is the register peak - counts ride along - and 4 more live registers cost a wave per
          //  SIMD: the run's head is parked in LDS until the hand-off block - in partial slot 0, which only this lane group reads
          //  and has just consumed: cslot_id == 0 happens at i == 0 only)
          A *park = pL + c * VEC;
#pragma unroll
          for (int q = 0; q < VEC; ++q) park[q] = sum[q];
          if (c == 0) cntL[0] = (int)csum; // (edges of this tile: fits)
        } else {
          ho_key = k;
          if constexpr (MEAN) ho_cnt = csum;
#pragma unroll
          for (int q = 0; q < VEC; ++q) ho_head[q] = sum[q];
        }
      }
    }
    if (!active) continue;
    A *cslot = static_cast<A *>(p.carry) + (tile * 2) * F + f0c;
    auto put_carry = [&](A *q, const A (&val)[VEC]) {
      if (!RAG || shift == 0) store_vec<A, VEC>(q, val);
      else
        for (int i = shift; i < VEC; ++i) q[i] = val[i];
    };
    if constexpr (kHandoff) {
      if (p.handoff && cslot_id >= 0) { // write-through: another workgroup reads this row while the kernel runs
        float *cs = reinterpret_cast<float *>(cslot + (cslot_id == 1 ? F : 0));
#pragma unroll
        for (int q = 0; q < VEC; q += 4) ho_store16(cs + q, ho_f4{sum[q], sum[q + 1], sum[q + 2], sum[q + 3]});
        continue;
      }
    }
    if (cslot_id == 0) {
      // continues a run that started in an earlier tile: slot 0, added by seg_fixup_kernel
      put_carry(cslot, sum);
    } else if (cslot_id == 1) {
      // starts here and continues into the next tile: slot 1; seg_fixup_kernel writes the row
      put_carry(cslot + F, sum);
    } else if ((uint64_t)k < (uint64_t)K) {
      if constexpr (MEAN) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) sum[q] = sum[q] / A(csum);
      }
      put(dstf + k * F, sum);
    }
  }

  // ---- hand-off: the tile in which a straddling run ENDS finishes it - no second pass over the tiles -----------------
  // Every tile has written its two carry rows and its meta word write-through; now every wave drains its stores, the
  // workgroup meets, ONE lane raises the tile's flag (this call's tag).  Lane group 0 of a tile whose incoming run ends
  // here walks back over its predecessors: the nearest one is waited for (bounded: it is an earlier workgroup, running or
  // done), further ones are sampled - lane q looks at tile j - q: flag first, then (only behind a raised flag) the meta word;
  // tiles that are `single` pass the run through (their slot 0 joins), the first one that is not starts it (slot 1).  The
  // partial rows are then read four at a time and added in a FIXED order (nearest tile first): deterministic.  If the wait
  // runs out the run is left to the second launch (ctrl[kCtrlDeferred]), which always follows and otherwise only tidies up.
  if constexpr (kHandoff) {
    if (p.handoff) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(&p.flags[tile], p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (g == 0 && ho_pending) {
        const float *carry = static_cast<const float *>(p.carry);
        // (one feature block in this mode: the lane's element offset is recomputed here rather than kept alive across the walk)
        int cc = c;
        asm volatile("" : "+v"(cc)); // (keeps everything derived from it behind this point: no register held across the row walk)
        const int64_t hf0 = (int64_t)cc * VEC < F ? (int64_t)cc * VEC : 0;
        // the tiles' partials meet in float64 and are rounded once: a hub over thousands of tiles is summed tile after tile here,
        // and a float32 running sum would lose ~(tiles) ulps on the way (two tiles: the same correctly rounded sum as a + b)
        if constexpr (MEAN && sizeof(T) == 4) { // (parked by this lane group during the merge; same lanes, no barrier needed beyond the one above)
          const A *park = pL + cc * VEC;
#pragma unroll
          for (int q = 0; q < VEC; ++q) ho_head[q] = park[q];
          ho_cnt = cntL[0];
          ho_key = keysL[0];
        }
        double hacc[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) hacc[q] = (double)ho_head[q];
        int64_t j = tile - 1; // the nearest predecessor not yet taken in
        bool done = false, ok = j >= 0;
        while (ok && !done) {
          const int64_t mine = j - cc;
          bool ready = false;
          if (mine >= 0) {
            ready = __hip_atomic_load(&p.flags[mine], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.epoch;
            if (cc == 0) {
              for (int tries = 0; !ready && tries < p.ho_tries; ++tries) {
                __builtin_amdgcn_s_sleep(1);
                ready = __hip_atomic_load(&p.flags[mine], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.epoch;
              }
            }
          }
          int64_t mw = 0;
          if (ready) mw = __hip_atomic_load(&p.meta[mine], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long gmask = lpr == 64 ? ~0ull : ((1ull << lpr) - 1ull);
          const unsigned long long R = __ballot(ready) & gmask, S = __ballot(ready && (mw & 2)) & gmask;
          if (!(R & 1ull)) { // the nearest predecessor never showed up
            ok = false;
            break;
          }
          // tiles j .. j-k+1 are single (and there), tile j-k is the first that is not single - or not there yet
          const int k = (~S & gmask) ? __builtin_ctzll(~S) : lpr;
          const bool start_here = k < lpr && ((R >> k) & 1ull);
          const int take = start_here ? k + 1 : k; // >= 1
          for (int q0 = 0; q0 < take; q0 += 4) {
            if constexpr (MEAN) { // the partial runs' edge counts (one word per tile and slot; every lane of the group reads the same ones)
              for (int u = 0; u < 4 && q0 + u < take; ++u) {
                const int q = q0 + u;
                ho_cnt += __hip_atomic_load(&p.ccnt[(j - q) * 2 + (q < k ? 0 : 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
            }
#pragma unroll
            for (int v0 = 0; v0 < VEC; v0 += 4) {
              ho_f4 r[4];
              const float *ptr[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int q = q0 + u < take ? q0 + u : take - 1; // (padding re-reads the last row; not added)
                ptr[u] = carry + ((j - q) * 2 + (q < k ? 0 : 1)) * F + hf0 + v0;
              }
              ho_load16x4(ptr[0], ptr[1], ptr[2], ptr[3], r[0], r[1], r[2], r[3]);
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                if (q0 + u < take) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) hacc[v0 + e] = red_op<double, RED>((double)r[u][e], hacc[v0 + e]);
                }
              }
            }
          }
          j -= take;
          done = start_here;
          if (!done && j < 0) ok = false; // (cannot happen: tile 0 is never single)
        }
        if (ok) {
          if constexpr (MEAN) {
            const double inv = 1.0 / (double)ho_cnt;
#pragma unroll
            for (int q = 0; q < VEC; ++q) hacc[q] *= inv;
          }
#pragma unroll
          for (int q = 0; q < VEC; ++q) ho_head[q] = (A)hacc[q];
          if (active && (uint64_t)ho_key < (uint64_t)K) store_vec<T, VEC, NTS>(dstf + ho_key * F, ho_head);
        } else if (c == 0) {
          p.ctrl[kCtrlDeferred] = 1ull;
        }
      }
    }
  }
}

template <typename T, int VEC, bool GATHER, int WMODE, bool ATOMIC, int NT, int RED = RED_SUM, int U = kU>
__global__ __launch_bounds__(kThreads) void seg_tile_kernel(SegParams p) {
  seg_tile_body<T, VEC, GATHER, WMODE, ATOMIC, NT, RED, U, false>(p);
}
// rows that are not whole vectors (see seg_tile_body, RAG)
template <typename T, int VEC, bool GATHER, int WMODE, bool ATOMIC, int NT, int RED = RED_SUM>
__global__ __launch_bounds__(kThreads) void seg_tile_rag_kernel(SegParams p) {
  seg_tile_body<T, VEC, GATHER, WMODE, ATOMIC, NT, RED, kU, true>(p);
}

// Tile bookkeeping + merge of the 2*NW wave partials of the narrow-row kernels (same rules as seg_tile_kernel:
// meta word, two carry slots per tile, rows behind the last key zero-filled by the last tile).
template <int F, int RED = RED_SUM, typename GapFill>
__device__ __forceinline__ void narrow_tile_epilogue(const SegParams &p, const int64_t *pkL, const int *pvL,
                                                     const float (*pL)[8], GapFill &gapfill, int te, int64_t tile) {
  constexpr int NW = kThreads / 64;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t ts = tile * (int64_t)te;
  const int64_t nnz = p.nnz, K = p.K;
  const int64_t *__restrict__ index = p.dst_index;
  float *__restrict__ dst = static_cast<float *>(p.dst);
  const int64_t rem = nnz - ts;
  const int n = rem < (int64_t)te ? (int)rem : te;
  const int64_t kprev_tile = ts > 0 ? index[ts - 1] : -1;
  const int64_t knext_tile = ts + n < nnz ? index[ts + n] : kNoKey;
  if (tid == 0) {
    const int64_t kf = index[ts], kl = index[ts + n - 1];
    const int64_t head = kf == kprev_tile;
    const int64_t single = head && kl == kf && knext_tile == kf;
    p.meta[tile] = (uint64_t)kf < (uint64_t)K ? (kf * 4 + head + 2 * single) : 0;
    if (tile == (int64_t)gridDim.x - 1) gapfill(kl + 1, K); // rows behind the last key
  }
  const bool active = lane < F;
  float *carry = static_cast<float *>(p.carry);
  for (int i = wv; i < 2 * NW; i += NW) {
    if (!pvL[i]) continue;
    const int64_t key = pkL[i];
    bool leader = true;
    if (!(i & 1) && i > 0) {
      const int pi = pvL[i - 1] ? i - 1 : i - 2;
      leader = pkL[pi] != key;
    }
    if (!leader) continue;
    float sum = active ? pL[i][lane] : 0.f;
    bool at_end = true;
    for (int j = i + 1; j < 2 * NW; ++j) {
      if (!pvL[j]) continue;
      if (pkL[j] != key) { at_end = false; break; }
      if (active) sum = red_op<float, RED>(sum, pL[j][lane]);
    }
    if (!active) continue;
    if (i == 0 && key == kprev_tile) carry[(tile * 2) * F + lane] = sum;
    else if (at_end && key == knext_tile && key != kNoKey) carry[(tile * 2 + 1) * F + lane] = sum;
    else if ((uint64_t)key < (uint64_t)K) dst[key * F + lane] = sum;
  }
}

// Narrow rows (F <= 8 fp32): lane-per-edge segmented scan.  This is the reference's "PR" idea
// (segreduce_pr_sorted_kernel, csrc/cuda/index_scatter_kernel.cuh:48-126: edges across lanes, segmented
// shuffle scan, atomicAdd per segment start) re-done for wave64 without atomics:
//   * a wave owns S steps of 64 consecutive edges; lane l of a step owns edge l: key and the F values are
//     loaded straight into registers, fully coalesced (64*F*4 contiguous bytes per step), all S steps in
//     flight at once; no LDS staging of keys;
//   * run starts come from one __ballot per step; the inclusive segmented scan is 6 __shfl_up rounds whose
//     "may I add lane l-d" test is a shift/mask of that ballot (no flag shuffles);
//   * the run that is open at the end of a step is carried to the next step in registers (readlane 63);
//   * a lane that ends a run stores it (plain store); the chunk's first and last run go to LDS and are
//     merged / carried exactly like the lane groups of seg_tile_kernel (same meta / carry format, same
//     seg_fixup_kernel), so every dst row is still written exactly once and results are deterministic.
template <int F, int S>
__global__ __launch_bounds__(kThreads) void seg_narrow_kernel(SegParams p) {
  constexpr int NW = kThreads / 64; // waves = "groups" of the tile merge
  __shared__ int64_t pkL[2 * NW];
  __shared__ int pvL[2 * NW];
  __shared__ float pL[2 * NW][8];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  publish_word(p);
  constexpr int te = kThreads * S;
  const int64_t tile = blockIdx.x;
  const int64_t ts = tile * (int64_t)te;
  const int64_t cs = ts + (int64_t)wv * (S * 64); // this wave's chunk
  const int64_t nnz = p.nnz, K = p.K;
  const int64_t *__restrict__ index = p.dst_index;
  const float *__restrict__ src = static_cast<const float *>(p.src);
  float *__restrict__ dst = static_cast<float *>(p.dst);

  // ---- all S steps in flight ---------------------------------------------------------------------------
  int64_t k[S];
  float v[S][F];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int64_t e = cs + s * 64 + lane;
    const bool ok = e < nnz;
    k[s] = ok ? index[e] : kNoKey;
    const int64_t ec = ok ? e : nnz - 1; // padding edges re-read the last edge, their run is discarded
    if constexpr (F == 1) v[s][0] = __builtin_nontemporal_load(src + ec);
    else if constexpr (F == 2) {
      const float2 t = *reinterpret_cast<const float2 *>(src + ec * 2);
      v[s][0] = t.x; v[s][1] = t.y;
    } else if constexpr (F == 4) {
      const float4 t = *reinterpret_cast<const float4 *>(src + ec * 4);
      v[s][0] = t.x; v[s][1] = t.y; v[s][2] = t.z; v[s][3] = t.w;
    } else {
#pragma unroll
      for (int i = 0; i < F; ++i) v[s][i] = src[ec * F + i];
    }
  }
  int64_t kbefore = -1; // key of the edge in front of the chunk
  if (cs > 0) kbefore = cs - 1 < nnz ? index[cs - 1] : kNoKey;

  auto gapfill = [&](int64_t lo, int64_t hi) { // executed by ONE lane
    if (hi <= lo || lo < 0 || hi > K) return;
    const int64_t cnt = hi - lo;
    if (cnt <= kGapInline) {
      for (int64_t r = lo; r < hi; ++r)
#pragma unroll
        for (int i = 0; i < F; ++i) dst[r * F + i] = 0.f;
    } else {
      const unsigned long long slot = atomicAdd(&p.ctrl[0], 1ull);
      if ((int64_t)slot < p.gap_cap) {
        p.gap_list[2 * slot] = lo;
        p.gap_list[2 * slot + 1] = cnt;
      }
    }
  };

  if (lane == 0) {
    pvL[2 * wv] = 1;
    pvL[2 * wv + 1] = 0;
  }
  float csum[F]; // the run that is open at the step boundary (wave-uniform)
#pragma unroll
  for (int i = 0; i < F; ++i) csum[i] = 0.f;
  int64_t klast = __shfl(k[0], 0, 64); // the chunk's first edge opens its first run (no run ends there)
  int nrun = 0;                        // runs of this chunk that have already ended
  if (lane == 0 && klast > kbefore + 1) gapfill(kbefore + 1, klast);
  raise_descent(p, lane == 0 && cs > 0 && cs < nnz && k[0] < kbefore && ((uint64_t)k[0] < (uint64_t)p.K || (uint64_t)kbefore < (uint64_t)p.K));

#pragma unroll
  for (int s = 0; s < S; ++s) {
    int64_t kp = __shfl_up(k[s], 1, 64);
    if (lane == 0) kp = klast;
    const bool h = k[s] != kp;
    const unsigned long long hb = __ballot(h);
    // (a real edge below its predecessor, one of the two keys inside [0, K) - see seg_tile_body; step 0's lane 0 was judged above)
    raise_descent(p, cs + s * 64 + lane < nnz && (s > 0 || lane > 0) && k[s] < kp && ((uint64_t)k[s] < (uint64_t)p.K || (uint64_t)kp < (uint64_t)p.K));
    if (h && k[s] > kp + 1) gapfill(kp + 1, k[s]);
    if ((hb & 1ull) && lane == 0) {
      // the run carried in ended exactly at the step boundary: nobody in this step continues it
      if (nrun == 0) {
#pragma unroll
        for (int i = 0; i < F; ++i) pL[2 * wv][i] = csum[i];
        pkL[2 * wv] = klast;
      } else if ((uint64_t)klast < (uint64_t)K) {
#pragma unroll
        for (int i = 0; i < F; ++i) dst[klast * F + i] = csum[i];
      }
    }
    float x[F];
#pragma unroll
    for (int i = 0; i < F; ++i) x[i] = v[s][i];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      // lane l may add lane l-d iff no run starts in (l-d, l]
      const bool can = lane >= d && ((hb >> (lane - d + 1)) & ((1ull << d) - 1ull)) == 0;
#pragma unroll
      for (int i = 0; i < F; ++i) {
        const float y = __shfl_up(x[i], d, 64);
        if (can) x[i] += y;
      }
    }
    const unsigned long long upto = hb & ((2ull << lane) - 1ull); // run starts at or before this lane
    if (upto == 0) {                                              // still inside the run carried in
#pragma unroll
      for (int i = 0; i < F; ++i) x[i] += csum[i];
    }
    const bool ends = lane < 63 && ((hb >> (lane + 1)) & 1ull);
    if (ends) {
      if (nrun + __popcll(upto) == 0) { // the chunk's first run: merged after the barrier
#pragma unroll
        for (int i = 0; i < F; ++i) pL[2 * wv][i] = x[i];
        pkL[2 * wv] = k[s];
      } else if ((uint64_t)k[s] < (uint64_t)K) {
        if constexpr (F == 2) *reinterpret_cast<float2 *>(dst + k[s] * 2) = float2{x[0], x[1]};
        else if constexpr (F == 4) *reinterpret_cast<float4 *>(dst + k[s] * 4) = float4{x[0], x[1], x[2], x[3]};
        else {
#pragma unroll
          for (int i = 0; i < F; ++i) dst[k[s] * F + i] = x[i];
        }
      }
    }
    // carry the open run to the next step
#pragma unroll
    for (int i = 0; i < F; ++i) csum[i] = __shfl(x[i], 63, 64);
    klast = __shfl(k[s], 63, 64);
    nrun += __popcll(hb);
  }
  if (lane == 0) { // the chunk's last run (or its only one)
    const int slot = 2 * wv + (nrun == 0 ? 0 : 1);
#pragma unroll
    for (int i = 0; i < F; ++i) pL[slot][i] = csum[i];
    pkL[slot] = klast;
    pvL[slot] = 1;
  }
  __syncthreads();

  narrow_tile_epilogue<F>(p, pkL, pvL, pL, gapfill, te, blockIdx.x);
}

// Narrow rows, second formulation ("lane-sequential"): the lane-per-edge scan above spends ~240 VALU + 40 LDS
// shuffle instructions per 64 edges on the segmented scan and its bookkeeping and is issue-bound at
// ~3 TB/s.  Here a lane owns E CONSECUTIVE edges instead:
//   * the tile's keys and values are loaded fully coalesced (16 B per lane per instruction, everything in
//     flight at once) and staged in LDS, each lane's chunk padded by 16 B (8 B for the keys) so that the
//     chunk reads that follow are at most 2-way bank conflicted;
//   * every lane walks its E edges in registers: a run that starts and ends inside the chunk is stored
//     straight to dst; the lane keeps its first-run partial ("head") and its last-run partial ("tail");
//   * ONE segmented shuffle scan per wave - not one per 64 edges - joins the tails with the heads that
//     continue them (the lane-level run structure comes from two ballots); the lane where a run ends stores
//     it; the wave's first and last run go to LDS and from there through the same tile merge, carry slots
//     and fix-up kernel as every other sorted kernel.
// ~25 instructions per 64 edges instead of ~490: the kernel is memory-bound.
// MODE 0: streamed rows (index_scatter), staged through LDS as described.  MODE 1 / 2: gathered rows
// (gather_scatter / gather_weight_scatter with a handful of features - SpMV at F = 1, label propagation,
// per-head scalars): nothing to stage, every lane gathers the rows of its own E edges (E independent random
// reads in flight per lane, 64 per instruction instead of the 8 that 8-lane groups with one active lane give).
template <int F, int E, int RED = RED_SUM, int MODE = 0>
__global__ __launch_bounds__(kThreads) void seg_lane_kernel(SegParams p) {
  static_assert(RED != RED_MEAN, "mean carries counts: served by seg_tile_kernel");
  constexpr int NW = kThreads / 64;
  constexpr int te = kThreads * E;
  constexpr int CW = E * F;      // floats per lane chunk (multiple of 4: E is)
  constexpr int CWP = CW + 4;    // padded chunk stride, floats
  constexpr int NV = CW / 4;     // float4 per lane chunk
  static_assert(E % 4 == 0, "chunks must be whole float4s");
  __shared__ __attribute__((aligned(16))) float valL[MODE == 0 ? kThreads * CWP : 4];
  __shared__ int64_t pkL[2 * NW];
  __shared__ int pvL[2 * NW];
  __shared__ float pL[2 * NW][8];
  typedef float f4_t __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  publish_word(p);
  int64_t tile = blockIdx.x;
  if constexpr (MODE != 0) { // contiguous tile ranges per XCD, as in the gather modes of seg_tile_kernel
    if (p.xcd_swizzle) {
      const int64_t nb = gridDim.x, per = nb / 8;
      if (tile < per * 8) tile = (tile % 8) * per + tile / 8;
    }
  }
  const int64_t ts = tile * (int64_t)te;
  const int64_t nnz = p.nnz, K = p.K;
  const int64_t *__restrict__ index = p.dst_index;
  const float *__restrict__ src = static_cast<const float *>(p.src);
  float *__restrict__ dst = static_cast<float *>(p.dst);
  const int64_t rem = nnz - ts;
  const int n = rem < (int64_t)te ? (int)rem : te;

  // ---- stage the tile: coalesced loads, padded per-lane chunks in LDS ---------------------------------
  if constexpr (MODE == 0) {
    const float *tsrc = src + ts * F;
    const int nfl = n * F; // valid floats of this tile
    f4_t v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int q = tid + kThreads * j; // float4 number inside the tile
      if (q * 4 + 3 < nfl) v[j] = __builtin_nontemporal_load(reinterpret_cast<const f4_t *>(tsrc) + q);
      else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[j][i] = q * 4 + i < nfl ? tsrc[q * 4 + i] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int q = tid + kThreads * j;
      *reinterpret_cast<f4_t *>(valL + (q / NV) * CWP + (q % NV) * 4) = v[j];
    }
  }
  // my E keys straight from global: 8*E contiguous bytes per lane, the E/2 16-B loads of a wave touch the same
  // 32 cache lines back to back (keys are a third or less of the traffic here; no LDS spent on them)
  int64_t mk[E];
  const int64_t e0 = ts + (int64_t)tid * E;
  const bool full = e0 + E <= nnz; // all my edges exist
  typedef long long l2_t __attribute__((ext_vector_type(2)));
  if (full) {
#pragma unroll
    for (int j = 0; j < E / 2; ++j) {
      const l2_t t = *reinterpret_cast<const l2_t *>(index + e0 + 2 * j);
      mk[2 * j] = t[0];
      mk[2 * j + 1] = t[1];
    }
  } else {
#pragma unroll
    for (int j = 0; j < E; ++j) mk[j] = e0 + j < nnz ? index[e0 + j] : kNoKey;
  }
  float vals[CW];
  if constexpr (MODE != 0) {
    // my E source rows: indices and weights are contiguous per lane, the rows are E independent gathers
    int64_t sr[E];
    if (full) {
#pragma unroll
      for (int j = 0; j < E / 2; ++j) {
        const l2_t t = *reinterpret_cast<const l2_t *>(p.src_index + e0 + 2 * j);
        sr[2 * j] = t[0];
        sr[2 * j + 1] = t[1];
      }
    } else {
#pragma unroll
      for (int j = 0; j < E; ++j) sr[j] = e0 + j < nnz ? p.src_index[e0 + j] : 0;
    }
    float wv_[E];
    if constexpr (MODE == 2) {
      const float *w = static_cast<const float *>(p.weight);
      if (full) {
#pragma unroll
        for (int j = 0; j < E / 4; ++j) {
          const f4_t t = *reinterpret_cast<const f4_t *>(w + e0 + 4 * j);
#pragma unroll
          for (int i = 0; i < 4; ++i) wv_[4 * j + i] = t[i];
        }
      } else {
#pragma unroll
        for (int j = 0; j < E; ++j) wv_[j] = e0 + j < nnz ? w[e0 + j] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < E; ++j) {
      int64_t r = sr[j];
      if ((uint64_t)r >= (uint64_t)p.src_rows) r = 0; // out-of-range gather index: memory-safe
      const float *row = src + r * F;
      if constexpr (F % 4 == 0) {
#pragma unroll
        for (int q = 0; q < F / 4; ++q) {
          const f4_t t = *reinterpret_cast<const f4_t *>(row + 4 * q);
#pragma unroll
          for (int i = 0; i < 4; ++i) vals[j * F + 4 * q + i] = t[i];
        }
      } else if constexpr (F == 2) {
        const float2 t = *reinterpret_cast<const float2 *>(row);
        vals[j * F] = t.x;
        vals[j * F + 1] = t.y;
      } else {
#pragma unroll
        for (int i = 0; i < F; ++i) vals[j * F + i] = row[i];
      }
      if constexpr (MODE == 2) {
#pragma unroll
        for (int i = 0; i < F; ++i) vals[j * F + i] *= wv_[j];
      }
    }
  }
  const int64_t kprev_tile = ts > 0 ? index[ts - 1] : -1;
  if constexpr (MODE == 0) __syncthreads();

  auto gapfill = [&](int64_t lo, int64_t hi) { // executed by ONE lane
    if (hi <= lo || lo < 0 || hi > K) return;
    const int64_t cnt = hi - lo;
    if (cnt <= kGapInline) {
      for (int64_t r = lo; r < hi; ++r)
#pragma unroll
        for (int i = 0; i < F; ++i) dst[r * F + i] = 0.f;
    } else {
      const unsigned long long slot = atomicAdd(&p.ctrl[0], 1ull);
      if ((int64_t)slot < p.gap_cap) {
        p.gap_list[2 * slot] = lo;
        p.gap_list[2 * slot + 1] = cnt;
      }
    }
  };
  auto store_row = [&](int64_t key, const float (&x)[F]) {
    if ((uint64_t)key >= (uint64_t)K) return;
    if constexpr (F == 4) *reinterpret_cast<float4 *>(dst + key * 4) = float4{x[0], x[1], x[2], x[3]};
    else if constexpr (F == 2) *reinterpret_cast<float2 *>(dst + key * 2) = float2{x[0], x[1]};
    else if constexpr (F == 8) {
      *reinterpret_cast<float4 *>(dst + key * 8) = float4{x[0], x[1], x[2], x[3]};
      *reinterpret_cast<float4 *>(dst + key * 8 + 4) = float4{x[4], x[5], x[6], x[7]};
    } else {
#pragma unroll
      for (int i = 0; i < F; ++i) dst[key * F + i] = x[i];
    }
  };

  // ---- the lane's E edges, sequentially in registers --------------------------------------------------
  const float *mv = MODE == 0 ? valL + tid * CWP : valL;
  // last key in front of my chunk: the previous lane's last key; a wave's lane 0 asks global memory
  int64_t kprev = __shfl_up(mk[E - 1], 1, 64);
  if (lane == 0) kprev = tid == 0 ? kprev_tile : (e0 - 1 < nnz ? index[e0 - 1] : kNoKey);
  if constexpr (MODE == 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const f4_t t = *reinterpret_cast<const f4_t *>(mv + 4 * j);
#pragma unroll
      for (int i = 0; i < 4; ++i) vals[4 * j + i] = t[i];
    }
  }
  int64_t cur = mk[0];
  if (cur > kprev + 1) gapfill(kprev + 1, cur);
  {
    const uint64_t uK = (uint64_t)p.K; // (a real edge below its predecessor, one of the two keys inside [0, K) - see seg_tile_body)
    bool desc = e0 > 0 && e0 < nnz && mk[0] < kprev && ((uint64_t)mk[0] < uK || (uint64_t)kprev < uK);
#pragma unroll
    for (int e = 1; e < E; ++e) desc = desc || (e0 + e < nnz && mk[e] < mk[e - 1] && ((uint64_t)mk[e] < uK || (uint64_t)mk[e - 1] < uK));
    raise_descent(p, desc);
  }
  float acc[F], head[F];
#pragma unroll
  for (int i = 0; i < F; ++i) { acc[i] = vals[i]; head[i] = 0.f; }
  bool single = true;  // all my edges so far belong to one run
  int64_t khead = cur; // key of my first run
#pragma unroll
  for (int e = 1; e < E; ++e) {
    const int64_t k = mk[e];
    if (k != cur) {
      if (single) {
#pragma unroll
        for (int i = 0; i < F; ++i) head[i] = acc[i];
        single = false;
      } else store_row(cur, acc); // a run that lies inside my chunk
      if (k > cur + 1) gapfill(cur + 1, k);
      cur = k;
#pragma unroll
      for (int i = 0; i < F; ++i) acc[i] = vals[e * F + i];
    } else {
#pragma unroll
      for (int i = 0; i < F; ++i) acc[i] = red_op<float, RED>(acc[i], vals[e * F + i]);
    }
  }

  // ---- one segmented scan per wave over the lanes' open runs ---------------------------------------------
  // partial sequence in edge order: a single-run lane contributes [only], any other lane [head, tail].
  const bool cont = lane > 0 && khead == kprev;            // my first run continues the previous lane's last run
  const unsigned long long sb = __ballot(single);
  const unsigned long long hb = __ballot(!(single && cont)); // scan heads: the open run restarts at this lane
  float x[F];
#pragma unroll
  for (int i = 0; i < F; ++i) x[i] = acc[i];
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const bool can = lane >= d && ((hb >> (lane - d + 1)) & ((1ull << d) - 1ull)) == 0;
#pragma unroll
    for (int i = 0; i < F; ++i) {
      const float y = __shfl_up(x[i], d, 64);
      if (can) x[i] = red_op<float, RED>(y, x[i]); // earlier edges on the left: the order the reference reduces in
    }
  }
  float xprev[F]; // open-run total at the end of the previous lane
#pragma unroll
  for (int i = 0; i < F; ++i) xprev[i] = __shfl_up(x[i], 1, 64);
  // does my FIRST partial belong to the wave's first run?  (every lane border before me continues, and every
  // lane before me is single-run)
  const unsigned long long brk = __ballot(lane > 0 && !(cont && ((sb >> (lane - 1)) & 1ull)));
  const bool wave_first = (brk & ((2ull << lane) - 1ull)) == 0;
  const unsigned long long cb = __ballot(cont);
  if (lane == 0) {
    pvL[2 * wv] = 1;
    pvL[2 * wv + 1] = 0;
  }
  if (!single) { // my first run ends inside my chunk
    float r[F];
#pragma unroll
    for (int i = 0; i < F; ++i) r[i] = cont ? red_op<float, RED>(xprev[i], head[i]) : head[i];
    if (wave_first) {
#pragma unroll
      for (int i = 0; i < F; ++i) pL[2 * wv][i] = r[i];
      pkL[2 * wv] = khead;
    } else store_row(khead, r);
  }
  const bool my_only_is_wave_first = single && wave_first;
  if (lane < 63) {
    if (!((cb >> (lane + 1)) & 1ull)) { // my last run ends at the end of my chunk
      if (my_only_is_wave_first) {
#pragma unroll
        for (int i = 0; i < F; ++i) pL[2 * wv][i] = x[i];
        pkL[2 * wv] = cur;
      } else store_row(cur, x);
    }
  } else { // the run that is open at the end of the wave
    const int slot = 2 * wv + (my_only_is_wave_first ? 0 : 1);
#pragma unroll
    for (int i = 0; i < F; ++i) pL[slot][i] = x[i];
    pkL[slot] = cur;
    pvL[slot] = 1;
  }
  __syncthreads();
  narrow_tile_epilogue<F, RED>(p, pkL, pvL, pL, gapfill, te, tile);
}

// Unsorted index, few output rows: LDS-binned atomics.  When the whole [K, F] output fits in LDS,
// every block accumulates its share of the edges into an LDS copy with ds_add (LDS float atomics:
// no global contention, any key order), then adds its copy to dst with ONE global atomic per
// element.  Global atomic traffic drops from nnz*F to blocks*K*F, and the pathological "every
// workgroup adds into the same few rows" case (14x slower than spread atomics on MI355X) never
// happens.  dst is zero-filled by the launcher.
template <typename T>
__global__ __launch_bounds__(kThreads) void seg_lds_bin_kernel(const int64_t *__restrict__ index,
                                                               const T *__restrict__ src,
                                                               T *__restrict__ dst, int64_t nnz,
                                                               int64_t F, int64_t K, int lpr_log2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T *tab = reinterpret_cast<T *>(smem);
  const int64_t n = K * F;
  for (int64_t i = threadIdx.x; i < n; i += kThreads) tab[i] = T(0);
  __syncthreads();
  const int lpr = 1 << lpr_log2;
  const int ng = kThreads >> lpr_log2;
  const int g = threadIdx.x >> lpr_log2, c = threadIdx.x & (lpr - 1);
  // contiguous chunk of edges per block (coalesced rows), lane group per row
  const int64_t per = (nnz + gridDim.x - 1) / gridDim.x;
  const int64_t e0 = (int64_t)blockIdx.x * per;
  const int64_t e1 = e0 + per < nnz ? e0 + per : nnz;
  for (int64_t e = e0 + g; e < e1; e += ng) {
    const int64_t k = index[e];
    if ((uint64_t)k >= (uint64_t)K) continue;
    for (int64_t f = c; f < F; f += lpr) atomicAdd(&tab[k * F + f], src[e * F + f]);
  }
  __syncthreads();
  for (int64_t i = threadIdx.x; i < n; i += kThreads) {
    const T v = tab[i];
    if (v != T(0)) atomicAdd(dst + i, v);
  }
}

// Long-run inputs only (few keys / hub segments; the launcher decides from nnz / K): one block per window of
// 64 tiles.  If every tile of the window [64w, 64w+63] is `single` (one run that enters and leaves the tile),
// then whenever tile 64w belongs to a chain, tiles 64w+1 .. 64w+64 join it: their slot-0 carries are summed
// here in a fixed order, so that seg_fixup_kernel walks a hub chain 64 tiles per row instead of tile by tile
// (10 M edges on ONE key: fix-up 0.81 ms -> a few tens of us).
template <typename T, int RED = RED_SUM>
__global__ __launch_bounds__(kThreads) void seg_wsum_kernel(SegParams p, int64_t num_tiles) {
  using A = typename AccOf<T>::type;
  const int64_t w = blockIdx.x;
  const int64_t t0 = w * 64;
  __shared__ int s_all;
  if (threadIdx.x < 64) {
    const int64_t t = t0 + threadIdx.x;
    const bool single = t < num_tiles && (p.meta[t] & 2);
    const unsigned long long S = __ballot(single);
    if (threadIdx.x == 0) s_all = (S == ~0ull) && (t0 + 64 < num_tiles);
  }
  __syncthreads();
  if (!s_all) {
    if (threadIdx.x == 0) p.wflag[w] = 0;
    return;
  }
  const A *carry = static_cast<const A *>(p.carry);
  A *wsum = static_cast<A *>(p.wsum);
  const int64_t F = p.F;
  for (int64_t f = threadIdx.x; f < F; f += kThreads) {
    A s = red_ident<A, RED>();
#pragma unroll 16
    for (int i = 1; i <= 64; ++i) s = red_op<A, RED>(s, carry[((t0 + i) * 2) * F + f]);
    wsum[w * F + f] = s;
  }
  if (threadIdx.x == 0) {
    if constexpr (RED == RED_MEAN) {
      int64_t n = 0;
      for (int i = 1; i <= 64; ++i) n += p.ccnt[(t0 + i) * 2];
      p.wcnt[w] = n;
    }
    p.wflag[w] = 1;
  }
}

// ---- descent guard: repair of a call whose index was NOT ascending --------------------------------------------------
// The reference's "sorted" kernels tolerate a wrong `sorted` promise because every run is flushed with atomicAdd into a
// zeroed dst (csrc/cuda/index_scatter_kernel.cuh:180,197).  The atomic-free kernels here do not - and although the host
// layer probes every index once per content, a write that does not move the tensor's version counter (`.data`, DLPack,
// another extension's kernel) leaves a stale "ascending" fact behind; callers of the C ABI may simply pass sorted=1 on
// faith.  So the staging loops ballot "key below its predecessor" (free: the keys are in registers there) and this
// routine, entered by EVERY workgroup of the fix-up launch when the flag is up, redoes the call the reference's way:
//   1. all workgroups zero dst (grid-stride), write their L2 back, take a ticket;
//   2. the workgroup holding the last ticket opens phase 2; workgroups still waiting for that (bounded wait - a grid
//      larger than the chip must let its later workgroups in, so a waiter gives up after a while and only the
//      workgroups resident at the end help; the opener alone would be enough: no co-residency is ever REQUIRED) then
//      claim chunks of 256 edges from a counter and add every edge's row into dst with float atomics;
//   3. the last workgroup to leave re-zeroes the control words and raises the host's alarm word, so the host layer
//      drops its facts about index tensors (the next call probes again and takes the sort path).
// sum over fp32 / fp64 (exactly what the reference's atomics cover); any other reduction or a 16-bit dtype has no float
// atomic to fall back on: its output is filled with NaN and the alarm says so - loud, never silently wrong.
template <typename T, int RED>
__device__ __forceinline__ void repair_call(const SegParams &p) {
  constexpr bool kCanRepair = RED == RED_SUM && (std::is_same<T, float>::value || std::is_same<T, double>::value);
  __shared__ long long s_word;
  const int tid = threadIdx.x;
  const int64_t G = gridDim.x;
  T *dst = static_cast<T *>(p.dst);
  const int64_t F = p.F, K = p.K, total = K * F;
  const T fill = kCanRepair ? T(0) : (T)NAN;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + tid; i < total; i += G * kThreads) dst[i] = fill;
  if constexpr (kCanRepair) {
    __threadfence(); // the zeros must be in memory before any workgroup's (memory-side) atomics
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      long long go = 0;
      if (atomicAdd(&p.ctrl[kCtrlZeroed], 1ull) == (unsigned long long)G - 1) {
        atomicExch(&p.ctrl[kCtrlGo], 1ull);
        go = 1;
      } else {
        for (int tries = 0; tries < 4096 && !go; ++tries) {
          go = atomicAdd(&p.ctrl[kCtrlGo], 0ull) != 0ull;
          if (!go) __builtin_amdgcn_s_sleep(64);
        }
      }
      s_word = go;
    }
    __syncthreads();
    const bool helper = s_word != 0;
    __syncthreads();
    if (helper) {
      const T *src = static_cast<const T *>(p.src);
      const T *w = static_cast<const T *>(p.weight);
      const int64_t nch = (p.nnz + 255) >> 8;
      for (;;) {
        if (tid == 0) s_word = (long long)atomicAdd(&p.ctrl[kCtrlChunk], 1ull);
        __syncthreads();
        const int64_t ch = s_word;
        __syncthreads();
        if (ch >= nch) break;
        const int64_t e0 = ch << 8;
        const int64_t n = p.nnz - e0 < 256 ? p.nnz - e0 : 256;
        for (int64_t idx = tid; idx < n * F; idx += kThreads) {
          const int64_t el = idx / F, f = idx - el * F, e = e0 + el;
          const int64_t k = p.dst_index[e];
          if ((uint64_t)k >= (uint64_t)K) continue;
          T v;
          if (p.mode == 0) v = src[e * F + f];
          else {
            int64_t r = p.src_index[e];
            if ((uint64_t)r >= (uint64_t)p.src_rows) r = 0; // as the tile kernel: memory-safe
            v = src[r * F + f];
            if (p.mode == 2) v *= w[e];
            else if (p.mode == 3) v *= w[e * p.H + f / p.Fh];
            else if (p.mode == 4) v *= w[(f / p.Fh) * p.nnz + e];
          }
          if constexpr (kCanRepair) atomicAdd(dst + k * F + f, v);
        }
      }
    }
  }
  __syncthreads();
  if (tid == 0 && atomicAdd(&p.ctrl[kCtrlDone], 1ull) == (unsigned long long)G - 1) {
    // every workgroup has read the flag and finished its share: leave the control words zero for the next call
    for (int i = 0; i <= kCtrlDeferred; ++i) atomicExch(&p.ctrl[i], 0ull);
    if (p.alarm) __hip_atomic_store(p.alarm + (kCanRepair ? 0 : 1), (int64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Second launch (= the only cross-workgroup ordering the sorted path needs):
//  (a) one LANE GROUP per tile (64/LPR tiles per wave): if the tile holds the FIRST carry of a
//      chain (its head run continues from the previous tile, and that tile is where the run
//      starts) the row  dst[k] = tail partial of tile t-1 + head partial of tile t  is written with
//      a plain store.  Every address (two meta words, two carry rows) is independent of loaded
//      data, so the whole pass is ONE memory round trip; no read-modify-write of dst.
//      Hub runs that cover whole follow-on tiles are rare: they are finished wave-cooperatively
//      afterwards (64 meta words per ballot window, rows summed by the wave's lane groups in
//      parallel, xor-shuffle combine in a fixed order => deterministic);
//  (b) zero-fill the large gaps the tile kernel recorded;
//  (c) the last block to finish re-zeroes the two control words for the next call.
template <typename T, int RED = RED_SUM>
__global__ __launch_bounds__(kThreads) void seg_fixup_kernel(SegParams p, int64_t num_tiles) {
  constexpr bool MEAN = RED == RED_MEAN;
  using A = typename AccOf<T>::type;
  const int lane = threadIdx.x & 63;
  const int lpr = 1 << p.lpr_log2;
  const int R = 64 >> p.lpr_log2; // lane groups per wave
  const int gq = lane >> p.lpr_log2;
  const int c = lane & (lpr - 1);
  T *dst = static_cast<T *>(p.dst);
  const A *carry = static_cast<const A *>(p.carry);
  const int64_t F = p.F;
  __shared__ unsigned long long s_ngap;
  if (threadIdx.x == 0) s_ngap = p.ctrl[0];

  constexpr int J = 4;
  const int64_t wave_t0 = ((int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6)) * R;
  const int64_t t = wave_t0 + gq; // this lane group's tile
  // Hand-off mode: the tile kernel has finished the straddling runs itself (seg_tile_kernel, "hand-off").  This launch then
  // only tidies up - it lowers the tiles' flags (a replayed hipGraph re-uses the call's tag) - and leaves, unless the tile
  // kernel left something behind: large gaps to fill, a descent to repair, or runs it gave up waiting for (`deferred`: then
  // the classic pass below redoes every straddling run from the carry rows, which is idempotent).
  bool do_chains = true;
  unsigned long long deferred = 0ull;
  if (p.handoff) {
    if (c == 0 && t < num_tiles) p.flags[t] = 0ull;
    deferred = p.ctrl[kCtrlDeferred];
    if ((p.ctrl[kCtrlGaps] | p.ctrl[kCtrlDescent] | deferred) == 0ull) return;
    do_chains = deferred != 0ull;
  }
  const bool valid = do_chains && t >= 1 && t < num_tiles;
  const int64_t tc = valid ? t : 1 < num_tiles ? 1 : 0; // clamped for the speculative loads
  const int64_t m = valid ? p.meta[tc] : 0;
  const int64_t mp = valid ? p.meta[tc - 1] : 2;
  A cv[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int64_t f = (int64_t)j * lpr + c;
    cv[j] = (valid && f < F) ? red_op<A, RED>(carry[((tc - 1) * 2 + 1) * F + f], carry[(tc * 2) * F + f]) : A(0);
  }
  // descent guard: the tile kernel found the index not ascending -> redo the whole call with atomics (rare; wave-uniform)
  if (p.ctrl[kCtrlDescent] != 0ull) {
    repair_call<T, RED>(p);
    return;
  }
  // Hand-off runs that the tile kernel gave up waiting for (`deferred`): redone HERE exactly as the tile kernel would have done them
  // - the tile in which a straddling run ENDS walks back over its predecessors, nearest first, the partial rows meet in float64
  // and are rounded once - so a call gives the same BITS whether a flag was seen in time or not (timing must never decide a
  // result; round 3 redid them the classic way: fp32 adds in another association, last-ulp differences between two calls).
  // Sequential per run (a hub over thousands of tiles is walked by one lane group): this is the rare path.
  const bool ho_redo = p.handoff && deferred != 0ull;
  if constexpr (std::is_same<A, float>::value) {
    if (ho_redo && valid && (m & 1) && !(m & 2)) {
      const int64_t kend = m >> 2;
      for (int64_t fb = 0; fb < F; fb += (int64_t)lpr * J) {
        double acc[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const int64_t f = fb + (int64_t)j * lpr + c;
          acc[j] = f < F ? (double)carry[(t * 2) * F + f] : 0.0;
        }
        int64_t cnt = 0;
        if constexpr (MEAN) cnt = p.ccnt[t * 2];
        for (int64_t jt = t - 1; jt >= 0; --jt) {
          const bool single = (p.meta[jt] & 2) != 0;
          const int64_t row = (jt * 2 + (single ? 0 : 1)) * F;
#pragma unroll
          for (int j = 0; j < J; ++j) {
            const int64_t f = fb + (int64_t)j * lpr + c;
            if (f < F) acc[j] = red_op<double, RED>((double)carry[row + f], acc[j]);
          }
          if constexpr (MEAN) cnt += p.ccnt[jt * 2 + (single ? 0 : 1)];
          if (!single) break; // the tile in which the run starts
        }
        double inv = 1.0;
        if constexpr (MEAN) inv = 1.0 / (double)cnt;
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const int64_t f = fb + (int64_t)j * lpr + c;
          if (f < F && (uint64_t)kend < (uint64_t)p.K) dst[kend * F + f] = (T)(A)(MEAN ? acc[j] * inv : acc[j]);
        }
      }
    }
  }
  const bool first = !ho_redo && valid && (m & 1) && !(mp & 2);
  const int64_t k = m >> 2;
  if (first && !(m & 2)) {
    A inv_div = A(1);
    if constexpr (MEAN) inv_div = A(p.ccnt[(t - 1) * 2 + 1] + p.ccnt[t * 2]);
    for (int64_t fb = 0; fb < F; fb += (int64_t)lpr * J) {
      if (fb > 0) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const int64_t f = fb + (int64_t)j * lpr + c;
          cv[j] = f < F ? red_op<A, RED>(carry[((t - 1) * 2 + 1) * F + f], carry[(t * 2) * F + f]) : A(0);
        }
      }
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int64_t f = fb + (int64_t)j * lpr + c;
        if (f < F) dst[k * F + f] = (T)(MEAN ? cv[j] / inv_div : cv[j]);
      }
    }
  }

  // hub chains (wave-uniform loop over the groups that found one)
  unsigned long long hubs = __ballot(first && (m & 2) && c == 0);
  while (hubs) {
    const int src_lane = __builtin_ctzll(hubs);
    hubs &= hubs - 1;
    const int64_t th = __shfl(t, src_lane, 64);
    const int64_t kh = __shfl(k, src_lane, 64);
    for (int64_t fb = 0; fb < F; fb += (int64_t)lpr * J) {
      A hv[J];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int64_t f = fb + (int64_t)j * lpr + c;
        hv[j] = (gq == 0 && f < F) ? red_op<A, RED>(carry[((th - 1) * 2 + 1) * F + f], carry[(th * 2) * F + f])
                                   : red_ident<A, RED>();
      }
      int64_t hcnt = 0; // mean: edges of the whole chain (every lane computes the same value)
      if constexpr (MEAN) hcnt = p.ccnt[(th - 1) * 2 + 1] + p.ccnt[th * 2];
      // tile x+1 joins the chain for as long as tile x is `single`; 64 tiles per window.  With window sums
      // (p.wflag) the first window is cut at the next multiple of 64, whole windows are then taken 64 at a
      // time from seg_wsum_kernel's rows, and the tail is finished tile by tile again.
      const int64_t nwin = (num_tiles + 63) >> 6;
      int64_t wb = th;
      for (;;) {
        const int lim = p.wflag ? 64 - (int)(wb & 63) : 64;
        const int64_t wt = wb + lane;
        const int64_t mcur = p.meta[wt < num_tiles ? wt : num_tiles - 1];
        const unsigned long long S = __ballot((mcur & 2) && (wt < num_tiles) && lane < lim);
        const int nn = ~S ? __builtin_ctzll(~S) : 64; // tiles wb+1 .. wb+nn join the chain
        const int64_t hi = wb + nn;
        for (int64_t i = wb + 1 + gq; i <= hi; i += (int64_t)R * 4) {
          A cr[4][J];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int64_t tt = i + (int64_t)q * R;
#pragma unroll
            for (int j = 0; j < J; ++j) {
              const int64_t f = fb + (int64_t)j * lpr + c;
              cr[q][j] = (tt <= hi && f < F) ? carry[(tt * 2) * F + f] : red_ident<A, RED>();
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < J; ++j) hv[j] = red_op<A, RED>(hv[j], cr[q][j]);
        }
        if constexpr (MEAN) {
          for (int64_t tt = wb + 1; tt <= hi; ++tt) hcnt += p.ccnt[tt * 2];
        }
        if (nn < lim) break;
        wb += nn;
        if (p.wflag) { // wb is a multiple of 64 and tile wb has joined
          const A *wsum = static_cast<const A *>(p.wsum);
          for (;;) {
            const int64_t w0 = wb >> 6;
            const int64_t wi = w0 + lane;
            const unsigned long long Wm = __ballot(wi < nwin && p.wflag[wi] != 0);
            const int nW = ~Wm ? __builtin_ctzll(~Wm) : 64; // windows w0 .. w0+nW-1 are one run each
            for (int64_t i = w0 + gq; i < w0 + nW; i += (int64_t)R * 4) {
              A cr[4][J];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int64_t ww = i + (int64_t)q * R;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                  const int64_t f = fb + (int64_t)j * lpr + c;
                  cr[q][j] = (ww < w0 + nW && f < F) ? wsum[ww * F + f] : red_ident<A, RED>();
                }
              }
#pragma unroll
              for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < J; ++j) hv[j] = red_op<A, RED>(hv[j], cr[q][j]);
            }
            if constexpr (MEAN) {
              for (int64_t ww = w0; ww < w0 + nW; ++ww) hcnt += p.wcnt[ww];
            }
            wb += (int64_t)64 * nW;
            if (nW < 64) break;
          }
        }
      }
      for (int off = lpr; off < 64; off <<= 1) {
#pragma unroll
        for (int j = 0; j < J; ++j) hv[j] = red_op<A, RED>(hv[j], __shfl_xor(hv[j], off, 64));
      }
      if (gq == 0) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const int64_t f = fb + (int64_t)j * lpr + c;
          if (f < F) dst[kh * F + f] = (T)(MEAN ? hv[j] / A(hcnt) : hv[j]);
        }
      }
    }
  }

  __syncthreads();
  // Common case: no large gap was recorded -> nothing to fill and nothing to reset (no atomics).
  if (s_ngap == 0 && deferred == 0ull) return;
  const int64_t ngap = (int64_t)s_ngap < p.gap_cap ? (int64_t)s_ngap : p.gap_cap;
  const int64_t gtid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const int64_t gsz = (int64_t)gridDim.x * kThreads;
  for (int64_t gidx = 0; gidx < ngap; ++gidx) {
    const int64_t lo = p.gap_list[2 * gidx];
    const int64_t total = p.gap_list[2 * gidx + 1] * F;
    T *base = dst + lo * F;
    for (int64_t i = gtid; i < total; i += gsz) base[i] = T(0);
  }
  // every block has read ctrl[0] before it takes a ticket, so the last ticket may clear it
  if (threadIdx.x == 0) {
    const unsigned long long prev = atomicAdd(&p.ctrl[1], 1ull);
    if (prev == (unsigned long long)gridDim.x - 1) {
      atomicExch(&p.ctrl[0], 0ull);
      atomicExch(&p.ctrl[1], 0ull);
      atomicExch(&p.ctrl[kCtrlDeferred], 0ull);
    }
  }
}

// out[e] = <m1[d[e]], m2[s[e]]>; one lane group of LPR lanes per edge, xor-shuffle reduce.
// Every block owns a CONTIGUOUS chunk of edges (dst-sorted edges share their m1 row, neighbours share
// m2 rows on graphs with locality) and chunks are handed to the XCDs in contiguous ranges (see the
// gather modes of seg_tile_kernel): both operands then hit in the XCD's L2 instead of 8 L2s.
// Multi-head form (H > 1; d/dweight of mh_spmm): rows are [H, F]; an "edge" of the loop is a pair (edge e, head h) = q, q = e * H + h,
// its operands the F values of head h of rows d[e] / s[e], its result out[e * H + h] (edge-major) or out[h * nnz_e + e] (head-major).
// `nnz` counts pairs, `stride` = H * F elements between rows.  H = 1: the plain SDDMM.
template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void sddmm_coo_kernel(const int64_t *src_index,
                                                             const int64_t *dst_index,
                                                             const T *m1, const T *m2, T *out,
                                                             int64_t nnz, int64_t F,
                                                             int64_t rows1, int64_t rows2,
                                                             int lpr_log2, int64_t chunk, int xcd_swizzle,
                                                             int H, int64_t stride, int head_major) {
  const int lpr = 1 << lpr_log2;
  const int ng = kThreads >> lpr_log2;
  const int g = threadIdx.x >> lpr_log2;
  const int c = threadIdx.x & (lpr - 1);
  int64_t blk = blockIdx.x;
  if (xcd_swizzle) {
    const int64_t nb = gridDim.x, per = nb / 8;
    if (blk < per * 8) blk = (blk % 8) * per + blk / 8;
  }
  const int64_t e0 = blk * chunk;
  const int64_t e1 = e0 + chunk < nnz ? e0 + chunk : nnz;
  using A = typename AccOf<T>::type;
  constexpr int UE = 4; // edges in flight per lane group: indices first, then all rows, then the dots
  for (int64_t eb = e0 + (int64_t)g * UE; eb < e1; eb += (int64_t)ng * UE) {
    int64_t r1[UE], r2[UE];
    int64_t hoff[UE];       // element offset of the pair's head inside a row (0 for H = 1)
#pragma unroll
    for (int u = 0; u < UE; ++u) {
      const int64_t q = eb + u < e1 ? eb + u : e1 - 1;
      int64_t e = q;
      hoff[u] = 0;
      if (H > 1) {
        e = q / H;
        hoff[u] = (q - e * H) * F;
      }
      r1[u] = dst_index[e];
      r2[u] = src_index[e];
      if ((uint64_t)r1[u] >= (uint64_t)rows1 || (uint64_t)r2[u] >= (uint64_t)rows2) r1[u] = -1;
    }
    A s[UE];
#pragma unroll
    for (int u = 0; u < UE; ++u) s[u] = A(0);
    // dst-sorted edges: the UE edges of a group usually share their m1 row - load it once (wave-uniform
    // test per group would diverge; a per-group predicate on the load is enough)
    bool same = true;
#pragma unroll
    for (int u = 1; u < UE; ++u) same = same && (r1[u] == r1[0]) && (hoff[u] == hoff[0]);
    for (int64_t f = (int64_t)c * VEC; f < F; f += (int64_t)lpr * VEC) {
      A x[UE][VEC], y[UE][VEC];
      load_vec<T, VEC, false>(m1 + (r1[0] < 0 ? 0 : r1[0]) * stride + hoff[0] + f, x[0]);
#pragma unroll
      for (int u = 0; u < UE; ++u) {
        const int64_t a1 = r1[u] < 0 ? 0 : r1[u], a2 = r1[u] < 0 ? 0 : r2[u];
        if (u > 0) {
          if (same) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) x[u][i] = x[0][i];
          } else {
            load_vec<T, VEC, false>(m1 + a1 * stride + hoff[u] + f, x[u]);
          }
        }
        load_vec<T, VEC, false>(m2 + a2 * stride + hoff[u] + f, y[u]);
      }
#pragma unroll
      for (int u = 0; u < UE; ++u)
#pragma unroll
        for (int i = 0; i < VEC; ++i) s[u] += x[u][i] * y[u][i];
    }
#pragma unroll
    for (int u = 0; u < UE; ++u) {
      A v = r1[u] < 0 ? A(0) : s[u];
      for (int o = lpr >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (c == 0 && eb + u < e1) {
        const int64_t q = eb + u;
        if (H > 1 && head_major) {
          const int64_t e = q / H;
          out[(q - e * H) * (nnz / H) + e] = (T)v;
        } else {
          out[q] = (T)v;
        }
      }
    }
  }
}

#if GEOT_SEG_PART == 0 // (the small kernels: part 0 only)
// Row rule + precondition probe in one pass over the index: out[0] = index[nnz-1] (the reference's
// index[-1].item() read, csrc/index_scatter.cpp:30), out[1] += number of descents index[i] > index[i+1]
// found (0 <=> ascending, the precondition of the atomic-free kernels).  out[1] is zeroed by the launcher.
__global__ __launch_bounds__(kThreads) void index_probe_kernel(const int64_t *__restrict__ index, int64_t nnz,
                                                                 int64_t *__restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  int bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i + 1 < nnz; i += stride)
    bad += index[i] > index[i + 1];
  if (__ballot(bad != 0) != 0) { // rare on the inputs this is for: one atomic per wave that saw a descent
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) bad += __shfl_xor(bad, d, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(reinterpret_cast<unsigned long long *>(out + 1), (unsigned long long)bad);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = index[nnz - 1];
}

// small index: one workgroup, no atomics, no memset (one launch instead of two on a launch-bound call)
__global__ __launch_bounds__(kThreads) void index_probe_small_kernel(const int64_t *__restrict__ index, int64_t nnz,
                                                                       int64_t *__restrict__ out) {
  __shared__ int part[kThreads / 64];
  int bad = 0;
  for (int64_t i = threadIdx.x; i + 1 < nnz; i += kThreads) bad += index[i] > index[i + 1];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) bad += __shfl_xor(bad, d, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = bad;
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < kThreads / 64; ++w) tot += part[w];
    out[0] = index[nnz - 1];
    out[1] = tot;
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void gather_rows_kernel(const int64_t *index, const T *src,
                                                               T *dst, int64_t nnz, int64_t F,
                                                               int64_t src_rows, int lpr_log2) {
  const int lpr = 1 << lpr_log2;
  const int ng = kThreads >> lpr_log2;
  const int g = threadIdx.x >> lpr_log2;
  const int c = threadIdx.x & (lpr - 1);
  for (int64_t e = (int64_t)blockIdx.x * ng + g; e < nnz; e += (int64_t)gridDim.x * ng) {
    const int64_t r = index[e];
    const bool ok = (uint64_t)r < (uint64_t)src_rows;
    for (int64_t f = (int64_t)c * VEC; f < F; f += (int64_t)lpr * VEC) {
      typename AccOf<T>::type x[VEC];
      if (ok) load_vec<T, VEC, false>(src + r * F + f, x);
      else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) x[i] = 0;
      }
      store_vec<T, VEC>(dst + e * F + f, x);
    }
  }
}

// CSR row pointers -> per-edge row ids (dst_index) so that csr_gws can run on the tile kernel.
// One lane group of 16 per row, rows round-robin over groups; a hub row is filled by its group in
// 16-edge steps (the pass moves 8 B per edge: ~3 % of a gws call at F=128).
// Backward of the max / min aggregation of the gather ops (PyG's aggr='max' on gather_scatter / gather_weight_scatter): the gradient of
// out[d, f] = max_e m(e, f), m(e, f) = w_e * x[s_e, f], goes to the messages that ATTAIN the extremum, divided evenly among ties
// (torch.scatter_reduce's rule).  Two passes over the dst-sorted list, a lane group per edge, three row gathers each (x[s_e], out[d_e],
// grad[d_e]; consecutive edges share d):
//   PASS 0  ties[d, f] += 1 for every attaining message                                  (float atomics: one per selected element)
//   PASS 1  grad_src[s_e, f] += w_e * grad[d_e, f] / ties[d_e, f]   for attaining messages (float atomics: ~K x F of them in all)
//           grad_weight[e]    = sum_f x[s_e, f] * grad[d_e, f] / ties[d_e, f] over them   (plain store)
// A message attains when it equals out[d, f] IN THE STORAGE TYPE - exactly how the forward produced it (one multiply, the same
// rounding).  NaN never compares equal: a row whose extremum is NaN passes no gradient.  The sums into grad_src are atomic adds: their
// order is not fixed (a source with several selected edges of one column), unlike every forward kernel here.  fp32 / fp64.
template <typename T, int PASS>
__global__ __launch_bounds__(kThreads) void select_backward_kernel(const int64_t *__restrict__ src_index, const int64_t *__restrict__ dst_index,
                                                                   const T *__restrict__ weight, const T *__restrict__ x, const T *__restrict__ out,
                                                                   const T *__restrict__ grad, T *__restrict__ ties, T *__restrict__ grad_src,
                                                                   T *__restrict__ grad_weight, int64_t nnz, int64_t F, int64_t src_rows, int64_t K,
                                                                   int lpr_log2) {
  const int lpr = 1 << lpr_log2, ng = kThreads >> lpr_log2;
  const int g = threadIdx.x >> lpr_log2, c = threadIdx.x & (lpr - 1);
  for (int64_t e = (int64_t)blockIdx.x * ng + g; e < nnz; e += (int64_t)gridDim.x * ng) {
    const int64_t s = src_index[e], d = dst_index[e];
    const bool ok = (uint64_t)s < (uint64_t)src_rows && (uint64_t)d < (uint64_t)K;
    const T w = weight ? weight[e] : T(1);
    T dot = T(0);
    if (ok) {
      for (int64_t f = c; f < F; f += lpr) {
        const T xv = x[s * F + f];
        const T m = weight ? xv * w : xv;
        if (m == out[d * F + f]) {
          if constexpr (PASS == 0) atomicAdd(ties + d * F + f, T(1));
          else {
            const T gshare = grad[d * F + f] / ties[d * F + f];
            atomicAdd(grad_src + s * F + f, w * gshare);
            dot += xv * gshare;
          }
        }
      }
    }
    if constexpr (PASS == 1) {
      if (grad_weight) {
        for (int o = lpr >> 1; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
        if (c == 0) grad_weight[e] = dot;
      }
    }
  }
}

template <typename PTR>
__global__ __launch_bounds__(kThreads) void csr_expand_kernel(const PTR *__restrict__ indptr,
                                                              int64_t nrow, int64_t nnz,
                                                              int64_t *__restrict__ dst_index) {
  const int g = threadIdx.x >> 4, c = threadIdx.x & 15;
  for (int64_t r = (int64_t)blockIdx.x * (kThreads / 16) + g; r < nrow;
       r += (int64_t)gridDim.x * (kThreads / 16)) {
    int64_t lo = (int64_t)indptr[r], hi = (int64_t)indptr[r + 1];
    lo = lo < 0 ? 0 : lo;
    hi = hi > nnz ? nnz : hi;
    for (int64_t e = lo + c; e < hi; e += 16) dst_index[e] = r;
  }
}

// COO row ids -> CSR row pointers.  hist pass (int32 atomics: the reference's coo_to_hist kernel,
// geot/triton/coo_to_csr.py:14-26, does the same) ...
__global__ __launch_bounds__(kThreads) void coo_hist_kernel(const int64_t *__restrict__ row, int64_t nnz,
                                                            int64_t nrow, int *__restrict__ hist) {
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * kThreads) {
    const int64_t r = row[e];
    if ((uint64_t)r < (uint64_t)nrow) atomicAdd(hist + r, 1);
  }
}

// ... and for an ascending row array the atomic-free form: the thread that sees the key jump at edge e
// writes rowptr[prev+1 .. cur] = e.
__global__ __launch_bounds__(kThreads) void coo_sorted_to_csr_kernel(const int64_t *__restrict__ row, int64_t nnz,
                                                                      int64_t nrow, int *__restrict__ rowptr) {
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e <= nnz; e += (int64_t)gridDim.x * kThreads) {
    const int64_t prev = e > 0 ? row[e - 1] : -1;
    int64_t cur = e < nnz ? row[e] : nrow;
    if (cur > nrow) cur = nrow; // (a descending step writes nothing: garbage in, memory-safe out)
    for (int64_t r = (prev < -1 ? -1 : prev) + 1; r <= cur; ++r) rowptr[r] = (int)e;
  }
}

// ---- measurement hooks (geot_profile_box): what THIS box can stream right now -------------------------------
// pure non-temporal 16-B-per-lane grid-stride read (the read ceiling the tile kernel's traffic is priced
// against), and the shader clock seen by a busy wave (s_memtime ticks per s_memrealtime tick x 100 MHz)
__global__ __launch_bounds__(kThreads) void box_read_kernel(const float *__restrict__ a, float *sink, size_t n4) {
  typedef float f4_t __attribute__((ext_vector_type(4)));
  const f4_t *p = reinterpret_cast<const f4_t *>(a);
  f4_t s = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * kThreads;
  size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  for (; i + 7 * stride < n4; i += 8 * stride) { // 8 loads in flight per lane
    f4_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; i < n4; i += stride) s += __builtin_nontemporal_load(p + i);
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) sink[0] = s[0];
}

// Uniform-random rows of `piece16 * 16` bytes out of the caller's table, 16 row reads in flight per lane, nothing else: the
// yardstick of the per-edge gather kernels (configs[4]: a 57 GB table, every row a miss) - the streamed-read ceiling above is
// the wrong one for them.  A group of `piece16` lanes reads one row per step; the row number is a hash of (group, step).
template <bool NT, bool MIX>
__global__ __launch_bounds__(kThreads) void box_rows_kernel(const float *__restrict__ a, float *sink, unsigned long long rows, int piece16,
                                                             int steps, unsigned seed, int run, float *__restrict__ out, unsigned long long out_rows) {
  typedef float f4_t __attribute__((ext_vector_type(4)));
  const f4_t *p = reinterpret_cast<const f4_t *>(a);
  f4_t *o = reinterpret_cast<f4_t *>(out);
  const unsigned tid = blockIdx.x * kThreads + threadIdx.x;
  const unsigned group = tid / (unsigned)piece16, lane = tid % (unsigned)piece16;
  f4_t s = {0, 0, 0, 0};
  unsigned long long x = ((unsigned long long)group << 32) ^ seed;
  // MIX: the sum of every `run` rows read is written out as one row - the operator's read / write mix with trivial segment handling;
  // a lane group's output rows are consecutive (as a dst-sorted edge range's are), starting at its own place in `out`
  unsigned long long orow = MIX ? ((unsigned long long)group * (unsigned long long)((steps + run - 1) / (run > 0 ? run : 1))) % out_rows : 0;
  int left = run;
  for (int i = 0; i < steps; i += 16) {
    f4_t v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      x = x * 6364136223846793005ull + 1442695040888963407ull; // (one 64-bit LCG step per row: the top 32 bits pick it)
      const unsigned long long r = ((x >> 32) * rows) >> 32;
      const f4_t *q = p + r * (unsigned long long)piece16 + lane;
      v[u] = NT ? __builtin_nontemporal_load(q) : *q;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      s += v[u];
      if (MIX && --left == 0) {
        __builtin_nontemporal_store(s, o + orow * (unsigned long long)piece16 + lane);
        orow = orow + 1 < out_rows ? orow + 1 : 0;
        s = f4_t{0, 0, 0, 0};
        left = run;
      }
    }
  }
  if (!MIX && s[0] + s[1] + s[2] + s[3] == 123.456f) sink[0] = s[0];
}

// geot_internal_fill (internal.h): 16 bytes a lane where the buffer allows, words otherwise
__global__ __launch_bounds__(kThreads) void fill_quads_kernel(uint32_t *p, size_t words, uint32_t value) {
  typedef uint32_t u4f_t __attribute__((ext_vector_type(4)));
  const size_t quads = words >> 2;
  u4f_t *q = reinterpret_cast<u4f_t *>(p);
  const u4f_t v = {value, value, value, value};
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < quads; i += (size_t)gridDim.x * kThreads) q[i] = v;
  if (blockIdx.x == 0 && threadIdx.x < (words & 3)) p[(quads << 2) + threadIdx.x] = value;
}
__global__ __launch_bounds__(kThreads) void fill_words_kernel(uint32_t *p, size_t words, uint32_t value) {
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < words; i += (size_t)gridDim.x * kThreads) p[i] = value;
}

__global__ void box_clock_kernel(unsigned long long *out, int spins) {
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  float x = (float)threadIdx.x;
  for (int i = 0; i < spins; ++i) x = x * 1.0001f + 0.5f;
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[0] = c1 - c0;
    out[1] = r1 - r0;
    out[2] = (unsigned long long)x;
  }
}
#endif // GEOT_SEG_PART == 0

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
} // namespace

// State shared by the objects made of this file: defined by part 0, declared by the others.
namespace geot_seg {
#if GEOT_SEG_PART
#define GEOT_SHARED(type, name, ...) extern type name;
#else
#define GEOT_SHARED(type, name, ...) type name __VA_ARGS__;
#endif
GEOT_SHARED(thread_local std::string, g_err)

// Experiment knobs (geot_tune / geot_set_option): process-wide, relaxed atomics - a call reads each of them once or
// twice while it plans; flipping one while calls are in flight on other threads changes which (equally correct)
// kernel shape those calls pick, nothing else.
struct Tune {
  std::atomic<int> cg{0}, vec{0}, nt{-1}, lpr_log2{-1};
};
GEOT_SHARED(Tune, g_tune)
GEOT_SHARED(std::atomic<int>, g_unroll, {0})       // 0 = rule, 8 / 16 = forced
GEOT_SHARED(std::atomic<int>, g_lds_floor, {-1})   // tile kernel, 16-bit storage: dynamic LDS asked for per workgroup at least (caps the workgroups per CU); -1 = by the rule
GEOT_SHARED(std::atomic<int>, g_sddmm_shift, {-1}) // sddmm_coo_kernel: lanes per row = natural >> shift (each lane then walks 2^shift 16-byte pieces); -1 = by the rule
GEOT_SHARED(std::atomic<int>, g_gather_grid, {4096}) // make_plan: tiles a gathered call is cut into at least, where its size allows (0 = no such bound)
GEOT_SHARED(std::atomic<int>, g_xcd, {1})          // XCD-aware tile mapping for the gather modes
GEOT_SHARED(std::atomic<int>, g_nt_keys, {0})      // nt key loads: measured neutral (within the +-4 % process-to-process noise), off
GEOT_SHARED(std::atomic<int>, g_hub, {-1})         // window sums for hub chains: -1 = by the nnz / K rule, 0 = never, 1 = whenever there are > 64 tiles
GEOT_SHARED(std::atomic<int>, g_handoff_tries, {20000}) // "handoff_tries": polls before a tile leaves a run to the second launch (0: tests of that path).  ~0.5 us a
                                          // poll: ~10 ms - the predecessor is an EARLIER workgroup (dispatched in order: running or done), its whole life is
                                          // microseconds; the 400 000 of round 3 let a stalled workgroup spin for 0.2 s
GEOT_SHARED(std::atomic<int>, g_handoff, {1})      // in-kernel hand-off of the tile carries ("handoff" option): 1 = where the kernels support it, 0 = classic second pass
GEOT_SHARED(std::atomic<int>, g_ragged, {1})       // "ragged": 1 = rows that are not whole 16-byte vectors keep full-width lanes (seg_tile_rag_kernel), 0 = the 8- / 4-byte-per-lane kernels
GEOT_SHARED(std::atomic<int>, g_narrow, {1})       // fp32 rows of <= kNarrowMaxF values: 1 = lane-sequential kernel, 2 = lane-per-edge scan kernel, 0 = lane groups
GEOT_SHARED(std::atomic<int>, g_lane_e, {0})       // experiment: 4 | 8 forces E of seg_lane_kernel where that instantiation exists (F <= 4)
// Tag of a call's hand-off flags: one counter for every storage type (calls of all types share a stream's workspace).  The second
// launch lowers the flags after every call (seg_fixup_kernel), so between calls they stand at zero; the tag keeps a call apart from
// whatever an interrupted call may have left behind
GEOT_SHARED(std::atomic<unsigned long long>, g_epoch, {0})

struct Prof {
  bool on = false;
  std::mutex mu;
  struct Rec { hipEvent_t e0, e1, e2, e3; bool has_fix; };
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;
  double main_ms = 0, fix_ms = 0, aux_ms = 0;
  int64_t calls = 0;
};
GEOT_SHARED(Prof, g_prof)

// name of the dominant kernel of the calling thread's last call, as rocprofv3 prints it (geot_last_kernel)
GEOT_SHARED(thread_local std::string, t_last_kernel)

// one-shot request of the calling thread (geot_publish_word): consumed by the next segment op it launches
struct PublishRequest {
  bool armed = false;
  const int64_t *src = nullptr;
  int64_t *dst = nullptr;
  int64_t seq = 0;
};
GEOT_SHARED(thread_local PublishRequest, t_pub)
GEOT_SHARED(thread_local int64_t *, t_alarm, = nullptr) // geot_set_alarm_word: sticky, per calling thread
#undef GEOT_SHARED
} // namespace geot_seg

namespace {
using namespace geot_seg;

hipEvent_t prof_event() {
  if (!g_prof.pool.empty()) {
    hipEvent_t e = g_prof.pool.back();
    g_prof.pool.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

template <typename T> constexpr const char *type_name() {
  if constexpr (std::is_same<T, float>::value) return "float";
  else if constexpr (std::is_same<T, double>::value) return "double";
  else if constexpr (std::is_same<T, half_t>::value) return "_Float16";
  else return "__bf16";
}
template <typename T, int VEC, bool GATHER, int WMODE, bool ATOMIC, int NT, int RED, int U> void note_tile_kernel() {
  char buf[160];
  std::snprintf(buf, sizeof buf, "seg_tile_kernel<%s, %d, %s, %d, %s, %d, %d, %d>", type_name<T>(), VEC, GATHER ? "true" : "false", WMODE,
                ATOMIC ? "true" : "false", NT, RED, U);
  t_last_kernel = buf;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return fail(GEOT_ELAUNCH, std::string(#expr) + ": " + hipGetErrorString(_e));            \
  } while (0)

struct Plan {
  int vec, lpr_log2, cg, te, unroll;
  bool ragged; // rows are not whole vectors (or not on 16-byte boundaries): seg_tile_rag_kernel
  int64_t num_tiles, nfb;
  size_t meta_off, cnt_off, carry_off, list_off, wsum_off, wcnt_off, wflag_off, flag_off, total; // ctrl block sits at offset 0
  int64_t gap_cap;
};

constexpr size_t kCtrlBytes = 256;

inline int ceil_log2(int64_t x) {
  int l = 0;
  while (((int64_t)1 << l) < x) ++l;
  return l;
}

inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// workspace layout for P.num_tiles tiles: ctrl | meta | counts | carries | gap list | window sums
inline void layout_workspace(Plan &P, int64_t F, int asize) {
  const size_t nt = (size_t)(P.num_tiles > 0 ? P.num_tiles : 1);
  const size_t nw = nt / 64 + 2;
  P.meta_off = kCtrlBytes;
  P.cnt_off = P.meta_off + up256(nt * sizeof(int64_t));
  P.carry_off = P.cnt_off + up256(nt * 2 * sizeof(int64_t));
  P.list_off = P.carry_off + up256(nt * 2 * (size_t)F * asize);
  P.wsum_off = P.list_off + up256((size_t)P.gap_cap * 16);
  P.wcnt_off = P.wsum_off + up256(nw * (size_t)F * asize);
  P.wflag_off = P.wcnt_off + up256(nw * sizeof(int64_t));
  P.flag_off = P.wflag_off + up256(nw * sizeof(int));
  P.total = P.flag_off + up256(nt * sizeof(unsigned long long)); // hand-off flags, one per tile
}

inline int lane_seq_edges(int64_t F) { return F <= 4 ? (g_lane_e == 4 ? 4 : 8) : 4; }  // E of seg_lane_kernel<F, E>
constexpr int scan_steps(int64_t F) { return F <= 2 ? 8 : 4; }      // S of seg_narrow_kernel<F, S>

// vec_unit: the feature granule that must stay inside one vector (F, or F per head for mh_spmm)
// hw: weights staged in LDS per edge (0, 1 or H); gather: src offsets staged in LDS
Plan make_plan(int64_t nnz, int64_t F, int64_t vec_unit, int64_t K, int tsize, bool aligned16,
               bool gather, int hw, bool atomic_flush = false, int asize = 0, bool aligned4 = false) {
  if (asize == 0) asize = tsize < 4 ? 4 : tsize; // accumulator size: fp32 for the 16-bit storage types
  Plan P;
  const int maxvec = 16 / tsize; // 16 B per lane
  int vec = 1;
  // atomic flushes want one element per lane: a lane group then adds LPR*4 contiguous bytes per
  // instruction (256 B at F>=64), the only shape that reaches the chip's float-atomic rate
  // (measured 2x over the 16-B-per-lane layout, whose atomics stride by 16 B)
  P.ragged = false;
  if (!atomic_flush) {
    if (aligned16 && vec_unit % maxvec == 0) vec = maxvec;
    else if (g_ragged && vec_unit == F && F >= maxvec && (F * tsize) % 4 == 0 && aligned4) {
      // rows that are not whole 16-byte vectors, or operands off the 16-byte grid: full-width lanes all the same (seg_tile_body, RAG)
      vec = maxvec;
      P.ragged = true;
    }   // (anything else - a head width that is not a whole vector, operands off the 4-byte grid - takes one element per lane; the
        //  8-byte-per-lane instantiations of rounds 1-3 are gone: the ragged lanes cover their cases at twice the rate)
  }
  if (g_tune.vec > 0 && g_tune.vec <= vec && vec_unit % g_tune.vec == 0) {
    if (g_tune.vec != vec) P.ragged = false;
    vec = g_tune.vec;
  }
  P.vec = vec;
  const int64_t lanes = (F + vec - 1) / vec;
  int l = ceil_log2(lanes);
  if (l > 6) l = 6;
  if (l < kMinLprLog2) l = kMinLprLog2;
  const int natural = l;
  // Shape-keyed rule (the role of the reference's generated decision tree, wrapper/*_rule.h, re-measured
  // on MI355X with `tools/kbench sweep` and `tools/sweep_rule.py`, profiles/r01/sweep_rule*.{csv,txt}):
  //  * rows of <= 4 lanes (F <= 16 fp32) get 8-lane groups although half the lanes then idle: 32 groups of
  //    32 edges beat 64 groups of 16 wherever runs are long (-10..-35 %: half the LDS partials to merge)
  //    and are neutral for gathers on short runs; streamed rows with short runs (nnz/K < 24) keep the
  //    natural 4-lane groups (5-9 % better there); the atomic flush keeps one lane per element;
  //  * streamed rows (index_scatter) like ~512-edge tiles, 512-B rows 256-edge tiles; gathered rows ~1024;
  //  * 16 row loads in flight per lane pay off only at 16 lanes per row (F in (32, 64], fp32).
  //  * launch-bound sizes (<= 400 k edges) take the 8-lane groups whatever the run length: twice the tiles in flight
  //    (index_scatter F=16 on 100-150 k edges: 18 -> 14 us per call).
  if (!atomic_flush && l < 3 && (gather || nnz >= 24 * (K > 0 ? K : 1) || nnz <= 400000)) l = 3;
  //  * launch-bound sizes with LONG runs (>= 128 edges per row: pooling over small graphs, hub-dominated batches): one more
  //    doubling of the lane group, up to 32 lanes - half the partials in the tile's LDS merge, which is what such a call
  //    waits for (round-3 sweep, profiles/r03/sweep_rule.csv: 100-300 k edges, avg 500, F = 16..64: 17-22 -> 13-17 us).
  const bool small_long = !atomic_flush && nnz <= 400000 && nnz >= 128 * (K > 0 ? K : 1);
  if (small_long && l < 5 && l <= natural) l = natural + 1 < 3 ? 3 : natural + 1;
  if (g_tune.lpr_log2 >= natural && g_tune.lpr_log2 <= 6) l = g_tune.lpr_log2;
  P.lpr_log2 = l;
  const int ng = kThreads >> l;
  int cg = (gather ? 1024 : 512) / ng;
  if (l > natural) cg = 32;
  // Gathered rows: ~1024-edge tiles were chosen on uniform-random sources, where the row gathers miss every cache and the tile
  // shape hardly matters.  On graphs with LOCALITY (sources near the destination: the rows a tile gathers are in the XCD's L2 if
  // its neighbours in flight have just used them) and for multi-head weights it matters a lot (late round-3 sweep, 20 M / 2 M
  // edges, sources within +-2000 rows: gws F=256 2.32 -> 1.73 ms, mh H=8 F=32 2.42 -> 1.81 ms, mh H=4 F=16 0.78 -> 0.57 ms; the
  // same shapes on uniform-random sources: -3 .. +15 %).  Three bounds on the tile, each only ever shrinking it, 32-edge groups at least:
  //  * the dst rows the ~160 tiles in flight per XCD cover, times the row size, stay within the XCD's 4 MiB L2 (the part of
  //    the gather footprint the tile shape controls; the graph's own locality window comes on top);
  //  * the per-edge LDS (keys, offsets, staged weights) stays within 20 KB: 5 workgroups per CU for every head count;
  //  * fp32 rows: at least ~4096 tiles where the edge count allows - three rounds of tiles over the chip's ~1280 resident
  //    workgroups (0.6-2 M edges with locality: +4..13 %; uniform-random sources: level).
  if (gather && !atomic_flush && l == natural) {
    const int64_t k = K > 0 ? K : 1;
    const int64_t by_l2 = (int64_t)(((double)nnz / (double)k) * (double)((int64_t)4 << 20) / ((double)F * tsize * 160.0 * ng));
    // (runs of hundreds of edges - configs[3], Reddit scale, on the per-edge kernels - keep the large tile: 16.6 vs 17.4 ms there)
    // where that bound is mild anyway (>= 128-edge groups); 16 lanes per row with 4 heads of weights keep 32-edge groups: 0.62 -> 0.46 ms)
    const int64_t lds_cg = 20480 / (16 + 4 * (hw > 0 ? hw : 0)) / ng;
    const int64_t by_lds = nnz >= 256 * k && lds_cg >= 128 ? cg : lds_cg;
    const int gg = g_gather_grid;
    // (16-bit storage: measured 7-15 % SLOWER with it at 1-2 M edges; runs of hundreds of edges: 12-17 % slower at 0.3-1 M edges - fewer,
    //  longer groups mean fewer partials to merge, which is what such a call waits for)
    const int64_t by_grid = gg > 0 && tsize == 4 && nnz < 128 * k ? nnz / ((int64_t)ng * gg) : cg;
    int64_t c = cg;
    if (by_l2 < c) c = by_l2;
    if (by_lds < c) c = by_lds;
    if (by_grid < c) c = by_grid;
    c = c / 16 * 16;
    if (c < 32) c = 32;
    if (c < cg) cg = (int)c;
  }
  if (!gather && vec == maxvec && l == 5) cg = 32; // 512-byte streamed rows, every dtype (bf16 F=256: +7..12 %, fp64 F=64: +3..6 % over 64-edge groups)
  // 1-KiB streamed rows (fp32 F = 256: a whole wave per row): 16 loads in flight over 32-edge groups (round-3 sweep:
  // -2..-4 % at 10 M edges, -10..-15 % at 0.3-1 M edges against 128-edge groups with 8 loads in flight)
  const bool wide_u16 = !gather && !atomic_flush && tsize == 4 && vec == 4 && l == 6;
  const int64_t keys1 = K > 0 ? K : 1;
  if (wide_u16) cg = 32; // (64-edge groups above 2 M edges, the first form of this rule, re-measured 5-6 % slower at 10 M edges)
  // ... the other dtypes' 1-KiB rows (8 loads in flight): 32-edge groups, 64 where runs are long (bf16 F=512 at 2-10 M edges:
  // +11 % / +6 % at an average run of 10, +2..4 % at 50; fp64 F=128: +2 %) - the generic 128 only suited the fp32 kernel of round 1
  else if (!gather && !atomic_flush && vec == maxvec && l == 6 && lanes <= 64) cg = nnz < 30 * keys1 ? 32 : 64;
  // SHORT runs on streamed fp32 rows of >= 256 B (average run of a few edges: molecules, road networks, meshes): the smallest
  // groups, 16 edges.  tools/_ab sweep of the round, normal-distributed run lengths, 10 M / 2 M edges, auto vs 16-edge groups:
  // average 1.5-2: F=64 +6..7 % / +3.5 %, F=128 +9 % / +7 %, F=256 +10 % / +6.5 %; average 4: +2 % / -1 %, +4.4 % / +3 %,
  // +11 % / +2 %; average 6: 0 / -3 %, +2 % / +1 %, +6 % / 0; from ~10 on the larger groups win again.
  // The other dtypes (bf16 / f16 / fp64, 8 loads in flight): the same at 512-byte and 1-KiB rows (bf16 +8..16 %, fp64 +1..8 %);
  // their 256-byte rows keep 32-edge groups (16-edge groups measured 15 % SLOWER there without the 16-load kernel).
  if (!gather && !atomic_flush && vec == maxvec && l >= 4 && l == natural && lanes <= 64) {
    const int64_t k = keys1;
    if ((l == 4 && tsize == 4 && 2 * nnz < 7 * k) || (l == 5 && nnz < 8 * k) || (l == 6 && nnz < 5 * k)) cg = 16;
  }
  // launch-bound sizes: a lane group walks its cg edges in dependent batches of U row loads, so on a chip that the grid
  // does not fill (< ~400 tiles) shorter groups = more tiles finish sooner.  Measured on graphs of 15 k - 250 k edges
  // (graph replay, us per call): gws F=64 19.6 -> 12.4 / 21.6 -> 14.2, F=128 26.9 -> 11.6 / 29.0 -> 19.6, index_scatter
  // F=64 18.4 -> 15.6; from ~400 k edges on the rule above is untouched.
  if (!atomic_flush && nnz > 0) {
    int64_t cap = nnz / ((int64_t)ng * 400) / 16 * 16;
    if (cap < 16) cap = 16;
    if (small_long && l <= 4 && cap < 32) cap = 32; // (long runs: 32-edge groups beat 16-edge ones even when that halves the tiles)
    if (cap < cg) cg = (int)cap;
  }
  if (g_tune.cg > 0) cg = g_tune.cg;
  if (cg < 16) cg = 16;
  cg = (cg + 15) / 16 * 16;
  if (cg > 256) cg = 256;
  // bound the tile: <= 2048 edges (4096 when the rows are so narrow that 256 groups x 16 edges is
  // the smallest legal tile), <= ~64 KB of LDS, and 32-bit byte offsets inside a tile
  while (cg > 16 && ((int64_t)ng * cg > 2048 ||
                     smem_layout(l, cg, vec, asize, gather, hw).bytes > 64 * 1024 ||
                     (int64_t)ng * cg * F * tsize >= ((int64_t)1 << 31)))
    cg -= 16;
  P.cg = cg;
  P.te = ng * cg;
  P.unroll = (!gather && !atomic_flush && tsize == 4 && vec == 4 && (l == 4 || wide_u16) && cg % 16 == 0) ? 16 : 8;
  if (g_unroll == 8 || (g_unroll == 16 && cg % 16 == 0 && tsize == 4 && vec == 4)) P.unroll = g_unroll;
  P.num_tiles = nnz > 0 ? (nnz + P.te - 1) / P.te : 0;
  const int64_t fb = ((int64_t)1 << l) * vec;
  P.nfb = (F + fb - 1) / fb;
  if (P.nfb < 1) P.nfb = 1;
  P.gap_cap = K / kGapInline + 2;
  layout_workspace(P, F, asize);
  return P;
}

// plan of the two narrow-row kernels (fp32, F <= kNarrowMaxF): tiles of 256 lanes x E edges (or S steps);
// 4 lanes per row in the fix-up kernel
Plan narrow_plan(int64_t nnz, int64_t F, int64_t K, bool lane_seq, bool gather = false) {
  Plan P = make_plan(nnz, F, F, K, (int)sizeof(float), false, false, 0);
  P.te = kThreads * (lane_seq ? (gather ? (F <= 4 ? 8 : 4) : lane_seq_edges(F)) : scan_steps(F));
  P.num_tiles = nnz > 0 ? (nnz + P.te - 1) / P.te : 0;
  P.nfb = 1;
  P.lpr_log2 = 2;
  layout_workspace(P, F, (int)sizeof(float));
  return P;
}


// Dynamic LDS a tile-kernel launch asks for: the layout's bytes, or MORE - the only handle on the number of workgroups a CU
// takes (160 KB of LDS per CU: 33 000 bytes -> 4 workgroups, 41 000 -> 3).  Streamed rows of >= 512 bytes run faster with
// fewer tiles in flight (see g_lds_floor and make_plan's notes); everything else takes what fits.
// Measured (tools/_ab sweep, 10 M / 2 M power-law edges, sum / mean / max; gains over "what fits"): fp32 rows of 512 B: 4 per CU
// +1..2 %, 1 KiB: 3 per CU +3..6 %; bf16 512 B: 4 (+3.4 %), 1 KiB: 3 (+5 %), fp64 256 B: 4 (+1..4 %), 512 B: 3 (+4..6 %), 1 KiB: 2 (+7..10 %).  Narrower rows: every cap measured slower.
// Fewer, longer streams reach the memory at once.  Only for grids that fill the chip several times over.
// The GRADED shape too (fp32, 256-byte rows, sum, 16 loads in flight per lane, 19 532 tiles): 4 workgroups per CU instead of the
// 5 that fit: 0.4690 vs 0.4744 ms (six alternating repetitions on one box, spread 0.001); 3 per CU: 0.4838.  Its max / mean
// (8 loads in flight) lose 1-2 % with the cap and keep what fits; so do 2 M edges (3906 tiles: -3 %).
inline size_t tile_lds(const SmemLayout &L, const SegParams &p, int64_t num_tiles, int tsize, bool gather, int red, int unroll = 8) {
  int floor_bytes = g_lds_floor;
  if (floor_bytes < 0) {
    floor_bytes = 0;
    if (!gather && num_tiles >= 4096) {
      constexpr int kPerCu[4] = {0, 33000, 41000, 54000}; // what fits / 4 / 3 / 2 workgroups per CU
      int step = p.rowbytes >= 1024 ? (red == RED_SUM ? 2 : 1) : (p.rowbytes >= 512 ? 1 : 0);
      if (tsize == 8) step = p.rowbytes >= 1024 ? 3 : (p.rowbytes >= 512 ? 2 : (p.rowbytes >= 256 ? 1 : 0));
      if (tsize == 4 && p.rowbytes == 256 && unroll == 16 && red == RED_SUM && num_tiles >= 8192) step = 1;
      floor_bytes = kPerCu[step];
    }
  }
  return (size_t)floor_bytes > L.bytes ? (size_t)floor_bytes : L.bytes;
}

template <typename T, int VEC, bool GATHER, int WMODE, bool ATOMIC, int NT, int RED = RED_SUM>
void launch_tile(const SegParams &p, const Plan &P, hipStream_t st) {
  const int hw = WMODE == 0 ? 0 : (WMODE == 1 ? 1 : (int)p.H);
  const SmemLayout L = smem_layout(P.lpr_log2, P.cg, VEC, (int)sizeof(typename AccOf<T>::type), GATHER, hw);
  dim3 grid((unsigned)P.num_tiles, (unsigned)P.nfb, 1);
  if constexpr (VEC == 16 / (int)sizeof(T) && WMODE <= 1) {
    if (P.ragged) { // rows that are not whole vectors: full-width lanes, the row's last lane overlaps its neighbour (seg_tile_body, RAG)
      note_tile_kernel<T, VEC, GATHER, WMODE, ATOMIC, NT, RED, kU>();
      t_last_kernel.replace(0, 15, "seg_tile_rag_kernel");
      t_last_kernel.erase(t_last_kernel.rfind(','));   // (the ragged kernel has no U parameter)
      t_last_kernel += ">";
      hipLaunchKernelGGL((seg_tile_rag_kernel<T, VEC, GATHER, WMODE, ATOMIC, NT, RED>), grid, dim3(kThreads), tile_lds(L, p, P.num_tiles, (int)sizeof(T), GATHER, RED), st, p);
      return;
    }
  }
  note_tile_kernel<T, VEC, GATHER, WMODE, ATOMIC, NT, RED, kU>();
  hipLaunchKernelGGL((seg_tile_kernel<T, VEC, GATHER, WMODE, ATOMIC, NT, RED>), grid, dim3(kThreads), tile_lds(L, p, P.num_tiles, (int)sizeof(T), GATHER, RED), st, p);
}

// non-sum reductions, sorted index.  index_scatter: streamed operand -> nt loads + stores;
// gather_scatter / gather_weight_scatter (PyG-style aggr= mean / max / ... over the messages): default policy
template <typename T, int RED, bool GATHER, int WMODE>
int dispatch_reduce_mode(const SegParams &p, const Plan &P, hipStream_t st) {
  constexpr int MAXV = 16 / (int)sizeof(T);
  constexpr int NTP = GATHER ? 0 : 3;
  if constexpr (!GATHER && WMODE == 0 && sizeof(T) == 4 && (RED == RED_MAX || RED == RED_MIN)) {
    if (P.unroll == 16 && P.vec == MAXV && !P.ragged) {   // 16 row loads in flight per lane, as the sum (the tile shape was chosen for it)
      const SmemLayout L = smem_layout(P.lpr_log2, P.cg, MAXV, (int)sizeof(T), false, 0);
      dim3 grid((unsigned)P.num_tiles, (unsigned)P.nfb, 1);
      note_tile_kernel<T, MAXV, false, 0, false, 3, RED, 16>();
      hipLaunchKernelGGL((seg_tile_kernel<T, MAXV, false, 0, false, 3, RED, 16>), grid, dim3(kThreads), tile_lds(L, p, P.num_tiles, (int)sizeof(T), false, RED, 16), st, p);
      return GEOT_OK;
    }
  }
  if (P.vec == MAXV) launch_tile<T, MAXV, GATHER, WMODE, false, NTP, RED>(p, P, st);
  else if (P.vec == 1) launch_tile<T, 1, GATHER, WMODE, false, NTP, RED>(p, P, st);
  else return fail(GEOT_EINVAL, "internal: bad vector width");
  return GEOT_OK;
}

template <typename T, int RED>
int dispatch_reduce(const SegParams &p, const Plan &P, hipStream_t st, int mode) {
  if (mode == 0) return dispatch_reduce_mode<T, RED, false, 0>(p, P, st);
  if (mode == 1) return dispatch_reduce_mode<T, RED, true, 0>(p, P, st);
  return dispatch_reduce_mode<T, RED, true, 1>(p, P, st);
}

template <typename T>
void launch_fixup(const SegParams &p, int64_t blocks, int64_t num_tiles, int red, hipStream_t st) {
  switch (red) {
  case RED_MAX: hipLaunchKernelGGL((seg_fixup_kernel<T, RED_MAX>), dim3((unsigned)blocks), dim3(kThreads), 0, st, p, num_tiles); break;
  case RED_MEAN: hipLaunchKernelGGL((seg_fixup_kernel<T, RED_MEAN>), dim3((unsigned)blocks), dim3(kThreads), 0, st, p, num_tiles); break;
  case RED_MIN: hipLaunchKernelGGL((seg_fixup_kernel<T, RED_MIN>), dim3((unsigned)blocks), dim3(kThreads), 0, st, p, num_tiles); break;
  case RED_PROD: hipLaunchKernelGGL((seg_fixup_kernel<T, RED_PROD>), dim3((unsigned)blocks), dim3(kThreads), 0, st, p, num_tiles); break;
  default: hipLaunchKernelGGL((seg_fixup_kernel<T, RED_SUM>), dim3((unsigned)blocks), dim3(kThreads), 0, st, p, num_tiles); break;
  }
}

template <typename T>
void launch_wsum(const SegParams &p, int64_t num_tiles, int red, hipStream_t st) {
  const dim3 grid((unsigned)((num_tiles + 63) / 64)), blk(kThreads);
  switch (red) {
  case RED_MAX: hipLaunchKernelGGL((seg_wsum_kernel<T, RED_MAX>), grid, blk, 0, st, p, num_tiles); break;
  case RED_MEAN: hipLaunchKernelGGL((seg_wsum_kernel<T, RED_MEAN>), grid, blk, 0, st, p, num_tiles); break;
  case RED_MIN: hipLaunchKernelGGL((seg_wsum_kernel<T, RED_MIN>), grid, blk, 0, st, p, num_tiles); break;
  case RED_PROD: hipLaunchKernelGGL((seg_wsum_kernel<T, RED_PROD>), grid, blk, 0, st, p, num_tiles); break;
  default: hipLaunchKernelGGL((seg_wsum_kernel<T, RED_SUM>), grid, blk, 0, st, p, num_tiles); break;
  }
}

// Cache policy of the row loads / dst stores.  Only what the rule can select is instantiated (round 4; the nt = 1 / 2 halves and
// nt gathers were round-1 experiments: measured, recorded in CHANGELOG, never selected since): streamed rows nt loads + nt
// stores (3) or the default policy (0); gathered rows - re-used across edges - the default policy only.
template <typename T, int VEC, bool GATHER, int WMODE, bool ATOMIC>
void dispatch_nt(const SegParams &p, const Plan &P, hipStream_t st, int nt) {
  if constexpr (GATHER) {
    // (round 5: nt row gathers for a table whose rows are read from HBM every time - configs[4]'s 57 GB: +2.5 %; fp32 gather_scatter
    //  only; run_segment_op selects it for tables beyond 4 GiB, geot_tune(nontemporal = 0 | 1) forces either)
    if constexpr (WMODE == 0 && !ATOMIC && sizeof(T) == 4 && VEC == 4) {
      if ((nt & 1) != 0) {
        launch_tile<T, VEC, GATHER, WMODE, ATOMIC, 1>(p, P, st);
        return;
      }
    }
    launch_tile<T, VEC, GATHER, WMODE, ATOMIC, 0>(p, P, st);
  }
  else if ((nt & 3) != 0) launch_tile<T, VEC, GATHER, WMODE, ATOMIC, 3>(p, P, st);
  else launch_tile<T, VEC, GATHER, WMODE, ATOMIC, 0>(p, P, st);
}

template <typename T, bool GATHER, int WMODE, bool ATOMIC>
int dispatch_vec(const SegParams &p, const Plan &P, hipStream_t st, int nt) {
  constexpr int MAXV = 16 / (int)sizeof(T);
  if constexpr (!GATHER && WMODE == 0 && !ATOMIC && sizeof(T) == 4) {
    if (P.unroll == 16 && P.vec == MAXV && !P.ragged && (nt & 3) == 3) {   // 16 row loads in flight per lane
      const SmemLayout L = smem_layout(P.lpr_log2, P.cg, MAXV, (int)sizeof(T), false, 0);
      dim3 grid((unsigned)P.num_tiles, (unsigned)P.nfb, 1);
      note_tile_kernel<T, MAXV, false, 0, false, 3, RED_SUM, 16>();
      hipLaunchKernelGGL((seg_tile_kernel<T, MAXV, false, 0, false, 3, RED_SUM, 16>), grid, dim3(kThreads), tile_lds(L, p, P.num_tiles, (int)sizeof(T), false, RED_SUM, 16), st, p);
      return GEOT_OK;
    }
  }
  if constexpr (GATHER && WMODE <= 1 && !ATOMIC && sizeof(T) == 4) {
    if (P.unroll == 16 && P.vec == MAXV && !P.ragged && (nt & 3) == 0) {   // gather modes, default cache policy
      const SmemLayout L = smem_layout(P.lpr_log2, P.cg, MAXV, (int)sizeof(T), true, WMODE);
      dim3 grid((unsigned)P.num_tiles, (unsigned)P.nfb, 1);
      note_tile_kernel<T, MAXV, true, WMODE, false, 0, RED_SUM, 16>();
      hipLaunchKernelGGL((seg_tile_kernel<T, MAXV, true, WMODE, false, 0, RED_SUM, 16>), grid, dim3(kThreads), tile_lds(L, p, P.num_tiles, (int)sizeof(T), true, RED_SUM), st, p);
      return GEOT_OK;
    }
  }
  if (P.vec == MAXV) dispatch_nt<T, MAXV, GATHER, WMODE, ATOMIC>(p, P, st, nt);
  else if (P.vec == 1) dispatch_nt<T, 1, GATHER, WMODE, ATOMIC>(p, P, st, nt);
  else return fail(GEOT_EINVAL, "internal: bad vector width");
  return GEOT_OK;
}

inline bool is_aligned16(const void *a) { return ((uintptr_t)a & 15) == 0; }

// mode: 0 index_scatter, 1 gather_scatter, 2 gather_weight_scatter, 3 mh edge-major, 4 mh head-major
template <typename T>
int run_segment_op(int mode, bool sorted, const int64_t *src_index, const int64_t *dst_index,
                   const void *weight, const void *src, void *dst, int64_t nnz, int64_t F,
                   int64_t H, int64_t src_rows, int64_t K, void *ws, size_t ws_bytes,
                   hipStream_t st, int red = RED_SUM) {
  if (red != RED_SUM && (mode > 2 || !sorted))
    return fail(GEOT_EUNSUPPORTED, "non-sum reductions: sorted index_scatter / gather_scatter / gather_weight_scatter only");
  if (nnz < 0 || F < 0 || K < 0 || src_rows < 0 || H < 1) return fail(GEOT_EINVAL, "negative size");
  if (K == 0 || F == 0) return GEOT_OK;
  if (!dst || (nnz > 0 && (!dst_index || !src))) return fail(GEOT_EINVAL, "null pointer");
  if (mode >= 1 && nnz > 0 && !src_index) return fail(GEOT_EINVAL, "null src_index");
  if (mode >= 2 && nnz > 0 && !weight) return fail(GEOT_EINVAL, "null weight");
  const int64_t Fh = F / H;
  if (F * (int64_t)sizeof(T) >= ((int64_t)1 << 31)) return fail(GEOT_EUNSUPPORTED, "row too wide");
  const bool al = is_aligned16(src) && is_aligned16(dst) && is_aligned16(ws);
  const int hw = mode <= 1 ? 0 : (mode == 2 ? 1 : (int)H);
  if (hw > 64) return fail(GEOT_EUNSUPPORTED, "more than 64 heads");
  const bool al4 = mode <= 2 && (((uintptr_t)src | (uintptr_t)dst) & 3) == 0; // (multi-head weights: a vector must stay inside one head - no ragged lanes)
  Plan P = make_plan(nnz, F, mode >= 3 ? Fh : F, K, (int)sizeof(T), al, mode >= 1, hw, !sorted, 0, al4);
  // narrow fp32 rows: lane-sequential kernel (sum / max / min / prod; needs a 16-B aligned src), or the
  // lane-per-edge scan kernel (sum only; option "narrow" = 2, and the fallback for an unaligned src)
  const bool narrow_ok = std::is_same<T, float>::value && mode <= 2 && sorted && F <= kNarrowMaxF && g_narrow &&
                         is_aligned16(dst); // both kernels store F = 2 / 4 / 8 rows as vectors
  // (the lane-sequential kernel reads its per-lane index / weight chunks and its rows as 16-B vectors)
  const bool lane_seq = narrow_ok && red != RED_MEAN && g_narrow == 1 && is_aligned16(src) && is_aligned16(dst_index) &&
                        (mode == 0 || is_aligned16(src_index)) && (mode != 2 || is_aligned16(weight)) &&
                        (mode == 0 || red != RED_PROD);
  const bool narrow_path = lane_seq || (narrow_ok && mode == 0 && red == RED_SUM);
  if (narrow_path) P = narrow_plan(nnz, F, K, lane_seq, mode != 0);
  if (!ws || ws_bytes < P.total) return fail(GEOT_EWORKSPACE, "workspace too small");
  if (((uintptr_t)ws & 255) != 0) return fail(GEOT_EWORKSPACE, "workspace must be 256-byte aligned");

  char *wsc = static_cast<char *>(ws);
  SegParams p;
  p.dst_index = dst_index;
  p.src_index = src_index;
  p.weight = weight;
  p.src = src;
  p.dst = dst;
  p.ctrl = reinterpret_cast<unsigned long long *>(wsc);
  p.meta = reinterpret_cast<int64_t *>(wsc + P.meta_off);
  p.ccnt = reinterpret_cast<int64_t *>(wsc + P.cnt_off);
  p.carry = wsc + P.carry_off;
  p.gap_list = reinterpret_cast<int64_t *>(wsc + P.list_off);
  p.gap_cap = P.gap_cap;
  p.nnz = nnz;
  p.F = F;
  p.K = K;
  p.src_rows = src_rows;
  p.H = H;
  p.Fh = Fh > 0 ? Fh : 1;
  p.rowbytes = (uint32_t)(F * (int64_t)sizeof(T));
  p.lpr_log2 = P.lpr_log2;
  p.cg = P.cg;
  p.xcd_swizzle = g_xcd;
  p.nt_keys = g_nt_keys;
  p.mode = mode;
  p.alarm = t_alarm;
  // Hand-off mode: the tile kernel finishes the straddling runs itself and the second launch only tidies up (seg_tile_kernel,
  // "hand-off"; -3.5 % per call at the graded configuration).  Streamed rows with fp32 accumulators in whole 16-byte pieces,
  // one feature block, no per-run counts; the few-key regime keeps the window sums and the classic second pass.
  const bool acc_f32 = std::is_same<typename AccOf<T>::type, float>::value;
  // Measured (tools/bench_handoff_ab.py: F = 16..256 x 1-10 M edges, fp32 / bf16, alternating modes on one box): rows of >= 256 bytes gain
  // 2.5-8 % at every size; rows of <= 128 bytes gain 2-5 % up to ~1 M edges and LOSE 2-16 % beyond (short tiles: the drain at
  // the end of every workgroup is a larger share of its life) - those keep the classic second pass.  `handoff` = 2 forces.
  // ... and not where such narrow rows come in runs of hundreds of edges above launch-bound sizes (2 M edges on 2 000-8 000 keys, F = 16 / 32:
  // -7..-25 %: the walk back over a hub's tiles is not hidden by anything there).
  const bool narrow_long = p.rowbytes <= 128 && nnz > 700000 && nnz >= 128 * (K > 0 ? K : 1);
  const bool ho_pays = p.rowbytes >= 256 || (nnz <= 2000000 && !narrow_long) || g_handoff == 2;
  p.handoff = (g_handoff && ho_pays && sorted && mode == 0 && !narrow_path && acc_f32 && P.vec % 4 == 0 && !P.ragged && P.nfb == 1 && nnz > 0) ? 1 : 0;
  p.ho_tries = g_handoff_tries;
  p.flags = reinterpret_cast<unsigned long long *>(wsc + P.flag_off);
  p.epoch = 0x6E07A5C300000000ull + (++g_epoch & 0xFFFFFFFFull); // (a tag no stale word of the workspace will equal)
  // a pending geot_publish_word is consumed by the first kernel of this call if that kernel is one of the three that
  // publish (tile / lane / narrow kernel; not the LDS-bin kernel of the unsorted atomic path)
  const bool lds_bin = !sorted && (size_t)K * (size_t)F * sizeof(T) <= 48 * 1024 && g_tune.lpr_log2 != 7;
  const bool publish = t_pub.armed && nnz > 0 && !lds_bin;
  p.pub_src = publish ? t_pub.src : nullptr;
  p.pub_dst = publish ? t_pub.dst : nullptr;
  p.pub_seq = publish ? t_pub.seq : 0;
  if (publish) t_pub.armed = false;
  // Long-run regime (few keys: global pooling, hub-dominated graphs): average run >= 4096 edges and enough
  // tiles for a chain to span whole 64-tile windows -> one extra small launch (seg_wsum_kernel) between the
  // tile kernel and the fix-up.  Everything else keeps two launches.  "hub" option: 1 forces, 0 forbids.
  const bool use_wsum = sorted && P.num_tiles > 64 &&
                        (g_hub == 1 || (g_hub < 0 && P.num_tiles >= 256 && nnz / 4096 >= K));
  if (use_wsum) p.handoff = 0; // (long chains of tiles under one key: the window sums and the classic second pass are the fast way)
  p.wsum = use_wsum ? wsc + P.wsum_off : nullptr;
  p.wcnt = use_wsum ? reinterpret_cast<int64_t *>(wsc + P.wcnt_off) : nullptr;
  p.wflag = use_wsum ? reinterpret_cast<int *>(wsc + P.wflag_off) : nullptr;

  // non-temporal policy: the streamed operand of index_scatter is read exactly once -> nt loads
  // (measured +13 % with the store mix of this op) and nt dst stores (a further ~5 %);
  // gathered rows are re-used across edges -> default cache policy there.
  const int tune_nt = g_tune.nt;
  int nt = tune_nt >= 0 ? tune_nt : (mode == 0 ? 3 : 0);
  // ... except gathers from a table that no cache can hold (> 4 GiB: 16 x the Infinity Cache): every row comes from HBM whatever the
  // policy, and non-temporal gathers keep the rows from pushing the indices and the output out - configs[4]'s shard (57 GB table):
  // 4.81 -> 4.93 TB/s of row reads (profiles/r05/cfg5_study/exp_gather_table__session2_quick_with_nt_gathers.txt).  fp32 gather_scatter
  // only (the one instantiation, see dispatch_nt); smaller tables keep the default policy (re-used rows: measured in round 1).
  if (tune_nt < 0 && mode == 1 && (uint64_t)src_rows * (uint64_t)p.rowbytes > ((uint64_t)4 << 30)) nt = 1;

  Prof::Rec rec{};
  const bool prof = g_prof.on;
  if (prof) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    rec.e0 = prof_event(); rec.e1 = prof_event(); rec.e2 = prof_event(); rec.e3 = prof_event();
  }
  if (prof) HIP_TRY(hipEventRecord(rec.e0, st));
  if (nnz == 0 || !sorted) {                                   // (a kernel, not a memset node: internal.h geot_internal_fill)
    const size_t zb = (size_t)K * (size_t)F * sizeof(T);
    if ((zb & 3) == 0 && (((uintptr_t)dst) & 3) == 0) { if (geot_internal_fill(dst, zb, 0u, st) != GEOT_OK) return GEOT_ELAUNCH; }
    else HIP_TRY(hipMemsetAsync(dst, 0, zb, st));
  }
  if (prof) HIP_TRY(hipEventRecord(rec.e1, st));

  int rc = GEOT_OK;
#ifdef GEOT_HEADLINE_ONLY
  // developer build (seconds instead of minutes; tools/isa.sh): only the kernels of the graded configuration are
  // instantiated - fp32 index_scatter, 16 lanes per row, 16 loads in flight, nt loads + stores - plus its fix-up
  if constexpr (std::is_same<T, float>::value) {
    const SmemLayout L = smem_layout(P.lpr_log2, P.cg, 4, 4, false, 0);
    hipLaunchKernelGGL((seg_tile_kernel<float, 4, false, 0, false, 3, RED_SUM, 16>), dim3((unsigned)P.num_tiles, (unsigned)P.nfb, 1), dim3(kThreads), L.bytes, st, p);
    const int64_t tpb = (kThreads / 64) * (64 >> P.lpr_log2);
    const bool skip_fixup = false;
    if (prof) HIP_TRY(hipEventRecord(rec.e2, st));
    if (!skip_fixup)
      hipLaunchKernelGGL((seg_fixup_kernel<float, RED_SUM>), dim3((unsigned)((P.num_tiles + tpb - 1) / tpb)), dim3(kThreads), 0, st, p, P.num_tiles);
    if (prof) {
      HIP_TRY(hipEventRecord(rec.e3, st));
      rec.has_fix = true;
      std::lock_guard<std::mutex> lk(g_prof.mu);
      g_prof.recs.push_back(rec);
    }
  }
  (void)lane_seq; (void)narrow_path; (void)nt; (void)use_wsum;
  return GEOT_OK;
#else
  if (nnz > 0) {
    if (!sorted) {
      if (mode != 0) return fail(GEOT_EUNSUPPORTED, "unsorted is index_scatter only");
      if constexpr (sizeof(T) < 4) return fail(GEOT_EUNSUPPORTED, "unsorted index needs float32/float64 (float atomics)");
      const size_t tab_bytes = (size_t)K * (size_t)F * sizeof(T);
      if (tab_bytes <= 48 * 1024 && g_tune.lpr_log2 != 7) {   // (lpr_log2 = 7: test knob, forces the direct path)
        int l = ceil_log2(F);
        if (l > 6) l = 6;
        int64_t blocks = (nnz + 4095) / 4096;                  // >= 4096 edges per block, <= 4 blocks per CU
        if (blocks > 1024) blocks = 1024;
        if (blocks < 1) blocks = 1;
        t_last_kernel = std::string("seg_lds_bin_kernel<") + type_name<T>() + ">";
        if constexpr (sizeof(T) >= 4)
          hipLaunchKernelGGL((seg_lds_bin_kernel<T>), dim3((unsigned)blocks), dim3(kThreads), tab_bytes, st,
                             dst_index, static_cast<const T *>(src), static_cast<T *>(dst), nnz, F, K, l);
      } else if constexpr (sizeof(T) >= 4) {
        rc = dispatch_vec<T, false, 0, true>(p, P, st, nt);
      }
    } else {
      bool narrow = false;
      if constexpr (sizeof(T) == 4 && std::is_same<T, float>::value) {
        // narrow fp32 rows: lane-per-edge scan kernel (its tile size was fixed by the plan below)
        if (lane_seq) {
          const dim3 grid((unsigned)P.num_tiles), blk(kThreads);
          const bool e8 = lane_seq_edges(F) == 8;
          t_last_kernel = "seg_lane_kernel<" + std::to_string((int)F) + ", " + std::to_string(mode == 0 ? (e8 ? 8 : 4) : (F <= 4 ? 8 : 4)) + ", " +
                          std::to_string(red) + ", " + std::to_string(mode) + ">";
#define GEOT_LANE_F(RED_)                                                                                          \
  switch ((int)F) {                                                                                                \
  case 1: if (e8) hipLaunchKernelGGL((seg_lane_kernel<1, 8, RED_>), grid, blk, 0, st, p); else hipLaunchKernelGGL((seg_lane_kernel<1, 4, RED_>), grid, blk, 0, st, p); break; \
  case 2: if (e8) hipLaunchKernelGGL((seg_lane_kernel<2, 8, RED_>), grid, blk, 0, st, p); else hipLaunchKernelGGL((seg_lane_kernel<2, 4, RED_>), grid, blk, 0, st, p); break; \
  case 3: if (e8) hipLaunchKernelGGL((seg_lane_kernel<3, 8, RED_>), grid, blk, 0, st, p); else hipLaunchKernelGGL((seg_lane_kernel<3, 4, RED_>), grid, blk, 0, st, p); break; \
  case 4: if (e8) hipLaunchKernelGGL((seg_lane_kernel<4, 8, RED_>), grid, blk, 0, st, p); else hipLaunchKernelGGL((seg_lane_kernel<4, 4, RED_>), grid, blk, 0, st, p); break; \
  case 5: hipLaunchKernelGGL((seg_lane_kernel<5, 4, RED_>), grid, blk, 0, st, p); break;                           \
  case 6: hipLaunchKernelGGL((seg_lane_kernel<6, 4, RED_>), grid, blk, 0, st, p); break;                           \
  case 7: hipLaunchKernelGGL((seg_lane_kernel<7, 4, RED_>), grid, blk, 0, st, p); break;                           \
  default: hipLaunchKernelGGL((seg_lane_kernel<8, 4, RED_>), grid, blk, 0, st, p); break;                          \
  }
#define GEOT_LANE_G(RED_, MODE_)                                                                                   \
  switch ((int)F) {                                                                                                \
  case 1: hipLaunchKernelGGL((seg_lane_kernel<1, 8, RED_, MODE_>), grid, blk, 0, st, p); break;                    \
  case 2: hipLaunchKernelGGL((seg_lane_kernel<2, 8, RED_, MODE_>), grid, blk, 0, st, p); break;                    \
  case 3: hipLaunchKernelGGL((seg_lane_kernel<3, 8, RED_, MODE_>), grid, blk, 0, st, p); break;                    \
  case 4: hipLaunchKernelGGL((seg_lane_kernel<4, 8, RED_, MODE_>), grid, blk, 0, st, p); break;                    \
  case 5: hipLaunchKernelGGL((seg_lane_kernel<5, 4, RED_, MODE_>), grid, blk, 0, st, p); break;                    \
  case 6: hipLaunchKernelGGL((seg_lane_kernel<6, 4, RED_, MODE_>), grid, blk, 0, st, p); break;                    \
  case 7: hipLaunchKernelGGL((seg_lane_kernel<7, 4, RED_, MODE_>), grid, blk, 0, st, p); break;                    \
  default: hipLaunchKernelGGL((seg_lane_kernel<8, 4, RED_, MODE_>), grid, blk, 0, st, p); break;                   \
  }
          if (mode == 0) {
            switch (red) {
            case RED_MAX: GEOT_LANE_F(RED_MAX) break;
            case RED_MIN: GEOT_LANE_F(RED_MIN) break;
            case RED_PROD: GEOT_LANE_F(RED_PROD) break;
            default: GEOT_LANE_F(RED_SUM) break;
            }
          } else if (mode == 1) {
            switch (red) {
            case RED_MAX: GEOT_LANE_G(RED_MAX, 1) break;
            case RED_MIN: GEOT_LANE_G(RED_MIN, 1) break;
            default: GEOT_LANE_G(RED_SUM, 1) break;
            }
          } else {
            switch (red) {
            case RED_MAX: GEOT_LANE_G(RED_MAX, 2) break;
            case RED_MIN: GEOT_LANE_G(RED_MIN, 2) break;
            default: GEOT_LANE_G(RED_SUM, 2) break;
            }
          }
#undef GEOT_LANE_G
#undef GEOT_LANE_F
          narrow = true;
        } else if (narrow_path) {
          const dim3 grid((unsigned)P.num_tiles), blk(kThreads);
          t_last_kernel = "seg_narrow_kernel<" + std::to_string((int)F) + ", " + std::to_string(scan_steps(F)) + ">";
          switch ((int)F) {
          case 1: hipLaunchKernelGGL((seg_narrow_kernel<1, 8>), grid, blk, 0, st, p); break;
          case 2: hipLaunchKernelGGL((seg_narrow_kernel<2, 8>), grid, blk, 0, st, p); break;
          case 3: hipLaunchKernelGGL((seg_narrow_kernel<3, 4>), grid, blk, 0, st, p); break;
          case 4: hipLaunchKernelGGL((seg_narrow_kernel<4, 4>), grid, blk, 0, st, p); break;
          case 5: hipLaunchKernelGGL((seg_narrow_kernel<5, 4>), grid, blk, 0, st, p); break;
          case 6: hipLaunchKernelGGL((seg_narrow_kernel<6, 4>), grid, blk, 0, st, p); break;
          case 7: hipLaunchKernelGGL((seg_narrow_kernel<7, 4>), grid, blk, 0, st, p); break;
          default: hipLaunchKernelGGL((seg_narrow_kernel<8, 4>), grid, blk, 0, st, p); break;
          }
          narrow = true;
        }
      }
      if (!narrow && red != RED_SUM) {
        switch (red) {
        case RED_MAX: rc = dispatch_reduce<T, RED_MAX>(p, P, st, mode); break;
        case RED_MEAN: rc = dispatch_reduce<T, RED_MEAN>(p, P, st, mode); break;
        case RED_MIN: rc = dispatch_reduce<T, RED_MIN>(p, P, st, mode); break;
        default: rc = dispatch_reduce<T, RED_PROD>(p, P, st, mode); break;
        }
      } else if (!narrow)
      switch (mode) {
      case 0: rc = dispatch_vec<T, false, 0, false>(p, P, st, nt); break;
      case 1: rc = dispatch_vec<T, true, 0, false>(p, P, st, nt); break;
      case 2: rc = dispatch_vec<T, true, 1, false>(p, P, st, nt); break;
      case 3: rc = dispatch_vec<T, true, 2, false>(p, P, st, nt); break;
      case 4: rc = dispatch_vec<T, true, 3, false>(p, P, st, nt); break;
      default: return fail(GEOT_EINVAL, "bad mode");
      }
    }
    if (rc != GEOT_OK) return rc;
    HIP_TRY(hipGetLastError());
  }
  if (prof) HIP_TRY(hipEventRecord(rec.e2, st));
  rec.has_fix = false;
  if (nnz > 0 && sorted) {
    const int64_t tiles_per_block = (kThreads / 64) * (64 >> P.lpr_log2); // one lane group per tile
    int64_t blocks = (P.num_tiles + tiles_per_block - 1) / tiles_per_block;
    if (blocks < 1) blocks = 1;
    if (use_wsum) launch_wsum<T>(p, P.num_tiles, red, st);
    launch_fixup<T>(p, blocks, P.num_tiles, red, st);
    HIP_TRY(hipGetLastError());
    rec.has_fix = true;
  }
  if (prof) {
    HIP_TRY(hipEventRecord(rec.e3, st));
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.recs.push_back(rec);
  }
  return GEOT_OK;
#endif
}

#define GEOT_SEG_ARGS                                                                                                   \
  int mode, bool sorted, const int64_t *si, const int64_t *di, const void *w, const void *src, void *dst, int64_t nnz,  \
      int64_t F, int64_t H, int64_t src_rows, int64_t K, void *ws, size_t wsb, hipStream_t st, int red
#define GEOT_SEG_PASS mode, sorted, si, di, w, src, dst, nnz, F, H, src_rows, K, ws, wsb, st, red

#if GEOT_SEG_PART
} // namespace
namespace geot_seg {
#if GEOT_SEG_PART == 1
int run_part_f32(GEOT_SEG_ARGS) { return run_segment_op<float>(GEOT_SEG_PASS); }
#elif GEOT_SEG_PART == 2
int run_part_f64(GEOT_SEG_ARGS) { return run_segment_op<double>(GEOT_SEG_PASS); }
#elif GEOT_SEG_PART == 3
int run_part_f16(GEOT_SEG_ARGS) { return run_segment_op<half_t>(GEOT_SEG_PASS); }
#else
int run_part_bf16(GEOT_SEG_ARGS) { return run_segment_op<bf16_t>(GEOT_SEG_PASS); }
#endif
} // namespace geot_seg
#else // part 0: the state, the small kernels' launchers, the C ABI
#if GEOT_SEG_SPLIT
} // namespace
namespace geot_seg {
int run_part_f32(GEOT_SEG_ARGS);
int run_part_f64(GEOT_SEG_ARGS);
int run_part_f16(GEOT_SEG_ARGS);
int run_part_bf16(GEOT_SEG_ARGS);
} // namespace geot_seg
namespace {
#else
int run_part_f32(GEOT_SEG_ARGS) { return run_segment_op<float>(GEOT_SEG_PASS); }
int run_part_f64(GEOT_SEG_ARGS) { return run_segment_op<double>(GEOT_SEG_PASS); }
int run_part_f16(GEOT_SEG_ARGS) { return run_segment_op<half_t>(GEOT_SEG_PASS); }
int run_part_bf16(GEOT_SEG_ARGS) { return run_segment_op<bf16_t>(GEOT_SEG_PASS); }
#endif

int run_typed(int dtype, int mode, bool sorted, const int64_t *si, const int64_t *di,
              const void *w, const void *src, void *dst, int64_t nnz, int64_t F, int64_t H,
              int64_t src_rows, int64_t K, void *ws, size_t wsb, void *stream, int red = RED_SUM) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == GEOT_F32) return run_part_f32(GEOT_SEG_PASS);
  if (dtype == GEOT_F64) return run_part_f64(GEOT_SEG_PASS);
  if (dtype == GEOT_F16) return run_part_f16(GEOT_SEG_PASS);
  if (dtype == GEOT_BF16) return run_part_bf16(GEOT_SEG_PASS);
  return fail(GEOT_EINVAL, "dtype must be GEOT_F32, GEOT_F64, GEOT_F16 or GEOT_BF16");
}

template <typename T> int pick_row_vec(int64_t F, const void *a, const void *b) {
  const int maxvec = 16 / (int)sizeof(T);
  if (is_aligned16(a) && is_aligned16(b)) {
    if (F % maxvec == 0) return maxvec;
    if (maxvec >= 4 && F % 2 == 0 ) return 2;
  }
  return 1;
}

// H > 1: the multi-head form - F values per head, rows of H * F values, results out[e * H + h] or (head_major) out[h * nnz + e]
template <typename T>
int run_sddmm(const int64_t *si, const int64_t *di, const void *m1, const void *m2, void *out,
              int64_t nnz, int64_t F, int64_t rows1, int64_t rows2, hipStream_t st, int64_t H = 1, int head_major = 0) {
  if (nnz < 0 || F < 0 || H < 1) return fail(GEOT_EINVAL, "negative size");
  if (nnz == 0 || (H > 1 && F == 0)) return GEOT_OK;
  if (!si || !di || !m1 || !m2 || !out) return fail(GEOT_EINVAL, "null pointer");
  if (H > 64) return fail(GEOT_EUNSUPPORTED, "more than 64 heads");
  const int64_t stride = H * F;
  nnz *= H;                                               // the loop runs over (edge, head) pairs
  int vec = pick_row_vec<T>(F, m1, m2);
  int l = ceil_log2((F + vec - 1) / vec);
  if (l > 6) l = 6;
  // One lane per 16-byte piece of the row (the gather kernels' layout) makes the dot product's cross-lane reduction the
  // bottleneck: 5 shuffle steps per edge at 32 lanes per row, two edges per wave step.  Rows of 16 lanes get 8, wider rows a
  // quarter of their lanes, each lane walking 2-4 pieces (tools/_ab sweep, 20 M / 2 M edges, fp32 and bf16, uniform-random and
  // local sources: F=64 1.1-1.4x, F=128 1.2-1.8x, F=256 1.2-1.95x; rows of <= 8 lanes: unchanged, fewer lanes measured up to 30 % slower).
  // (16-bit rows of 8 lanes: 4 lanes from ~8 M edges on, +15-25 %; at 2 M edges 9 % slower)
  const int shift = g_sddmm_shift >= 0 ? (int)g_sddmm_shift : (l >= 5 ? 2 : (l == 4 ? 1 : (l == 3 && sizeof(T) == 2 && nnz >= 8000000 ? 1 : 0)));
  l -= shift;
  if (l < 2) l = 2;
  const int ng = kThreads >> l;
  int64_t chunk = (int64_t)ng * 64;                       // 64 edges per lane group and block
  int64_t blocks = (nnz + chunk - 1) / chunk;
  const T *a = static_cast<const T *>(m1), *b = static_cast<const T *>(m2);
  T *o = static_cast<T *>(out);
  constexpr int MAXV = 16 / (int)sizeof(T);
  if (vec == MAXV)
    hipLaunchKernelGGL((sddmm_coo_kernel<T, MAXV>), dim3((unsigned)blocks), dim3(kThreads), 0, st, si, di, a, b, o, nnz, F, rows1, rows2, l, chunk, g_xcd, (int)H, stride, head_major);
  else if (vec == 2 && MAXV >= 4)
    hipLaunchKernelGGL((sddmm_coo_kernel<T, 2>), dim3((unsigned)blocks), dim3(kThreads), 0, st, si, di, a, b, o, nnz, F, rows1, rows2, l, chunk, g_xcd, (int)H, stride, head_major);
  else
    hipLaunchKernelGGL((sddmm_coo_kernel<T, 1>), dim3((unsigned)blocks), dim3(kThreads), 0, st, si, di, a, b, o, nnz, F, rows1, rows2, l, chunk, g_xcd, (int)H, stride, head_major);
  HIP_TRY(hipGetLastError());
  return GEOT_OK;
}

template <typename T>
int run_gather_rows(const int64_t *index, const void *src, void *dst, int64_t nnz, int64_t F,
                    int64_t src_rows, hipStream_t st) {
  if (nnz < 0 || F < 0) return fail(GEOT_EINVAL, "negative size");
  if (nnz == 0 || F == 0) return GEOT_OK;
  if (!index || !src || !dst) return fail(GEOT_EINVAL, "null pointer");
  int vec = pick_row_vec<T>(F, src, dst);
  int l = ceil_log2((F + vec - 1) / vec);
  if (l > 6) l = 6;
  const int ng = kThreads >> l;
  int64_t blocks = (nnz + ng - 1) / ng;
  if (blocks > 256 * 32) blocks = 256 * 32;
  const T *s = static_cast<const T *>(src);
  T *d = static_cast<T *>(dst);
  constexpr int MAXV = 16 / (int)sizeof(T);
  if (vec == MAXV)
    hipLaunchKernelGGL((gather_rows_kernel<T, MAXV>), dim3((unsigned)blocks), dim3(kThreads), 0, st, index, s, d, nnz, F, src_rows, l);
  else if (vec == 2 && MAXV >= 4)
    hipLaunchKernelGGL((gather_rows_kernel<T, 2>), dim3((unsigned)blocks), dim3(kThreads), 0, st, index, s, d, nnz, F, src_rows, l);
  else
    hipLaunchKernelGGL((gather_rows_kernel<T, 1>), dim3((unsigned)blocks), dim3(kThreads), 0, st, index, s, d, nnz, F, src_rows, l);
  HIP_TRY(hipGetLastError());
  return GEOT_OK;
}

template <typename T>
int run_select_backward(const int64_t *si, const int64_t *di, const void *w, const void *x, const void *out, const void *grad, void *ties,
                               void *gsrc, void *gw, int64_t nnz, int64_t F, int64_t src_rows, int64_t K, hipStream_t st) {
  if (nnz < 0 || F < 0 || src_rows < 0 || K < 0) return fail(GEOT_EINVAL, "negative size");
  if (!gsrc || !ties || (nnz > 0 && (!si || !di || !x || !out || !grad))) return fail(GEOT_EINVAL, "null pointer");
  if (geot_internal_fill(gsrc, (size_t)src_rows * (size_t)F * sizeof(T), 0u, st) != GEOT_OK) return GEOT_ELAUNCH;   // (fp32 / fp64: whole words)
  if (geot_internal_fill(ties, (size_t)K * (size_t)F * sizeof(T), 0u, st) != GEOT_OK) return GEOT_ELAUNCH;
  if (nnz == 0 || F == 0) return GEOT_OK;
  int l = ceil_log2(F);
  if (l > 6) l = 6;
  if (l < 2) l = 2;
  const int ng = kThreads >> l;
  int64_t blocks = (nnz + ng - 1) / ng;
  if (blocks > 256 * 64) blocks = 256 * 64;
  const dim3 grid((unsigned)blocks), blk(kThreads);
  hipLaunchKernelGGL((select_backward_kernel<T, 0>), grid, blk, 0, st, si, di, static_cast<const T *>(w), static_cast<const T *>(x), static_cast<const T *>(out),
                     static_cast<const T *>(grad), static_cast<T *>(ties), static_cast<T *>(gsrc), static_cast<T *>(gw), nnz, F, src_rows, K, l);
  hipLaunchKernelGGL((select_backward_kernel<T, 1>), grid, blk, 0, st, si, di, static_cast<const T *>(w), static_cast<const T *>(x), static_cast<const T *>(out),
                     static_cast<const T *>(grad), static_cast<T *>(ties), static_cast<T *>(gsrc), static_cast<T *>(gw), nnz, F, src_rows, K, l);
  HIP_TRY(hipGetLastError());
  return GEOT_OK;
}

} // namespace

extern "C" {

int geot_internal_fail(int code, const char *msg) { return fail(code, msg ? msg : ""); }
int geot_internal_fill(void *p, unsigned long bytes, unsigned int word, void *stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (bytes == 0) return GEOT_OK;
  if (!p || (bytes & 3) || (((uintptr_t)p) & 3)) return fail(GEOT_EINVAL, "internal: fill wants whole, aligned 32-bit words");
  const size_t words = bytes / 4;
  const bool quads = (((uintptr_t)p) & 15) == 0 && words >= 4;
  size_t blocks = ((quads ? words / 4 : words) + kThreads - 1) / kThreads;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  if (quads) hipLaunchKernelGGL(fill_quads_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, st, static_cast<uint32_t *>(p), words, (uint32_t)word);
  else hipLaunchKernelGGL(fill_words_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, st, static_cast<uint32_t *>(p), words, (uint32_t)word);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? GEOT_OK : fail(GEOT_ELAUNCH, hipGetErrorString(e));
}

void geot_internal_note_kernel(const char *name) { t_last_kernel = name ? name : ""; }
const char *geot_last_kernel(void) { return t_last_kernel.c_str(); }

int geot_abi_version(void) { return GEOT_ABI_VERSION; }

const char *geot_last_error(void) { return g_err.c_str(); }

#ifdef GEOT_DEV_EXPERIMENTS
const char *geot_build_info(void) { return "libgeot_hip_dev gfx950 (CDNA4) DEVELOPMENT build (experiment switches) built " __DATE__ " " __TIME__; }
#else
const char *geot_build_info(void) { return "libgeot_hip gfx950 (CDNA4) built " __DATE__ " " __TIME__; }
#endif

// exact need of every plan the launcher can pick for these sizes (storage alignment unknown here:
// both vector widths are priced), so the answer is tight: cfg2 needs ~11 MB, not a 64-edge-tile bound
static size_t plan_bytes(int64_t nnz, int64_t feat, int64_t vec_unit, int64_t out_rows, int tsize, bool gather,
                         int hw) {
  size_t best = 0;
  for (int aligned = 0; aligned < 3; ++aligned) // every shape the operands' alignment can lead to: off the grid, 4-byte (ragged lanes), 16-byte
    for (int atomic_flush = 0; atomic_flush < (gather ? 1 : 2); ++atomic_flush) {
      const Plan P = make_plan(nnz, feat, vec_unit, out_rows, tsize, aligned == 2, gather, hw, atomic_flush != 0, 0, aligned >= 1);
      if (P.total > best) best = P.total;
    }
  return best;
}

static int storage_size(int dtype) { return dtype == GEOT_F64 ? 8 : (dtype == GEOT_F32 ? 4 : 2); }

size_t geot_workspace_bytes(int64_t nnz, int64_t feat, int64_t out_rows, int dtype) {
  const int tsize = storage_size(dtype);
  if (nnz < 0) nnz = 0;
  if (feat < 1) feat = 1;
  if (out_rows < 0) out_rows = 0;
  size_t need = plan_bytes(nnz, feat, feat, out_rows, tsize, false, 0);          // index_scatter (+ unsorted)
  const size_t g = plan_bytes(nnz, feat, feat, out_rows, tsize, true, 1);        // gather_scatter / gws
  if (g > need) need = g;
  if (feat <= kNarrowMaxF && dtype == GEOT_F32) {                                 // the narrow-row kernels' tiles
    for (int lane_seq = 0; lane_seq < 2; ++lane_seq) {
      const size_t nb = narrow_plan(nnz, feat, out_rows, lane_seq != 0).total;
      if (nb > need) need = nb;
    }
  }
  return need;
}

size_t geot_mh_workspace_bytes(int64_t nnz, int64_t heads, int64_t feat, int64_t out_rows, int dtype) {
  if (nnz < 0) nnz = 0;
  if (heads < 1) heads = 1;
  if (feat < 1) feat = 1;
  if (out_rows < 0) out_rows = 0;
  return plan_bytes(nnz, heads * feat, feat, out_rows, storage_size(dtype), true, (int)heads);
}

int geot_workspace_init(void *workspace, size_t workspace_bytes, void *stream) {
  if (!workspace || workspace_bytes < kCtrlBytes) return fail(GEOT_EWORKSPACE, "workspace too small");
  HIP_TRY(hipMemsetAsync(workspace, 0, kCtrlBytes, static_cast<hipStream_t>(stream)));
  return GEOT_OK;
}

int geot_index_scatter(const int64_t *index, const void *src, void *dst, int64_t nnz,
                       int64_t feat, int64_t out_rows, int dtype, int sorted, void *workspace,
                       size_t workspace_bytes, void *stream) {
  return run_typed(dtype, 0, sorted != 0, nullptr, index, nullptr, src, dst, nnz, feat, 1, nnz,
                   out_rows, workspace, workspace_bytes, stream);
}

int geot_index_scatter_reduce(const int64_t *index, const void *src, void *dst, int64_t nnz,
                              int64_t feat, int64_t out_rows, int dtype, int reduce, void *workspace,
                              size_t workspace_bytes, void *stream) {
  if (reduce < GEOT_REDUCE_MAX || reduce > GEOT_REDUCE_PROD) return fail(GEOT_EINVAL, "bad reduce code");
  return run_typed(dtype, 0, true, nullptr, index, nullptr, src, dst, nnz, feat, 1, nnz, out_rows,
                   workspace, workspace_bytes, stream, reduce);
}

int geot_gather_reduce(const int64_t *src_index, const int64_t *dst_index, const void *weight, const void *src,
                       void *dst, int64_t nnz, int64_t feat, int64_t src_rows, int64_t out_rows, int dtype,
                       int reduce, void *workspace, size_t workspace_bytes, void *stream) {
  if (reduce < GEOT_REDUCE_MAX || reduce > GEOT_REDUCE_PROD) return fail(GEOT_EINVAL, "bad reduce code");
  return run_typed(dtype, weight ? 2 : 1, true, src_index, dst_index, weight, src, dst, nnz, feat, 1, src_rows,
                   out_rows, workspace, workspace_bytes, stream, reduce);
}

int geot_gather_scatter(const int64_t *src_index, const int64_t *dst_index, const void *src,
                        void *dst, int64_t nnz, int64_t feat, int64_t src_rows,
                        int64_t out_rows, int dtype, void *workspace, size_t workspace_bytes,
                        void *stream) {
  return run_typed(dtype, 1, true, src_index, dst_index, nullptr, src, dst, nnz, feat, 1,
                   src_rows, out_rows, workspace, workspace_bytes, stream);
}

int geot_gather_weight_scatter(const int64_t *src_index, const int64_t *dst_index,
                               const void *weight, const void *src, void *dst, int64_t nnz,
                               int64_t feat, int64_t src_rows, int64_t out_rows, int dtype,
                               void *workspace, size_t workspace_bytes, void *stream) {
  return run_typed(dtype, 2, true, src_index, dst_index, weight, src, dst, nnz, feat, 1,
                   src_rows, out_rows, workspace, workspace_bytes, stream);
}

int geot_mh_spmm(const int64_t *src_index, const int64_t *dst_index, const void *weight,
                 const void *src, void *dst, int64_t nnz, int64_t heads, int64_t feat,
                 int64_t src_rows, int64_t out_rows, int weight_layout, int dtype,
                 void *workspace, size_t workspace_bytes, void *stream) {
  if (heads < 1 || feat < 0) return fail(GEOT_EINVAL, "heads must be >= 1");
  if (weight_layout != GEOT_W_EDGE_MAJOR && weight_layout != GEOT_W_HEAD_MAJOR)
    return fail(GEOT_EINVAL, "Invalid weight size");
  return run_typed(dtype, weight_layout == GEOT_W_EDGE_MAJOR ? 3 : 4, true, src_index, dst_index,
                   weight, src, dst, nnz, heads * feat, heads, src_rows, out_rows, workspace,
                   workspace_bytes, stream);
}

int geot_sddmm_coo(const int64_t *src_index, const int64_t *dst_index, const void *mat_1,
                   const void *mat_2, void *out, int64_t nnz, int64_t feat, int64_t rows_1,
                   int64_t rows_2, int dtype, void *stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == GEOT_F32) return run_sddmm<float>(src_index, dst_index, mat_1, mat_2, out, nnz, feat, rows_1, rows_2, st);
  if (dtype == GEOT_F64) return run_sddmm<double>(src_index, dst_index, mat_1, mat_2, out, nnz, feat, rows_1, rows_2, st);
  if (dtype == GEOT_F16) return run_sddmm<half_t>(src_index, dst_index, mat_1, mat_2, out, nnz, feat, rows_1, rows_2, st);
  if (dtype == GEOT_BF16) return run_sddmm<bf16_t>(src_index, dst_index, mat_1, mat_2, out, nnz, feat, rows_1, rows_2, st);
  return fail(GEOT_EINVAL, "bad dtype");
}

int geot_mh_sddmm_coo(const int64_t *src_index, const int64_t *dst_index, const void *mat_1, const void *mat_2, void *out, int64_t nnz,
                      int64_t heads, int64_t feat, int64_t rows_1, int64_t rows_2, int weight_layout, int dtype, void *stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (weight_layout != GEOT_W_EDGE_MAJOR && weight_layout != GEOT_W_HEAD_MAJOR) return fail(GEOT_EINVAL, "bad weight layout");
  const int hm = weight_layout == GEOT_W_HEAD_MAJOR ? 1 : 0;
  if (dtype == GEOT_F32) return run_sddmm<float>(src_index, dst_index, mat_1, mat_2, out, nnz, feat, rows_1, rows_2, st, heads, hm);
  if (dtype == GEOT_F64) return run_sddmm<double>(src_index, dst_index, mat_1, mat_2, out, nnz, feat, rows_1, rows_2, st, heads, hm);
  if (dtype == GEOT_F16) return run_sddmm<half_t>(src_index, dst_index, mat_1, mat_2, out, nnz, feat, rows_1, rows_2, st, heads, hm);
  if (dtype == GEOT_BF16) return run_sddmm<bf16_t>(src_index, dst_index, mat_1, mat_2, out, nnz, feat, rows_1, rows_2, st, heads, hm);
  return fail(GEOT_EINVAL, "bad dtype");
}

int geot_gather_select_backward(const int64_t *src_index, const int64_t *dst_index, const void *weight, const void *src, const void *out,
                                const void *grad, void *ties, void *grad_src, void *grad_weight, int64_t nnz, int64_t feat, int64_t src_rows,
                                int64_t out_rows, int dtype, void *stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == GEOT_F32) return run_select_backward<float>(src_index, dst_index, weight, src, out, grad, ties, grad_src, grad_weight, nnz, feat, src_rows, out_rows, st);
  if (dtype == GEOT_F64) return run_select_backward<double>(src_index, dst_index, weight, src, out, grad, ties, grad_src, grad_weight, nnz, feat, src_rows, out_rows, st);
  return fail(GEOT_EUNSUPPORTED, "gather_select_backward: float32 / float64 (float atomics)");
}

int geot_gather_rows(const int64_t *index, const void *src, void *dst, int64_t nnz,
                     int64_t feat, int64_t src_rows, int dtype, void *stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == GEOT_F32) return run_gather_rows<float>(index, src, dst, nnz, feat, src_rows, st);
  if (dtype == GEOT_F64) return run_gather_rows<double>(index, src, dst, nnz, feat, src_rows, st);
  if (dtype == GEOT_F16) return run_gather_rows<half_t>(index, src, dst, nnz, feat, src_rows, st);
  if (dtype == GEOT_BF16) return run_gather_rows<bf16_t>(index, src, dst, nnz, feat, src_rows, st);
  return fail(GEOT_EINVAL, "bad dtype");
}

int geot_publish_word(const int64_t *device_word, int64_t *host_slot2, int64_t seq) {
  if (!device_word || !host_slot2) return fail(GEOT_EINVAL, "publish_word: null pointer");
  t_pub.armed = true;
  t_pub.src = device_word;
  t_pub.dst = host_slot2;
  t_pub.seq = seq;
  return GEOT_OK;
}

int geot_set_alarm_word(int64_t *host_slot2) {
  t_alarm = host_slot2;
  return GEOT_OK;
}

int geot_publish_pending(void) {
  const bool armed = t_pub.armed;
  t_pub.armed = false;
  return armed ? 1 : 0;
}

int geot_index_probe(const int64_t *index, int64_t nnz, int64_t *out2, void *stream) {
  if (nnz <= 0 || !index || !out2) return fail(GEOT_EINVAL, "index_probe: needs a non-empty index and an output");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (nnz <= 32768) {
    hipLaunchKernelGGL(index_probe_small_kernel, dim3(1), dim3(kThreads), 0, st, index, nnz, out2);
    HIP_TRY(hipGetLastError());
    return GEOT_OK;
  }
  if (geot_internal_fill(out2, 2 * sizeof(int64_t), 0u, st) != GEOT_OK) return GEOT_ELAUNCH;
  int64_t blocks = (nnz + kThreads * 4 - 1) / (kThreads * 4);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(index_probe_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, st, index, nnz, out2);
  HIP_TRY(hipGetLastError());
  return GEOT_OK;
}

size_t geot_csr_workspace_bytes(int64_t nnz, int64_t feat, int64_t out_rows, int dtype) {
  return geot_workspace_bytes(nnz, feat, out_rows, dtype) + up256((size_t)(nnz > 0 ? nnz : 1) * sizeof(int64_t));
}

int geot_csr_gws(const int64_t *indptr, const int64_t *indices, const void *weight, const void *src,
                 void *dst, int64_t nrow, int64_t nnz, int64_t feat, int64_t src_rows, int64_t out_rows,
                 int dtype, void *workspace, size_t workspace_bytes, void *stream) {
  if (nrow < 0 || nnz < 0 || out_rows < nrow) return fail(GEOT_EINVAL, "csr_gws: bad sizes");
  if (!workspace || ((uintptr_t)workspace & 255) != 0) return fail(GEOT_EWORKSPACE, "workspace must be 256-byte aligned");
  const size_t seg = geot_workspace_bytes(nnz, feat, out_rows, dtype);
  const size_t need = seg + up256((size_t)(nnz > 0 ? nnz : 1) * sizeof(int64_t));
  if (workspace_bytes < need) return fail(GEOT_EWORKSPACE, "workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  int64_t *dst_index = reinterpret_cast<int64_t *>(static_cast<char *>(workspace) + seg);
  if (nnz > 0) {
    if (!indptr) return fail(GEOT_EINVAL, "null indptr");
    int64_t blocks = (nrow + kThreads / 16 - 1) / (kThreads / 16);
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((csr_expand_kernel<int64_t>), dim3((unsigned)blocks), dim3(kThreads), 0, st, indptr, nrow, nnz, dst_index);
    HIP_TRY(hipGetLastError());
  }
  if (weight)
    return run_typed(dtype, 2, true, indices, dst_index, weight, src, dst, nnz, feat, 1, src_rows, out_rows, workspace, seg, stream);
  return run_typed(dtype, 1, true, indices, dst_index, nullptr, src, dst, nnz, feat, 1, src_rows, out_rows, workspace, seg, stream);
}

int geot_coo_to_csr(const int64_t *coo_row, int64_t nnz, int64_t nrow, int32_t *rowptr, int assume_sorted,
                    void *stream) {
  if (nnz < 0 || nrow < 0 || !rowptr || (nnz > 0 && !coo_row)) return fail(GEOT_EINVAL, "coo_to_csr: bad arguments");
  if (nnz >= ((int64_t)1 << 31)) return fail(GEOT_EUNSUPPORTED, "coo_to_csr: int32 row pointers need nnz < 2^31");
  hipStream_t st = static_cast<hipStream_t>(stream);
  int64_t blocks = (nnz + 1 + kThreads - 1) / kThreads;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  if (assume_sorted) {
    hipLaunchKernelGGL(coo_sorted_to_csr_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, st, coo_row, nnz, nrow, rowptr);
  } else {
    // hist into rowptr[1..nrow], the caller turns it into an inclusive prefix sum
    if (geot_internal_fill(rowptr, (size_t)(nrow + 1) * sizeof(int32_t), 0u, st) != GEOT_OK) return GEOT_ELAUNCH;
    hipLaunchKernelGGL(coo_hist_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, st, coo_row, nnz, nrow, rowptr + 1);
  }
  HIP_TRY(hipGetLastError());
  return GEOT_OK;
}

void geot_profile_enable(int on) { g_prof.on = on != 0; }

static void prof_drain_locked() {
  for (auto &r : g_prof.recs) {
    hipEventSynchronize(r.e3);
    float a = 0, m = 0, f = 0;
    hipEventElapsedTime(&a, r.e0, r.e1);
    hipEventElapsedTime(&m, r.e1, r.e2);
    hipEventElapsedTime(&f, r.e2, r.e3);
    g_prof.aux_ms += a;
    g_prof.main_ms += m;
    g_prof.fix_ms += f;
    g_prof.calls += 1;
    g_prof.pool.push_back(r.e0);
    g_prof.pool.push_back(r.e1);
    g_prof.pool.push_back(r.e2);
    g_prof.pool.push_back(r.e3);
  }
  g_prof.recs.clear();
}

void geot_profile_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  prof_drain_locked();
  g_prof.main_ms = g_prof.fix_ms = g_prof.aux_ms = 0;
  g_prof.calls = 0;
}

int geot_profile_read(double *main_ms, double *fixup_ms, double *aux_ms, int64_t *calls) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  prof_drain_locked();
  if (main_ms) *main_ms = g_prof.main_ms;
  if (fixup_ms) *fixup_ms = g_prof.fix_ms;
  if (aux_ms) *aux_ms = g_prof.aux_ms;
  if (calls) *calls = g_prof.calls;
  return GEOT_OK;
}

int geot_profile_box(const void *buf, size_t bytes, int iters, double *read_gbps, double *sclk_mhz, void *stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!buf || bytes < (1u << 20) || iters < 1) return fail(GEOT_EINVAL, "profile_box: needs a device buffer of >= 1 MiB");
  unsigned long long *d = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  struct Cleanup { // (every early return below frees what has been made so far)
    unsigned long long *&d;
    hipEvent_t &e0, &e1;
    ~Cleanup() {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
      if (d) (void)hipFree(d);
    }
  } cleanup{d, e0, e1};
  HIP_TRY(hipMalloc(&d, 64));
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  float best = 1e30f;
  for (int it = 0; it < 2 * iters + 2; ++it) {
    HIP_TRY(hipEventRecord(e0, st));
    // best of two grid shapes (the faster one differs between devices of the pool)
    hipLaunchKernelGGL(box_read_kernel, dim3(256 * ((it & 1) ? 8 : 32)), dim3(kThreads), 0, st, static_cast<const float *>(buf),
                       reinterpret_cast<float *>(d), bytes / 16);
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (it > 1 && ms < best) best = ms; // the first pass of each shape warms up
  }
  if (read_gbps) *read_gbps = (double)(bytes / 16 * 16) / (best * 1e-3) / 1e9;
  hipLaunchKernelGGL(box_clock_kernel, dim3(1), dim3(64), 0, st, d, 200000);
  unsigned long long h[3] = {0, 0, 0};
  HIP_TRY(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (sclk_mhz) *sclk_mhz = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0; // s_memrealtime ticks at 100 MHz
  return GEOT_OK;
}

int geot_profile_box_rows(const void *table, int64_t rows, int64_t row_bytes, int iters, double *row_gbps, double *row_gbps_nt, void *out,
                          int64_t out_rows, int run, double *mix_row_gbps, void *stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t piece16 = row_bytes / 16;
  if (!table || rows < 1 || rows >= (int64_t(1) << 32) || row_bytes < 16 || row_bytes % 16 || piece16 > 64 || (piece16 & (piece16 - 1)) || iters < 1)
    return fail(GEOT_EINVAL, "profile_box_rows: needs a device table of 1 .. 2^32 - 1 rows of 16 * 2^k <= 1024 bytes");
  if (mix_row_gbps && (!out || out_rows < 1 || run < 1)) return fail(GEOT_EINVAL, "profile_box_rows: the read / write mix needs an output buffer and a run length");
  float *d = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  struct Cleanup {
    float *&d;
    hipEvent_t &e0, &e1;
    ~Cleanup() {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
      if (d) (void)hipFree(d);
    }
  } cleanup{d, e0, e1};
  HIP_TRY(hipMalloc(&d, 64));
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  // ~16 GB of row reads a pass (a few ms): 256 CUs x 8 workgroups, every lane group 16 rows a step
  const int grid = 256 * 8;
  const int64_t groups = (int64_t)grid * kThreads / piece16;
  int steps = (int)(((int64_t(16) << 30) / row_bytes / groups + 15) / 16 * 16);
  if (steps < 16) steps = 16;
  const double bytes = (double)groups * steps * row_bytes;
  float best[3] = {1e30f, 1e30f, 1e30f};
  const int forms = mix_row_gbps ? 3 : 2;
  for (int it = 0; it < forms * (iters + 1); ++it) {
    const int form = it % forms;
    const float *t = static_cast<const float *>(table);
    HIP_TRY(hipEventRecord(e0, st));
    if (form == 0)
      hipLaunchKernelGGL((box_rows_kernel<false, false>), dim3(grid), dim3(kThreads), 0, st, t, d, (unsigned long long)rows, (int)piece16, steps, 12345u + it, 0, nullptr, 1ull);
    else if (form == 1)
      hipLaunchKernelGGL((box_rows_kernel<true, false>), dim3(grid), dim3(kThreads), 0, st, t, d, (unsigned long long)rows, (int)piece16, steps, 12345u + it, 0, nullptr, 1ull);
    else
      hipLaunchKernelGGL((box_rows_kernel<true, true>), dim3(grid), dim3(kThreads), 0, st, t, d, (unsigned long long)rows, (int)piece16, steps, 12345u + it, run,
                         static_cast<float *>(out), (unsigned long long)out_rows);
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (it >= forms && ms < best[form]) best[form] = ms; // the first pass of each form warms up
  }
  if (row_gbps) *row_gbps = bytes / (best[0] * 1e-3) / 1e9;
  if (row_gbps_nt) *row_gbps_nt = bytes / (best[1] * 1e-3) / 1e9;
  if (mix_row_gbps) *mix_row_gbps = bytes / (best[2] * 1e-3) / 1e9;
  return GEOT_OK;
}

int geot_set_option(const char *name, int value) {
  if (!name) return fail(GEOT_EINVAL, "set_option: null name");
  const std::string n(name);
  if (n == "unroll") g_unroll = value;
  else if (n == "gather_grid") g_gather_grid = value;
  else if (n == "sddmm_shift") g_sddmm_shift = value;
  else if (n == "lds_floor") g_lds_floor = value;
  else if (n == "narrow") g_narrow = value;
  else if (n == "handoff") g_handoff = value;
  else if (n == "ragged") g_ragged = value != 0;
  else if (n == "handoff_tries") g_handoff_tries = value;
  else if (n == "hub") g_hub = value;
  else if (n == "lane_e") g_lane_e = value;
  else if (n == "xcd") g_xcd = value;
  else if (n == "nt_keys") g_nt_keys = value;
  else if (!geot_internal_slab_option(name, value))
    // (the switches of measured-and-rejected variants - slab_probe, slab_pair, slab_wrow_all, slab_nt, slab_tight, slab_stage,
    // slab_unroll - exist in the development build only, geot_amd/libgeot_hip_dev.so: a name this build does not know is an error,
    // never silently ignored)
    return fail(GEOT_EINVAL, (std::string("set_option: unknown option '") + n + "' in this build (" +
#ifdef GEOT_DEV_EXPERIMENTS
                              "development"
#else
                              "product; experiment switches live in libgeot_hip_dev.so"
#endif
                              + ")").c_str());
  return GEOT_OK;
}

void geot_tune(int edges_per_group, int vec, int nontemporal, int lpr_log2) {
  g_tune.cg = edges_per_group;
  g_tune.vec = vec;
  g_tune.nt = nontemporal;
  g_tune.lpr_log2 = lpr_log2;
}

} // extern "C"
#endif // GEOT_SEG_PART == 0
