/*
 * geot_hip.h -- C ABI of libgeot_hip.so, the MI355X (gfx950) segment-reduction engine.
 *
 * This is the drop-in boundary.  Each entry point replaces one of the reference's
 * device-side entry functions, which the reference's dispatcher shims call after they have
 * computed the output row count and allocated the output (citations are relative to the
 * reference tree):
 *
 *   geot_index_scatter            <- index_scatter_cuda          csrc/cuda/header_cuda.h:4-6
 *                                    (impl csrc/cuda/index_scatter_cuda.cu:86-105)
 *   geot_gather_scatter           <- gather_scatter_cuda         csrc/cuda/header_cuda.h:8-10
 *                                    (impl csrc/cuda/gather_scatter_cuda.cu:15-28)
 *   geot_gather_weight_scatter    <- gather_weight_scatter_cuda  csrc/cuda/header_cuda.h:12-17
 *                                    (impl csrc/cuda/gather_weight_scatter_cuda.cu:22-39)
 *   geot_mh_spmm                  <- mh_spmm_cuda                csrc/cuda/header_cuda.h:23-26
 *                                    (impl csrc/cuda/mh_spmm_cuda.cu:20-38, layout pick
 *                                     csrc/cuda/wrapper/mh_spmm_base.h:38-49)
 *   geot_sddmm_coo                <- sddmm_coo_cuda              csrc/cuda/header_cuda.h:28-30
 *                                    (impl csrc/cuda/gather_weight_scatter_cuda.cu:41-62)
 *   geot_gather_rows              <- gather_eb_sorted_kernel     csrc/cuda/index_scatter_kernel.cuh:266-315
 *                                    (backward of index_scatter; unwired in the reference)
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless stated otherwise;
 *   - indices are int64 (the reference's `Index`, csrc/util/utils.h:9); values are float32
 *     (dtype GEOT_F32) or float64 (GEOT_F64), the AT_DISPATCH_FLOATING_TYPES set of the reference;
 *   - `dst` is written in full by every sorted call: it does NOT need to be zeroed
 *     (the reference requires torch::zeros, csrc/index_scatter.cpp:35, because it flushes with
 *     atomicAdd); rows with no edge come out as 0 exactly as in the reference;
 *   - `out_rows` is the reference's row rule, index[-1] + 1 (csrc/index_scatter.cpp:30-34),
 *     computed by the caller; a larger value is allowed (rows behind the last key come out 0);
 *     keys outside [0, out_rows) are ignored (never written);
 *   - `workspace` is caller-owned scratch of at least geot_workspace_bytes(...) bytes, 256-B
 *     aligned.  Its first 256 bytes are control words that must be ZERO before the first call
 *     (geot_workspace_init, or any zeroing allocation); every call leaves them zero again, so a
 *     workspace can be reused call after call - by one stream at a time - with no per-call
 *     memset.  The rest is scratch whose content is irrelevant on entry;
 *   - `stream` is a hipStream_t (NULL = default stream);
 *   - all launches are asynchronous on `stream`; no host synchronisation, no allocation: the
 *     calls are hipGraph-capturable;
 *   - return value: GEOT_OK or a negative GEOT_E* code; geot_last_error() gives the message
 *     (the reference's TORCH_CHECK texts are reproduced by the host layer above this ABI).
 */
#ifndef GEOT_HIP_H
#define GEOT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): geot_slab_plan carries slab_shift / n_slabs and its `units` select the kernel form (geot_slab_units_for /
 * geot_slab_rows_per_group_shape), split hubs are interleaved, weights may arrive in plan order for the multi-head mode too
 * (weight_mode 5) or are staged into it by the kernel (geot_slab_workspace_bytes_staged), geot_slab_mh_sddmm / geot_mh_sddmm_coo;
 * the measurement hooks and experiment switches moved to geot_hip_dev.h. */
#define GEOT_ABI_VERSION 2

enum {
  GEOT_OK = 0,
  GEOT_EINVAL = -1,     /* bad argument (negative size, null pointer, bad dtype ...)     */
  GEOT_EWORKSPACE = -2, /* workspace too small or misaligned                              */
  GEOT_ELAUNCH = -3,    /* HIP runtime reported an error (see geot_last_error)            */
  GEOT_EUNSUPPORTED = -4
};

/* GEOT_F16 / GEOT_BF16: 16-bit storage, fp32 accumulation, one rounding at the end (what the reference's
 * CPU path does, csrc/cpu/index_scatter_cpu.cpp:78-86); sorted calls only (no 16-bit float atomics). */
enum { GEOT_F32 = 0, GEOT_F64 = 1, GEOT_F16 = 2, GEOT_BF16 = 3 };

/* weight layouts of geot_mh_spmm (csrc/cuda/wrapper/mh_spmm_base.h:38-49) */
enum { GEOT_W_EDGE_MAJOR = 0 /* [nnz, H] */, GEOT_W_HEAD_MAJOR = 1 /* [H, nnz] */ };

int geot_abi_version(void);
const char *geot_last_error(void);
const char *geot_build_info(void); /* "gfx950 <date> ..." */

/* Scratch needed by any of the segment-reduction calls below for an edge list of `nnz`
 * edges, `feat` values per output row (H*F for mh_spmm) and `out_rows` output rows. */
size_t geot_workspace_bytes(int64_t nnz, int64_t feat, int64_t out_rows, int dtype);
/* geot_mh_spmm stages heads weights per edge in LDS, so its tiles (and scratch) depend on `heads` */
size_t geot_mh_workspace_bytes(int64_t nnz, int64_t heads, int64_t feat, int64_t out_rows, int dtype);

/* Zero the control words of a freshly allocated workspace (asynchronous on `stream`). */
int geot_workspace_init(void *workspace, size_t workspace_bytes, void *stream);

/* dst[index[e], :] += src[e, :]        src [nnz, feat], dst [out_rows, feat]
 * sorted != 0: index ascending; atomic-free, deterministic, dst written exactly once.
 * sorted == 0: any order; runs of equal keys are pre-reduced, then float atomics
 *              (dst is zero-filled by the call). */
int geot_index_scatter(const int64_t *index, const void *src, void *dst, int64_t nnz,
                       int64_t feat, int64_t out_rows, int dtype, int sorted, void *workspace,
                       size_t workspace_bytes, void *stream);

/* Sorted index_scatter with any reduction of the reference's CPU path
 * (csrc/cpu/index_scatter_cpu.cpp:124-134; the reference's GPU kernels only ever add).
 * reduce: GEOT_REDUCE_* (same order as csrc/reducetype.h:3).  mean divides by the edge count;
 * min/max propagate NaN like ATen; rows without edges come out 0 for every reduction. */
enum { GEOT_REDUCE_MAX = 0, GEOT_REDUCE_MEAN = 1, GEOT_REDUCE_MIN = 2, GEOT_REDUCE_SUM = 3, GEOT_REDUCE_PROD = 4 };
int geot_index_scatter_reduce(const int64_t *index, const void *src, void *dst, int64_t nnz,
                              int64_t feat, int64_t out_rows, int dtype, int reduce,
                              void *workspace, size_t workspace_bytes, void *stream);

/* dst[dst_index[e], :] = reduce over e of (weight[e] *) src[src_index[e], :]; weight may be NULL.
 * The message-passing aggregations of PyG call sites (the reference's models forward their `aggr` as the
 * trailing `reduce`, models/conv/spmm.py:5-14): sum / mean / min / max / prod over the messages of a row. */
int geot_gather_reduce(const int64_t *src_index, const int64_t *dst_index, const void *weight,
                       const void *src, void *dst, int64_t nnz, int64_t feat, int64_t src_rows,
                       int64_t out_rows, int dtype, int reduce, void *workspace,
                       size_t workspace_bytes, void *stream);

/* dst[dst_index[e], :] += src[src_index[e], :]     dst_index ascending */
int geot_gather_scatter(const int64_t *src_index, const int64_t *dst_index, const void *src,
                        void *dst, int64_t nnz, int64_t feat, int64_t src_rows,
                        int64_t out_rows, int dtype, void *workspace, size_t workspace_bytes,
                        void *stream);

/* dst[dst_index[e], :] += weight[e] * src[src_index[e], :]     dst_index ascending */
int geot_gather_weight_scatter(const int64_t *src_index, const int64_t *dst_index,
                               const void *weight, const void *src, void *dst, int64_t nnz,
                               int64_t feat, int64_t src_rows, int64_t out_rows, int dtype,
                               void *workspace, size_t workspace_bytes, void *stream);

/* dst[dst_index[e], h, :] += w(e, h) * src[src_index[e], h, :]   src [src_rows, heads, feat]
 * w(e,h) = weight[e*heads + h] (GEOT_W_EDGE_MAJOR) or weight[h*nnz + e] (GEOT_W_HEAD_MAJOR) */
int geot_mh_spmm(const int64_t *src_index, const int64_t *dst_index, const void *weight,
                 const void *src, void *dst, int64_t nnz, int64_t heads, int64_t feat,
                 int64_t src_rows, int64_t out_rows, int weight_layout, int dtype,
                 void *workspace, size_t workspace_bytes, void *stream);

/* out[e] = < mat_1[dst_index[e], :], mat_2[src_index[e], :] >      (float32) */
int geot_sddmm_coo(const int64_t *src_index, const int64_t *dst_index, const void *mat_1,
                   const void *mat_2, void *out, int64_t nnz, int64_t feat, int64_t rows_1,
                   int64_t rows_2, int dtype, void *stream);

/* Multi-head SDDMM: out(e, h) = < mat_1[dst_index[e], h, :], mat_2[src_index[e], h, :] >, mat_* [rows, heads, feat];
 * out laid out like geot_mh_spmm's weight: out[e*heads + h] (GEOT_W_EDGE_MAJOR) or out[h*nnz + e] (GEOT_W_HEAD_MAJOR).
 * d/dweight of geot_mh_spmm (mat_1 = the output's gradient, mat_2 = src).  The reference has no counterpart: its mh_spmm has
 * no backward (geot/mh_spmm.py:4-12); the pattern is sddmm_coo_cuda for the single-head op (geot/gather_weight_scatter.py:8-12,46-50). */
int geot_mh_sddmm_coo(const int64_t *src_index, const int64_t *dst_index, const void *mat_1, const void *mat_2, void *out, int64_t nnz,
                      int64_t heads, int64_t feat, int64_t rows_1, int64_t rows_2, int weight_layout, int dtype, void *stream);

/* Backward of the max / min aggregation of the gather ops (geot_gather_reduce with GEOT_REDUCE_MAX / MIN): the gradient of
 * out[d, f] goes to the messages (weight[e] *) src[src_index[e], f] that ATTAIN out[d, f], divided evenly among ties
 * (torch.scatter_reduce's rule).  grad_src [src_rows, feat] is written in full; grad_weight [nnz] may be NULL; `ties` is
 * out_rows x feat elements of scratch.  The sums into grad_src are float atomics (order not fixed).  float32 / float64.
 * No counterpart in the reference (its GPU kernels ignore `reduce`, its wrappers differentiate the sum only). */
int geot_gather_select_backward(const int64_t *src_index, const int64_t *dst_index, const void *weight, const void *src, const void *out,
                                const void *grad, void *ties, void *grad_src, void *grad_weight, int64_t nnz, int64_t feat, int64_t src_rows,
                                int64_t out_rows, int dtype, void *stream);

/* dst[e, :] = src[index[e], :] */
int geot_gather_rows(const int64_t *index, const void *src, void *dst, int64_t nnz,
                     int64_t feat, int64_t src_rows, int dtype, void *stream);

/* Row rule and precondition in one pass (device int64 out2[2], written asynchronously on `stream`):
 *   out2[0] = index[nnz-1]        <- the index[-1].item() read of csrc/index_scatter.cpp:30 and
 *                                    csrc/gather_scatter.cpp:27 (output rows = out2[0] + 1)
 *   out2[1] = number of positions with index[i] > index[i+1]  (0 <=> ascending)
 * The reference's own test and benchmark pass sorted=False with a sorted index
 * (test/test_index_scatter.py:14, benchmark/bench_index_scatter.py:32); a host layer that reads out2
 * anyway for the row count can route such calls to the atomic-free kernels. */
int geot_index_probe(const int64_t *index, int64_t nnz, int64_t *out2, void *stream);

/* Row-rule read-back without a copy engine round trip (launch-bound calls).  One-shot request of the calling thread: the
 * FIRST kernel of the next geot_index_scatter* / geot_gather_* / geot_mh_spmm call this thread makes copies *device_word
 * (normally &index[nnz-1]) to host_slot2[0] and then stores `seq` to host_slot2[1], while it runs.  host_slot2 is
 * fine-grained pinned host memory (hipHostMalloc) the device can write; the host spins on host_slot2[1] == seq instead of
 * hipMemcpyAsync + hipEventSynchronize.  geot_publish_pending() returns 1 (and disarms) if the request was NOT taken by
 * that call (a path whose first kernel does not publish, an empty call): the host then reads the word back the usual way. */
int geot_publish_word(const int64_t *device_word, int64_t *host_slot2, int64_t seq);
int geot_publish_pending(void);

/* Descent guard of the SORTED calls.  `sorted != 0` / an ascending dst_index is a promise the reference never has to
 * check: its "sorted" kernels flush every run with atomicAdd into a zeroed dst (csrc/cuda/index_scatter_kernel.cuh:180,197)
 * and still add up when the promise is wrong.  The atomic-free kernels here need it, so they verify it as they go (one
 * wave ballot per 64 keys while the keys are staged) and, when an index turns out to have descents, the call REPAIRS
 * ITSELF on the device before it completes: dst is zero-filled and every edge is added with float atomics - the
 * reference's own formulation (slow, correct; sum over fp32 / fp64).  Other reductions and the 16-bit dtypes have no
 * float atomic: their dst is filled with NaN instead.  Nothing is ever silently wrong, and nothing is needed from the
 * caller.  A host layer that caches facts about index tensors can ask to be told: host_slot2 = two int64 in
 * fine-grained pinned host memory (hipHostMalloc), sticky for the calling thread (NULL = off); a repaired call stores 1
 * to host_slot2[0], a NaN-filled one to host_slot2[1] (while it runs; the host clears the words when it has seen them). */
int geot_set_alarm_word(int64_t *host_slot2);

/* Content fingerprint: the guard of what a host layer derives from index tensors and keeps (slab plans, sorted edge
 * lists ...).  The reference keeps nothing between calls (csrc/gather_scatter.cpp:25-34 reads the caller's tensors every
 * time), so a host layer that does must notice bytes that changed behind the tensors' version counters.  128-bit,
 * position-sensitive, not cryptographic.  bufs[0..nbufs) (1..4 device buffers, 2-byte aligned, whole 2-byte words) are read
 * once, streamed.  compare == 0: the fingerprint is STORED to fp[0..1] (device).  compare != 0: it is compared with
 * fp[0..1] on the device.  verdict2 (two int64 of fine-grained pinned host memory, or NULL): [0] = 1 (stored / equal) or
 * 2 (different), then [1] = seq, written by the kernel when it is done - the host polls [1].  scratch:
 * geot_content_fingerprint_scratch_bytes() of device memory, zero on entry, first word left zero (one stream at a time).
 * Asynchronous on `stream`. */
size_t geot_content_fingerprint_scratch_bytes(void);
int geot_content_fingerprint(const void *const *bufs, const size_t *bytes, int nbufs, unsigned long long *fp, int compare,
                             int64_t *verdict2, int64_t seq, void *scratch, void *stream);

/* The same pass with the key range: out4 = {index[nnz-1], descents, min(index), max(index)}.  The range sizes the sort
 * of an index with descents (below) and tells an index with negative keys apart. */
int geot_index_probe_range(const int64_t *index, int64_t nnz, int64_t *out4, void *stream);

/* Stable sort of an index by key - what replaces scatter_reduce_kernel's one atomicAdd per edge
 * (csrc/cuda/index_scatter_kernel.cuh:204-263; dispatch csrc/cuda/index_scatter_cuda.cu:28-63) for an index with
 * descents: the host layer reduces over (keys_out, perm_out) with geot_gather_reduce (deterministic, any reduction).
 *   keys_out[i]  ascending, perm_out[i] = position of that key in `index`, equal keys in their original order.
 * Keys must lie in [0, key_max], key_max < 2^32, nnz < 2^32 (geot_sort_supported; otherwise the host layer falls back to
 * a generic 64-bit sort).  Radix passes cover only bit_width(key_max) bits of 32-bit (key, position) pairs.
 * Workspace: geot_sort_workspace_bytes(nnz) bytes (0 = the query failed), 256-B aligned, no initialisation needed. */
int geot_sort_supported(int64_t nnz, int64_t key_min, int64_t key_max);
size_t geot_sort_workspace_bytes(int64_t nnz);
int geot_sort_index(const int64_t *index, int64_t nnz, int64_t key_max, int64_t *keys_out, int64_t *perm_out, void *ws,
                    size_t ws_bytes, void *stream);

/* ---- CSR path ("next" row f2 of SURVEY.md section 8) ---------------------------------------
 * geot_csr_gws  <- csr_gws_cuda   csrc/cuda/header_cuda.h:19-21 (impl csrc/cuda/csr_gws_cuda.cu,
 *                                 kernel csrc/cuda/csr_gws_kernel.cuh:12-186)
 * dst[r, :] = sum over e in [indptr[r], indptr[r+1]) of weight[e] * src[indices[e], :]
 * indptr has nrow+1 entries; weight may be NULL (all ones).  out_rows >= nrow (the reference shim
 * allocates indptr.size(0) = nrow+1 rows, csrc/csr_gws.cpp:29-31; rows >= nrow come out 0).
 * Workspace: geot_csr_workspace_bytes (the per-edge row ids are expanded into it). */
size_t geot_csr_workspace_bytes(int64_t nnz, int64_t feat, int64_t out_rows, int dtype);
int geot_csr_gws(const int64_t *indptr, const int64_t *indices, const void *weight, const void *src,
                 void *dst, int64_t nrow, int64_t nnz, int64_t feat, int64_t src_rows,
                 int64_t out_rows, int dtype, void *workspace, size_t workspace_bytes, void *stream);

/* geot_coo_to_csr <- geot::coo_to_csr  geot/match_replace/format_transform.py:5-18 (+ the Triton
 * histogram geot/triton/coo_to_csr.py:14-26).  rowptr is int32[nrow+1] like the reference's.
 * assume_sorted != 0: coo_row ascending -> rowptr is final on return (atomic-free);
 * assume_sorted == 0: any order -> rowptr[0] = 0, rowptr[1+r] = count of r (int32 atomics); the
 *                     caller finishes with an inclusive prefix sum over rowptr. */
int geot_coo_to_csr(const int64_t *coo_row, int64_t nnz, int64_t nrow, int32_t *rowptr,
                    int assume_sorted, void *stream);

/* ---- source-blocked form of the gather ops for dense graphs (csrc/seg_slab.hip) ----------------------------------
 * Replaces the same reference kernels as geot_mh_spmm / geot_gather_weight_scatter / geot_gather_scatter
 * (csrc/cuda/mh_spmm_kernel.cuh:28-111, gather_weight_scatter_kernel.cuh:118-185, gather_scatter_kernel.cuh:118-186)
 * when a graph is dense enough for its source rows to be re-used out of an XCD's L2: Reddit scale runs ~2x
 * faster than the per-edge gather.  The edge list is pre-arranged ONCE (Phase A: geot_slab_plan_rows / _groups / _edges
 * below, driven by the host layer and kept per edge list): all arrays below are DEVICE memory in processing order.
 *   units            streams of the persistent grid (geot_slab_units_for): its waves (geot_slab_units()) where a row is a whole
 *                    wave-instruction, waves x 1024 / rowbytes lane groups otherwise
 *   groups           <= rows_per_group consecutive (virtual) dst rows with about equal edge counts, ordered by
 *                    size (descending); group at position p is run by unit (p % units) in round (p / units)
 *   e_src/e_dl/e_perm  per edge, grouped by position and sorted by (source slab, row in group): source row,
 *                    row inside the group, original edge number (for the weights)
 *   g_begin[p]       first edge of the group at position p (n_groups + 1 entries); g_vrow0 / g_nv its rows
 *   v_out[v]         >= 0: the dst row of virtual row v;  < 0: -(carry slot + 1) for a piece of a split hub row
 *   c_row/c_first/c_count  the split rows: dst row, first carry slot, number of slots
 * A row above `cap` edges is split by INTERLEAVING (edge j of the row belongs to piece j mod nv): every piece samples the row's
 * whole source range, also on an edge list whose sources ascend inside every row (a CSR, coalesce(), any transposed list) -
 * contiguous pieces there put a hub's groups into a few steps of the sweep and the lockstep waits for them (2.4x at configs[3]).
 * A plan made by other means must do the same.                                                                */
typedef struct geot_slab_plan {
  const int32_t *e_src;
  const uint8_t *e_dl;
  const int32_t *e_perm;
  const int64_t *g_begin;
  const int32_t *g_vrow0;
  const int32_t *g_nv;
  const int64_t *v_out;
  const int64_t *c_row;
  const int64_t *c_first;
  const int32_t *c_count;
  const int32_t *v_row;   /* per virtual row: its dst row (geot_slab_sddmm); may be NULL for geot_slab_spmm */
  const int32_t *v_total; /* per virtual row: edges of its WHOLE dst row (mean); may be NULL for sum / max / min */
  const int64_t *c_total; /* per split row: its edge count (mean) */
  int64_t n_groups, n_vrows, n_carry, n_split, nnz;
  int32_t units, rows_per_group;
  int32_t slab_shift, n_slabs; /* the plan's slabs: source row >> slab_shift, n_slabs of them (0 / 0: unknown -> no pacing) */
} geot_slab_plan;

int geot_slab_units(void);                                     /* waves of the persistent grid: CUs of this device x workgroups per CU x 4 */
int geot_slab_full_chip(void);                                 /* 1: all 256 CUs / 160 KB LDS (what the density rule was measured on) */
int geot_slab_rows_per_group(int weight_mode, int64_t heads);  /* R that fits the LDS budget (float32 storage) */
int geot_slab_rows_per_group_dtype(int weight_mode, int64_t heads, int dtype); /* ... 16-bit storage: fp32 accumulators, half the rows */
/* What a plan's `units` and rows per group must be for the kernel that will run it.  A unit is the rowbytes / 16 lanes of a row
 * (units = waves x 1024 / rowbytes) - except under multi-head weights (weight_mode 2 / 3 / 5) on rows of 512 / 256 bytes, which run
 * one row per wave-instruction (8 / 4 bytes per lane): there a unit is a wave and a group holds more rows.  geot_slab_spmm /
 * geot_slab_sddmm read the form off the plan's `units` (a plan cut into waves runs row-per-wave under any weight mode; measured
 * slower than lane groups except under multi-head weights: profiles/r05/slab_cases__row_per_wave_*).
 * Round 6: 16-bit multi-head plans over rows of 1 KiB get 16 rows per group (the matrix-core kernels run them in two passes; the
 * vector-ALU kernels cannot: a non-finite source table is served by a slow stand-in, geot_slab_sddmm refuses the unstaged form).
 * A 16-bit plan cut into waves with at most 16 rows per group (geot_slab_rows_per_group_shape caps 16-bit wave-cut plans
 * there) is run on the matrix cores - SUMS (one head: MEANS too) by geot_slab_spmm under any weight mode, geot_slab_sddmm / geot_slab_mh_sddmm with their
 * results staged - and there it is FASTER than lane groups for one weight per edge or none over 256- / 512-byte rows too (bf16 at
 * Reddit scale: F = 128 1.78 against 2.30 ms, F = 256 3.26 against 4.03, profiles/r06/slab_cases__rows_of_*): ask for weight_mode 2's units and rows for
 * such a plan (the host layer and geot_amd.Graph do; max / min are faster on lane groups). */
int geot_slab_units_for(int weight_mode, int64_t rowbytes);
int geot_slab_rows_per_group_shape(int weight_mode, int64_t heads, int dtype, int64_t rowbytes);
size_t geot_slab_workspace_bytes(const geot_slab_plan *plan, int64_t feat_total);
/* ... plus room for the call's weights in plan order (nnz x heads elements: weight mode 1, and weight mode 2 with four or eight
 * 16-bit heads - 8 / 16 bytes an edge; otherwise the same number).  Given that much, geot_slab_spmm brings EDGE-order weights into plan order by a
 * pre-pass of its own ahead of the persistent kernel instead of reading each one through the edge permutation in the row loop (gws
 * F=128 fp32 at Reddit scale 4.50 -> 3.98 ms, profiles/r05/slab_cases_*; bf16 H=4 x F=64, a group at a time through LDS, 4.60 ->
 * 4.25 ms, profiles/r06/slab_cases__mh_weights_staged_a_group_at_a_time.txt; four fp32 heads: measured, no gain, not staged). */
size_t geot_slab_workspace_bytes_staged(const geot_slab_plan *plan, int64_t feat_total, int weight_mode, int64_t heads, int dtype);
/* dst[d, h, :] = reduce_e w(e, h) * src[s[e], h, :] over the plan's edges.  weight_mode: 0 none (gather_scatter),
 * 1 weight[e] (gather_weight_scatter, heads = 1), 2 weight[e*heads + h], 3 weight[h*nnz + e] (mh_spmm layouts),
 * 4 = 1 with weight[] already permuted into the plan's edge order (weight[i] belongs to e_perm[i]: a static weight,
 * e.g. a normalised adjacency, permuted once by the caller), 5 = 2 likewise (weight[i*heads + h] belongs to e_perm[i]: attention
 * coefficients that were COMPUTED in plan order - geot_slab_mh_sddmm with out = NULL - never pass through the edge permutation).
 * reduce: GEOT_REDUCE_SUM | MEAN | MAX | MIN over the messages of a row (weight modes 0 / 1; the multi-head modes sum) -
 * the aggregations PyG call sites forward (GraphSAGE mean / max on Reddit-like graphs).
 * float32 / float16 / bfloat16 storage (16-bit: fp32 accumulation, one rounding at the end, weights in the storage type),
 * rows (heads * feat * element size) of 128 / 256 / 512 / 1024 bytes; src_rows * row bytes <= 4 GiB (the kernel addresses a row as a
 * 32-bit offset from the table's base; GEOT_EUNSUPPORTED beyond - such a table is no candidate for a slab sweep anyway).
 * dst is written in full. */
int geot_slab_spmm(const geot_slab_plan *plan, const void *weight, int weight_mode, const void *src, void *dst,
                   int64_t heads, int64_t feat, int64_t src_rows, int64_t out_rows, int dtype, int reduce,
                   void *workspace, size_t workspace_bytes, void *stream);

/* Phase A on the device (csrc/seg_plan.hip): builds the arrays of a geot_slab_plan from the dst-sorted COO edge list in
 * three stages.  The caller owns every buffer; the two read-backs of the whole procedure are the sizes it needs to
 * allocate them (stages 1 and 2 synchronise `stream` for that; stage 3 is asynchronous).  The reference has no
 * counterpart (its kernels gather per edge); its one per-graph pass is the row-pointer histogram of
 * geot/match_replace/format_transform.py:5-25.
 *   job      in: src_index / dst_index (int64, dst ascending, keys in [0, out_rows)), nnz < 2^31, out_rows, src_rows,
 *            rowbytes (of a source row), slab_bytes (2 MiB: geot's measured choice), units (geot_slab_units() *
 *            (1024 / rowbytes)), rows_per_group (geot_slab_rows_per_group).
 *   stage 1  geot_slab_plan_rows: fills nonempty / budget / cap / n_vrows / n_split / n_carry / slab_shift / n_slabs.
 *            scratch1: geot_slab_plan_scratch_bytes(job, 1) bytes, kept until stage 3 has run.
 *   stage 2  geot_slab_plan_groups: writes v_out / v_row / v_total [n_vrows] and c_row / c_first / c_count / c_total
 *            [n_split, at least 1 entry each], fills n_groups.  scratch2: ..._scratch_bytes(job, 2), kept likewise.
 *   stage 3  geot_slab_plan_edges: writes g_begin [n_groups + 1], g_vrow0 / g_nv [n_groups], e_src / e_dl / e_perm [nnz].
 *            scratch3: ..._scratch_bytes(job, 3) (16 B per edge + the sort's buffer), free once the stream has passed it.
 * GEOT_EUNSUPPORTED: keys outside [0, out_rows), or groups x slabs x rows_per_group >= 2^32 (the host layer then keeps
 * the per-edge kernels).  All scratch 256-byte aligned, no initialisation needed. */
typedef struct geot_slab_plan_job {
  const int64_t *src_index, *dst_index;
  int64_t nnz, out_rows, src_rows, rowbytes, slab_bytes, units;
  int32_t rows_per_group;
  int64_t nonempty, budget, cap, n_vrows, n_split, n_carry; /* stage 1 */
  int64_t n_groups;                                        /* stage 2 */
  int32_t slab_shift, n_slabs;                             /* stage 1 */
} geot_slab_plan_job;
size_t geot_slab_plan_scratch_bytes(const geot_slab_plan_job *job, int stage);
int geot_slab_plan_rows(geot_slab_plan_job *job, void *scratch1, size_t scratch1_bytes, void *stream);
int geot_slab_plan_groups(geot_slab_plan_job *job, const void *scratch1, void *scratch2, size_t scratch2_bytes, int64_t *v_out,
                          int32_t *v_row, int32_t *v_total, int64_t *c_row, int64_t *c_first, int32_t *c_count, int64_t *c_total,
                          void *stream);
int geot_slab_plan_edges(const geot_slab_plan_job *job, const void *scratch1, const void *scratch2, void *scratch3,
                         size_t scratch3_bytes, int64_t *g_begin, int32_t *g_vrow0, int32_t *g_nv, int32_t *e_src, uint8_t *e_dl,
                         int32_t *e_perm, void *stream);

/* out[e] = < mat_1[dst(e), :], mat_2[src(e), :] > in ORIGINAL edge order over the plan's edges - geot_sddmm_coo
 * (sddmm_coo_cuda, csrc/cuda/header_cuda.h:28-30) for a graph that has a plan: the backward (d/dweight) of
 * gather_weight_scatter on a dense graph.  float32 / float16 / bfloat16 (fp32 dot products, out in the storage type), rows
 * of 256 / 512 / 1024 bytes, rows_2 * row bytes <= 4 GiB (as geot_slab_spmm); workspace as geot_slab_spmm. */
int geot_slab_sddmm(const geot_slab_plan *plan, const void *mat_1, const void *mat_2, void *out, int64_t feat,
                    int64_t rows_1, int64_t rows_2, int dtype, void *workspace, size_t workspace_bytes, void *stream);
/* The same result, faster: the persistent kernel writes its dot products in the PLAN's edge order into `staging` (plan->nnz elements
 * of the storage type, the caller's scratch; 8 consecutive results per 32-byte piece) and a second kernel brings them into original
 * edge order, a group at a time through LDS (a group's edges are a contiguous range of the dst-sorted list unless it holds a piece
 * of a split hub).  Written straight to out[original edge id], every 4-byte result is a partial write of its own: 3.6 GB written
 * for 0.46 GB of results at 115 M edges (F=128 fp32: 5.41 ms direct, 4.52 ms staged). */
int geot_slab_sddmm_staged(const geot_slab_plan *plan, const void *mat_1, const void *mat_2, void *out, void *staging, int64_t feat,
                           int64_t rows_1, int64_t rows_2, int dtype, void *workspace, size_t workspace_bytes, void *stream);

/* out[i, :] = weight[e_perm[i], :]: per-edge values (heads per edge) from the caller's edge order into the plan's - what weight modes
 * 4 / 5 read (a static weight permuted once).  heads x element size of 2 / 4 / 8 / 16 bytes; GEOT_EUNSUPPORTED otherwise. */
int geot_slab_to_plan_order(const geot_slab_plan *plan, const void *weight, void *out, int64_t heads, int dtype, void *stream);

/* Multi-head SDDMM over the plan (d/dweight of geot_mh_spmm on a dense graph; the scores of an attention layer):
 * out(e, h) = < mat_1[dst(e), h, :], mat_2[src(e), h, :] >, mat_* [rows, heads, feat], heads 1 / 2 / 4 / 8, rows of 256 / 512 / 1024
 * bytes.  `staging`: plan->nnz x heads elements of scratch.  out != NULL: results in ORIGINAL edge order, out[e*heads + h] (heads x
 * element size of 2 / 4 / 8 / 16 bytes).  out == NULL: the results STAY in `staging`, in the plan's edge order - what geot_slab_spmm
 * reads back under weight_mode 5 (heads = 1: weight_mode 4) with no permutation in between.  The reference has no counterpart
 * (geot/mh_spmm.py:4-12: no backward; its attention layers compute scores with torch ops, models/conv/gatconv.py). */
int geot_slab_mh_sddmm(const geot_slab_plan *plan, const void *mat_1, const void *mat_2, void *out, void *staging, int64_t heads, int64_t feat,
                       int64_t rows_1, int64_t rows_2, int dtype, void *workspace, size_t workspace_bytes, void *stream);

/* THE DESCENT CONTRACT of every sorted entry point (geot_index_scatter(sorted = 1), geot_index_scatter_reduce, geot_gather_*,
 * geot_mh_spmm, geot_csr_gws): the kernels verify "dst_index ascending" while they stage the keys (keys outside
 * [0, out_rows) are ignored and never count).  A call that meets a DESCENT repairs itself on the device - sum over
 * float32 / float64: dst is zero-filled and every edge added with float atomics (the reference's own scheme,
 * csrc/cuda/index_scatter_kernel.cuh:180,197), correct but slow and not deterministic; any other reduction or 16-bit storage
 * has no float atomic to fall back on: dst is FILLED WITH NaN.  Either way the thread's alarm word (geot_set_alarm_word) is
 * raised: word[0] = repaired, word[1] = NaN-filled.  A caller that set no alarm word hears nothing - it must not pass
 * sorted = 1 on faith for reductions other than sum (probe first: geot_index_probe).  The reference tolerates a wrong
 * `sorted` promise for sum only (its kernels ignore `reduce` on the GPU): this is a documented difference. */

#ifdef __cplusplus
}
#endif
#endif /* GEOT_HIP_H */
