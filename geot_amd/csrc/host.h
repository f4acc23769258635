// host.h -- what the translation units of the dispatcher plugin (geot_amd/_C.so) share.
//
//   host_state.cpp   options, statistics, the cache mutex, small helpers, the per-(device, stream) workspace, the pinned slot
//   host_cache.cpp   what the host layer REMEMBERS about the caller's index tensors and how it stays true: index facts, stable
//                    sorts, widened int32 indices, expanded CSR row ids, transposed edge lists; the content guard; the byte budget
//   host_plan.cpp    dense graphs: Phase A (the source-blocked plan), its routing rule and cache
//   torch_ops.cpp    the geot::* operators (the reference's checks, texts and row rule) and their registrations
//
// The reference's host side keeps nothing between calls (csrc/gather_scatter.cpp:25-34); everything here beyond forwarding
// pointers is described in DESIGN.md sections 3.1c / 3.1d / 5.
#ifndef GEOT_HOST_H
#define GEOT_HOST_H
#include <ATen/ATen.h>
#include <ATen/OpMathType.h>
#include <ATen/Parallel.h>
#include <ATen/hip/impl/HIPCachingAllocatorMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <hip/hip_runtime_api.h>
#include <torch/library.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <list>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "geot_hip.h"

namespace geot_host {

// ---- options (environment at load time, geot::_host_option at run time) -------------------------------------------
struct Options {
  int speculate_rows = 1;    // GEOT_SPECULATE_ROWS
  int trust_version = 1;     // GEOT_TRUST_VERSION: 0 probe every call, 1 facts per content + row count read back every call,
                             //   2 the remembered row count is trusted too (no read-back, no host wait: lowest latency)
  int unsorted_mode = 0;     // GEOT_UNSORTED: 0 auto (sort), 1 sort, 2 atomic
  int slab_mode = 0;         // GEOT_SLAB: -1 never, 0 auto, 1 always
  int transpose_cache = 4;   // GEOT_TRANSPOSE_CACHE (entries)
  int slab_keep = 2;
  int publish_rows = 1;      // GEOT_PUBLISH_ROWS: small calls get index[-1] from their own first kernel (geot_publish_word)
  int64_t cache_mb = 0;      // GEOT_CACHE_MB: byte budget of all cached artefacts together (0 = 1/8 of the device's memory)
  int slab_builder = 0;      // Phase A: 0 = the device builder (csrc/seg_plan.hip), 1 = the ATen formulation (CPU tensors always; cross-check)
  int content_guard = 1;     // GEOT_CONTENT_GUARD: every use of a remembered product re-reads the tensors it was derived from (fingerprint)
  int slab_bytes = 0;        // slab size of new plans in bytes (0 = by the rule, host_plan.cpp slab_bytes_rule)
  int slab_min_coverage_pct = 50; // graphs whose groups touch less than this share of the source slabs keep the per-edge kernels (slab_source_coverage; 0: no probe)
  Options() {
    if (const char *e = std::getenv("GEOT_PUBLISH_ROWS")) publish_rows = std::strcmp(e, "0") != 0;
    if (const char *e = std::getenv("GEOT_SPECULATE_ROWS")) speculate_rows = std::strcmp(e, "0") != 0;
    if (const char *e = std::getenv("GEOT_TRUST_VERSION")) trust_version = !std::strcmp(e, "0") ? 0 : (!std::strcmp(e, "2") ? 2 : 1);
    if (const char *e = std::getenv("GEOT_UNSORTED")) unsorted_mode = !std::strcmp(e, "atomic") ? 2 : (!std::strcmp(e, "sort") ? 1 : 0);
    if (const char *e = std::getenv("GEOT_SLAB")) slab_mode = !std::strcmp(e, "0") ? -1 : (!std::strcmp(e, "1") ? 1 : 0);
    if (const char *e = std::getenv("GEOT_TRANSPOSE_CACHE")) transpose_cache = std::atoi(e);
    if (const char *e = std::getenv("GEOT_CACHE_MB")) cache_mb = std::atoll(e);
    if (const char *e = std::getenv("GEOT_CONTENT_GUARD")) content_guard = std::strcmp(e, "0") != 0;
  }
};
extern Options g_opt;
struct Stats {
  int64_t probes = 0, row_mismatches = 0, sorts = 0, transposes = 0, plans_built = 0, slab_calls = 0, plan_us = 0, published = 0, alarms = 0, stale_products = 0, guard_checks = 0, plan_trials = 0, plans_rejected = 0, trial_plan_us = 0, trial_edges_us = 0, plans_declined = 0, last_coverage_permille = 0;
};
extern Stats g_stats;
extern std::mutex g_mu; // guards the caches below (facts, transposed edge lists, slab plans)


// ---- small helpers (host_state.cpp) ----------------------------------------------------------------------------------------
int dtype_code(const at::Tensor &t, const char *op);
int reduce_code(c10::string_view reduce, bool pyg_add = false); // csrc/reduceutils.h:5-22 (+ PyG's 'add' for the gather ops)
void require_gpu(const char *op, std::initializer_list<const at::Tensor *> ts);
// (a ROCm build of PyTorch calls the GPU "cuda": the guard / stream types that accept that device type)
void *stream_of(const at::Tensor &t);

// ---- hipGraph capture --------------------------------------------------------------------------------------------------
// Under stream capture (torch.cuda.graph around a model) nothing may synchronise and nothing enqueued has run yet:
// an op whose index facts are already known launches with the remembered row count and skips the read-back; nothing
// produced during the capture enters a cache (its kernels have only been recorded); a plan that is not there is not
// built.  An index that has never been seen cannot be probed: the call fails with a clear message (run it once first).
extern thread_local bool tl_capturing;
struct CaptureScope {
  bool prev;
  explicit CaptureScope(const at::Tensor &t) : prev(tl_capturing) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    tl_capturing = t.is_cuda() && hipStreamIsCapturing(static_cast<hipStream_t>(stream_of(t)), &st) == hipSuccess &&
                   st != hipStreamCaptureStatusNone;
  }
  ~CaptureScope() { tl_capturing = prev; }
};

#define GEOT_DEVICE_GUARD(t)                                                                                            \
  const c10::hip::HIPGuardMasqueradingAsCUDA geot_device_guard_((t).device());                                         \
  const CaptureScope geot_capture_scope_(t)
#define GEOT_CALL(expr)                                                                                                 \
  do {                                                                                                                  \
    const int rc_ = (expr);                                                                                             \
    TORCH_CHECK(rc_ == GEOT_OK, #expr, " failed (code ", rc_, "): ", geot_last_error());                                \
  } while (0)

inline const int64_t *index_ptr(const at::Tensor &t) { return t.data_ptr<int64_t>(); } // "expected scalar type Long but found ..."


// ---- cached device artefacts and streams -----------------------------------------------------------------------------------
// What the caches below keep (the sort of an index, a widened index, a plan, a transposed edge list) was enqueued on the
// stream that was current when it was made.  A later call on ANOTHER stream must not read it before that work is done,
// and the caching allocator must not recycle its memory for the producing stream while the consumer still reads it:
// the entry keeps an event of its production; a consumer on a different stream waits for it and records itself.
struct Produced {
  std::shared_ptr<void> ev;
  void *stream = nullptr;
  void mark(const at::Tensor &on) {
    if (tl_capturing) return; // (nothing made during a capture is cached; see CaptureScope)
    hipEvent_t e = nullptr;
    TORCH_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
    ev = std::shared_ptr<void>(e, [](void *p) { (void)hipEventDestroy(static_cast<hipEvent_t>(p)); });
    stream = stream_of(on);
    TORCH_CHECK(hipEventRecord(e, static_cast<hipStream_t>(stream)) == hipSuccess, "hipEventRecord failed");
  }
  template <typename Each> void consume(const at::Tensor &on, Each each) const {
    if (!ev) return;
    const auto cur = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(on.device().index());
    if (cur.stream() == stream) return;
    if (tl_capturing) {
      // a capturing stream must not wait for an event recorded outside the capture (capture isolation).  After torch's
      // usual recipe - warm up, synchronize, capture on a fresh stream - the artefact is long done: nothing to wait for.
      const hipError_t q = hipEventQuery(static_cast<hipEvent_t>(ev.get()));
      if (q != hipSuccess) (void)hipGetLastError();
      TORCH_CHECK(q == hipSuccess, "geot: a cached artefact of this call (index sort / plan / transposed edges) is still being produced on "
                  "another stream and cannot be waited for inside a graph capture.  Run the call once and synchronize before capturing.");
    } else {
      TORCH_CHECK(hipStreamWaitEvent(cur.stream(), static_cast<hipEvent_t>(ev.get()), 0) == hipSuccess, "hipStreamWaitEvent failed");
    }
    each([&](const at::Tensor &t) {
      if (t.defined() && t.is_cuda()) c10::hip::HIPCachingAllocatorMasqueradingAsCUDA::recordStreamMasqueradingAsCUDA(t.storage().data_ptr(), cur);
    });
  }
  void before_use(const at::Tensor &on, std::initializer_list<const at::Tensor *> ts) const {
    consume(on, [&](auto rec) { for (const at::Tensor *t : ts) rec(*t); });
  }
  void before_use(const at::Tensor &on, const std::vector<at::Tensor> &ts) const {
    consume(on, [&](auto rec) { for (const at::Tensor &t : ts) rec(t); });
  }
};

// The caches below hold DERIVED artefacts only.  The user's tensors they were derived from are referenced weakly: a weak
// reference pins the StorageImpl object (so its address - part of the content key - cannot be handed to a new tensor while
// the entry lives: no aliasing of a dead tensor) but not its data; an entry whose source has died is dropped at the next
// lookup, and all artefacts together stay under a byte budget (largest least-recently-used entry goes first).
using WeakStorage = c10::weak_intrusive_ptr<c10::StorageImpl>;
inline WeakStorage weak_of(const at::Tensor &t) { return t.storage().getWeakStorageImpl(); }
inline int64_t nbytes_of(const at::Tensor &t) { return t.defined() ? (int64_t)t.numel() * (int64_t)t.element_size() : 0; }
void enforce_cache_budget_locked(); // (defined behind the caches; call with g_mu held)
void sweep_expired_locked();


// one zero-initialised workspace per (device, stream), grown on demand (the ABI: one stream at a time per workspace).  Under graph
// capture a workspace that is not there yet (or too small) is made for THIS call only and not kept: memory of a graph's private pool
// must not outlive the graph in a thread-local cache (exit-order crashes), and the captured memset re-zeroes it on every replay
at::Tensor workspace(const at::Tensor &like, size_t bytes);
void clear_all_caches_locked();
int64_t cache_bytes_locked();

// ---- pinned read-back slot per (thread, device) -------------------------------------------------------------------------
struct Slot {
  int64_t *host = nullptr;   // [0..3] copies (probe, row rule), [4] word published by a kernel, [5] its sequence number,
                             // [6] / [7] descent alarm of the kernels (geot_set_alarm_word): a call repaired itself / NaN-filled its output
                             // [8 + 2i], [9 + 2i], i < kGuardSlots: verdict and sequence number of a content fingerprint (guard_check)
  hipEvent_t ev = nullptr;
  int64_t seq = 0;
  int guard_next = 0;
};
constexpr int kGuardSlots = 8;
constexpr size_t kSlotWords = 8 + 2 * kGuardSlots;

Slot &slot_for(int device);

// ---- content guard of the remembered products (host_cache.cpp; csrc/seg_guard.hip) -----------------------------------------
bool guard_on();
bool guardable(std::initializer_list<const at::Tensor *> ts);
// may this call look a product of these tensors up / remember one?
bool may_remember(std::initializer_list<const at::Tensor *> ts);
// fingerprint of `ts` as they are now, for a product that is being made from them (undefined when the guard is off)
at::Tensor guard_store(std::initializer_list<const at::Tensor *> ts);
// a remembered product is about to be used: are `ts` still the bytes it was made from?  Only NOTES the question (callable with
// g_mu / a plan's wmu held); guard_flush launches the fingerprint kernels, guard_settle reads the verdicts
void guard_check(const at::Tensor &fp, std::initializer_list<const at::Tensor *> ts);
void guard_flush();
struct GuardFlush { // declare FIRST in a function that looks products up: its destructor runs after the function's lock_guards'
  ~GuardFlush();
};
// true: every product this operator call used was made from the bytes the tensors hold now
bool guard_settle();
// is `t` itself one of the remembered products (the transposed edge list handed to the backward pass)?
bool owned_product(const at::Tensor &t);

// ---- facts of an index tensor, keyed on its content identity (host_cache.cpp) ------------------------------------------------
struct ContentKey {
  const void *storage;
  int64_t offset, numel;
  uint32_t version;
  int64_t size[2], stride[2]; // two views of one storage with the same offset and numel but another shape are other contents
  int dim, dtype;
  bool operator==(const ContentKey &o) const {
    return storage == o.storage && offset == o.offset && numel == o.numel && version == o.version && dim == o.dim && dtype == o.dtype &&
           size[0] == o.size[0] && size[1] == o.size[1] && stride[0] == o.stride[0] && stride[1] == o.stride[1];
  }
};
bool content_key(const at::Tensor &t, ContentKey *k);
struct FactsView {
  int64_t rows;
  bool ascending;
  bool cached;
  int64_t kmin, kmax;
};
void probe_index(const at::Tensor &index, int64_t out4[4]); // one pass: {index[-1], descents, min, max}
FactsView index_facts(const at::Tensor &index);
void remember_rows(const at::Tensor &index, int64_t rows);
std::pair<at::Tensor, at::Tensor> stable_sort_index(const at::Tensor &index, int64_t kmin, int64_t kmax);
std::pair<at::Tensor, at::Tensor> sorted_form(const at::Tensor &index, int64_t kmin, int64_t kmax);
at::Tensor as_int64(const at::Tensor &t);
at::Tensor expand_indptr(const at::Tensor &indptr, int64_t nnz);
std::tuple<at::Tensor, at::Tensor, at::Tensor> transpose_edges_op(const at::Tensor &si, const at::Tensor &di);
at::Tensor transposed_weight_op(const at::Tensor &si, const at::Tensor &di, const at::Tensor &weight);
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> transpose_edges_weighted_op(const at::Tensor &si, const at::Tensor &di,
                                                                                       const at::Tensor &weight);
int64_t host_option_op(c10::string_view name, int64_t value);
std::vector<int64_t> host_stats_op();

// ---- dense graphs: source-blocked kernel (host_plan.cpp; csrc/seg_slab.hip) ---------------------------------------------------
struct SlabPlanHolder {
  std::vector<at::Tensor> keep; // the device arrays the struct points into
  geot_slab_plan plan;
  int64_t rounds, budget, cap, slabs, slab_rows;
  // a STATIC per-edge weight (a normalised adjacency: the same tensor content call after call) is permuted into the
  // plan's edge order on its second sighting; a weight that changes every call (attention, a trained parameter)
  // never is - it is read through the edge permutation
  std::mutex wmu;
  bool w_seen_valid = false;
  ContentKey w_seen{}, w_key{};
  c10::optional<WeakStorage> w_given; // the weight tensor w_planorder was made from (weak)
  at::Tensor w_planorder;
  Produced made, w_made; // events of Phase A / of the weight permutation (consumers on other streams wait for them)
  at::Tensor fp, w_fp;   // fingerprints of the edge list / of the weight those were made from (guard_store)
  // Is the plan FASTER than the per-edge kernels on this graph?  The density rule that routes a graph here was calibrated on
  // uniform-random sources.  A dense graph whose sources sit NEAR their destinations (nodes numbered by community) is another
  // matter: its per-edge gathers hit in L2 anyway, and the plan's chip-wide slab walk makes waves wait for slabs they do not need
  // - measured 3-20x SLOWER than the per-edge kernels (Reddit scale, sources within +-2000 rows: 48.8 vs 6.2 ms).  So the first
  // call that would use a plan runs BOTH ways, timed with events on the call's stream, and the plan is kept only if it wins
  // (per kind of operator).  0 undecided, 1 the plan, 2 the per-edge kernels (the plan's arrays are released then).
  std::atomic<int> verdict[2] = {{0}, {0}}; // [0] the forward reductions, [1] SDDMM
  float trial_ms[2][2] = {{0, 0}, {0, 0}};  // [kind][0 plan, 1 per-edge]: best of the timed repetitions
  std::mutex trial_mu;                       // one trial at a time per plan; a thread that finds it taken serves its call per edge
  std::atomic<int> uses_since_trial[2] = {{0}, {0}}, trials_done[2] = {{0}, {0}};
  void release() { // (keeps the holder as the record of the decision; lock order everywhere: g_mu, then wmu)
    std::vector<at::Tensor> gone;
    {
      std::lock_guard<std::mutex> lk(wmu);
      gone.swap(keep);
      w_planorder = at::Tensor();
      fp = w_fp = at::Tensor();
    }
    // `gone` dies here: a launch in flight on another thread holds its own references (pinned()) and has told the allocator
    // which stream reads them (launched_on), so the memory is not handed out again under a running kernel
  }
  // the arrays a launch is about to read through the raw pointers of `plan` (empty: released)
  std::vector<at::Tensor> pinned() {
    std::lock_guard<std::mutex> lk(wmu);
    return keep;
  }
  // after the launch: the arrays were allocated on the stream that built the plan; a launch on another stream is recorded with
  // the caching allocator AFTER it is enqueued (a block freed later is then only re-used behind this launch)
  void launched_on(const at::Tensor &on, const std::vector<at::Tensor> &arrays) const {
    if (tl_capturing) return;
    const auto cur = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(on.device().index());
    if (cur.stream() == made.stream) return;
    for (const at::Tensor &t : arrays)
      if (t.defined() && t.is_cuda()) c10::hip::HIPCachingAllocatorMasqueradingAsCUDA::recordStreamMasqueradingAsCUDA(t.storage().data_ptr(), cur);
  }
  int64_t bytes() {
    std::lock_guard<std::mutex> lk(wmu);
    int64_t b = nbytes_of(w_planorder);
    for (const at::Tensor &t : keep) b += nbytes_of(t);
    return b;
  }
};

struct SlabEntry {
  ContentKey k1, k2;
  int64_t rows, src_rows, rowbytes, heads;
  int wmode;
  int rpg;   // rows per group the plan was built for: what the weight mode / head count changes ...
  int64_t units; // ... and the lane-group streams it was cut for (waves for rows of >= 256 bytes): part of the lookup
  WeakStorage w1, w2; // the edge list the plan was built from (weak: the plan goes when the edge list dies)
  std::shared_ptr<SlabPlanHolder> plan;
};
extern std::list<SlabEntry> &g_slab;
extern std::list<std::pair<ContentKey, ContentKey>> g_sightings, g_declined;
bool slab_worthwhile(int64_t nnz, int64_t out_rows, int64_t src_rows, int64_t rowbytes, int dtype = GEOT_F32);
std::shared_ptr<SlabPlanHolder> slab_plan_for(const at::Tensor &si, const at::Tensor &di, int64_t rows, const at::Tensor &src, int wmode,
                                              int64_t heads, int red = GEOT_REDUCE_SUM);
// false: the plan's arrays have been released (a trial on another thread rejected it) - the caller runs the per-edge kernels
bool run_slab(SlabPlanHolder &H, const void *weight, int wmode, const at::Tensor &src, at::Tensor &out, int64_t heads, int64_t feat,
              int red = GEOT_REDUCE_SUM);
// values[index] along dim 0 (int64 index) for a per-edge tensor ([nnz] or [nnz, heads]): the library's row gather for contiguous
// floating CUDA tensors (torch's own gathers of such tensors have returned garbage at configs[3]'s size twice: index_select with an
// int32 index, [nnz, 4], round 5; advanced indexing of [115 M, 8] 16-bit rows, round 6), index_select otherwise
at::Tensor take_rows(const at::Tensor &values, const at::Tensor &index);
std::vector<at::Tensor> slab_plan_op(const at::Tensor &si, const at::Tensor &di, int64_t rows, int64_t src_rows, int64_t rowbytes,
                                     int64_t weight_mode, int64_t heads, int64_t slab_bytes, int64_t rows_per_group, int64_t units);
bool slab_worthwhile_op(int64_t nnz, int64_t rows, int64_t src_rows, int64_t rowbytes);

} // namespace geot_host
#endif
