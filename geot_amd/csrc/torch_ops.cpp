// torch_ops.cpp -- the geot::* dispatcher operators (geot_amd/_C.so), the HOST side of the drop-in.
//
// The reference's host side is a PyTorch C++ plugin: csrc/index_scatter.cpp:26-56, csrc/gather_scatter.cpp:13-34,
// csrc/gather_weight_scatter.cpp:11-49, csrc/mh_spmm.cpp:10-23, csrc/csr_gws.cpp:11-60 register the schemas in
// namespace `geot` and call the device entry points of csrc/cuda/header_cuda.h.  This file is that plugin for
// MI355X: the same schema strings, checks and error texts, over the C ABI of libgeot_hip.so (include/geot_hip.h).
// geot_amd/ops.py loads it (torch.ops.load_library, like geot/__init__.py:12-19) and only adds the fake-tensor
// rules and the autograd formulas, as the reference's Python files do.
//
// What the host layer does beyond forwarding pointers:
//   * row rule rows = index[-1] + 1 (csrc/index_scatter.cpp:30): read back on EVERY call, but the kernels are
//     launched for the row count remembered for that index tensor while the 8-byte copy is in flight, and the
//     count is verified afterwards (a mismatch relaunches); GEOT_SPECULATE_ROWS=0 restores the blocking order;
//   * facts of an index (ascending? row count, stable sort), probed once per CONTENT - storage identity, offset,
//     length, version counter, guarded by a weak reference to the storage: the atomic-free kernels are only ever
//     given an ascending index, whatever `sorted` promised; an index with descents is reduced over its sort;
//   * dense graphs are re-arranged once for the source-blocked kernel (csrc/seg_slab.hip) on their second call;
//   * one workspace per (device, stream), a device guard, the current stream.
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

namespace {

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
  int guard_side_stream = 0; // 1: the fingerprint kernels run on a side stream BESIDE the call's kernels instead of in front of them.  Measured
                             //    and rejected (profiles/r04/bench_content_guard_side_stream.txt): a bandwidth-bound read beside bandwidth-bound
                             //    kernels saves nothing (gws forward + backward, 40 M edges: +14.8 % either way), and beside the persistent
                             //    source-blocked kernel it breaks the lockstep (configs[3]: 8.0 -> 14.4 ms)
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
Options g_opt;
struct Stats {
  int64_t probes = 0, row_mismatches = 0, sorts = 0, transposes = 0, plans_built = 0, slab_calls = 0, plan_us = 0, published = 0, alarms = 0, stale_products = 0, guard_checks = 0, plan_trials = 0, plans_rejected = 0, trial_plan_us = 0, trial_edges_us = 0;
};
Stats g_stats;
std::mutex g_mu; // guards the caches below (facts, transposed edge lists, slab plans)

// ---- small helpers -------------------------------------------------------------------------------------------------------
int dtype_code(const at::Tensor &t, const char *op) {
  switch (t.scalar_type()) {
  case at::kFloat: return GEOT_F32;
  case at::kDouble: return GEOT_F64;
  case at::kHalf: return GEOT_F16;
  case at::kBFloat16: return GEOT_BF16;
  default: TORCH_CHECK(false, "\"", op, "\" not implemented for '", toString(t.scalar_type()), "'");
  }
}

int reduce_code(c10::string_view reduce, bool pyg_add = false) { // csrc/reduceutils.h:5-22 (+ PyG's 'add' for the gather ops)
  if (reduce == "max" || reduce == "amax") return GEOT_REDUCE_MAX;
  if (reduce == "mean") return GEOT_REDUCE_MEAN;
  if (reduce == "min" || reduce == "amin") return GEOT_REDUCE_MIN;
  if (reduce == "sum" || (pyg_add && reduce == "add")) return GEOT_REDUCE_SUM;
  if (reduce == "prod") return GEOT_REDUCE_PROD;
  TORCH_CHECK(false, "reduce argument must be either sum, prod, mean, amax or amin, got ", reduce);
}

void require_gpu(const char *op, std::initializer_list<const at::Tensor *> ts) {
  const at::Tensor *first = nullptr;
  for (const at::Tensor *t : ts) {
    if (!t || !t->defined()) continue;
    TORCH_CHECK(t->is_cuda(), "geot::", op, ": CPU tensors are not supported by geot_amd (MI355X-only package, no CPU "
                "fallback).  Move the tensors to the GPU.");
    if (!first) first = t;
    TORCH_CHECK(t->device() == first->device(), "all tensors must be on the same device");
  }
}

// (a ROCm build of PyTorch calls the GPU "cuda": the guard / stream types that accept that device type)
void *stream_of(const at::Tensor &t) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

// ---- hipGraph capture --------------------------------------------------------------------------------------------------
// Under stream capture (torch.cuda.graph around a model) nothing may synchronise and nothing enqueued has run yet:
// an op whose index facts are already known launches with the remembered row count and skips the read-back; nothing
// produced during the capture enters a cache (its kernels have only been recorded); a plan that is not there is not
// built.  An index that has never been seen cannot be probed: the call fails with a clear message (run it once first).
thread_local bool tl_capturing = false;
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

const int64_t *index_ptr(const at::Tensor &t) { return t.data_ptr<int64_t>(); } // "expected scalar type Long but found ..."

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
WeakStorage weak_of(const at::Tensor &t) { return t.storage().getWeakStorageImpl(); }
int64_t nbytes_of(const at::Tensor &t) { return t.defined() ? (int64_t)t.numel() * (int64_t)t.element_size() : 0; }
void enforce_cache_budget_locked(); // (defined behind the caches; call with g_mu held)
void sweep_expired_locked();

// one zero-initialised workspace per (device, stream), grown on demand (the ABI: one stream at a time per workspace)
at::Tensor &workspace(const at::Tensor &like, size_t bytes) {
  static thread_local std::map<std::pair<int, void *>, at::Tensor> ws;
  auto &w = ws[{(int)like.device().index(), stream_of(like)}];
  if (!w.defined() || (size_t)w.numel() < bytes)
    w = at::zeros({(int64_t)std::max<size_t>(bytes, 1 << 20)}, like.options().dtype(at::kByte));
  return w;
}

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
Slot &slot_for(int device) {
  static thread_local std::map<int, Slot> slots;
  Slot &s = slots[device];
  if (!s.host) {
    // fine-grained (coherent) pinned memory: a running kernel's stores become visible to the spinning host
    if (hipHostMalloc(reinterpret_cast<void **>(&s.host), kSlotWords * sizeof(int64_t), hipHostMallocCoherent) != hipSuccess) {
      (void)hipGetLastError();
      TORCH_CHECK(hipHostMalloc(reinterpret_cast<void **>(&s.host), kSlotWords * sizeof(int64_t), hipHostMallocDefault) == hipSuccess,
                  "hipHostMalloc failed");
    }
    std::memset(s.host, 0, kSlotWords * sizeof(int64_t));
    TORCH_CHECK(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
  }
  return s;
}

// ---- content guard of the remembered products (csrc/seg_guard.hip) -----------------------------------------------------------
// Everything the caches below keep was derived from the caller's index tensors and is found again by their identity and
// version counter.  A write behind the version counter (.data, DLPack, a raw pointer) leaves a product that describes bytes
// that are gone; the reference, which keeps nothing, would follow the new bytes.  So a product carries the fingerprint of
// the tensors it was made from (guard_store), every later use re-reads them and compares on the device (guard_check:
// 8-16 streamed bytes per edge, enqueued in FRONT of the call's kernels, verdict published into pinned memory), and the
// operator looks at the verdicts before it returns (guard_settle; by then the call has waited for its own read-back, or
// waits here for the fingerprint alone - the kernels behind it keep running).  One mismatch drops every remembered product
// and the operator runs once more from the caller's bytes: the result the caller gets is always the one of the tensors as
// they are.  Off under graph capture (a captured call is a contract about static content anyway), with trust_version 2
// (the caller opted out of read-backs) and with content_guard 0.
struct PendingVerdict {
  int64_t *slot; // pinned: [0] verdict, [1] sequence number
  int64_t seq;
  void *stream;
};
thread_local std::vector<PendingVerdict> tl_pending;
thread_local bool tl_guard_tripped = false;
// (fingerprint, first tensor) pairs this operator call has already asked about: transposed_weight looks the edge list up
// through transpose_edges and then again for the weight - one read of the edge list answers both
thread_local std::vector<std::pair<const void *, const void *>> tl_asked;

bool guard_on() { return g_opt.content_guard && g_opt.trust_version == 1 && !tl_capturing; }
// (the fingerprint reads one flat range per tensor)
bool guardable(std::initializer_list<const at::Tensor *> ts) {
  for (const at::Tensor *t : ts)
    if (!t->defined() || !t->is_cuda() || !t->is_contiguous() || (t->element_size() & 1)) return false;
  return true;
}
// may this call look a product of these tensors up / remember one?
bool may_remember(std::initializer_list<const at::Tensor *> ts) { return g_opt.trust_version && (!guard_on() || guardable(ts)); }

at::Tensor &guard_scratch(const at::Tensor &like, void *stream) { // per (device, stream): the kernel's ticket and per-workgroup sums
  static thread_local std::map<std::pair<int, void *>, at::Tensor> sc;
  auto &t = sc[{(int)like.device().index(), stream}];
  if (!t.defined()) t = at::zeros({(int64_t)geot_content_fingerprint_scratch_bytes()}, like.options().dtype(at::kByte));
  return t;
}

void guard_drain() {
  for (const PendingVerdict &p : tl_pending) {
    bool have = false;
    for (int spin = 0; spin < 200000 && !have; ++spin) {
      have = __atomic_load_n(&p.slot[1], __ATOMIC_ACQUIRE) == p.seq;
      if (!have) __builtin_ia32_pause();
    }
    if (!have) { // a long queue in front of the fingerprint: wait properly
      TORCH_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(p.stream)) == hipSuccess, "hipStreamSynchronize failed");
      have = __atomic_load_n(&p.slot[1], __ATOMIC_ACQUIRE) == p.seq;
    }
    if (!have || __atomic_load_n(&p.slot[0], __ATOMIC_ACQUIRE) != 1) tl_guard_tripped = true; // (no verdict counts as a changed content)
  }
  tl_pending.clear();
}

void launch_fingerprint(const std::vector<at::Tensor> &ts, at::Tensor &fp, bool compare, int64_t *slot, int64_t seq, void *stream) {
  const void *bufs[4];
  size_t bytes[4];
  int n = 0;
  for (const at::Tensor &t : ts) {
    TORCH_CHECK(n < 4, "guard: at most four tensors per product");
    bufs[n] = t.data_ptr();
    bytes[n] = (size_t)t.numel() * (size_t)t.element_size();
    ++n;
  }
  GEOT_CALL(geot_content_fingerprint(bufs, bytes, n, reinterpret_cast<unsigned long long *>(fp.data_ptr<int64_t>()), compare ? 1 : 0, slot, seq,
                                     guard_scratch(ts.front(), stream).data_ptr(), stream));
}

std::vector<at::Tensor> tensors_of(std::initializer_list<const at::Tensor *> ts) {
  std::vector<at::Tensor> v;
  for (const at::Tensor *t : ts) v.push_back(*t);
  return v;
}

// fingerprint of `ts` as they are now, for a product that is being made from them (undefined when the guard is off); on the
// call's stream, behind whatever produced the tensors
at::Tensor guard_store(std::initializer_list<const at::Tensor *> ts) {
  if (!guard_on() || !guardable(ts)) return at::Tensor();
  at::Tensor fp = at::empty({2}, (*ts.begin())->options().dtype(at::kLong));
  launch_fingerprint(tensors_of(ts), fp, false, nullptr, 0, stream_of(**ts.begin()));
  return fp;
}

// A remembered product is about to be used: are `ts` still the bytes it was made from?  Called from inside the cache lookups,
// i.e. with g_mu (or a plan's wmu) HELD: it only notes the question.  guard_flush - run by the GuardFlush object at the top of
// every function that looks a product up, after the locks are gone - launches the fingerprint kernels; guard_settle reads the answers.
struct GuardRequest {
  at::Tensor fp;
  std::vector<at::Tensor> ts;
};
thread_local std::vector<GuardRequest> tl_requests;

void guard_check(const at::Tensor &fp, std::initializer_list<const at::Tensor *> ts) {
  if (!guard_on() || !fp.defined() || !guardable(ts)) return;
  const std::pair<const void *, const void *> what{fp.data_ptr(), (*ts.begin())->data_ptr()};
  for (const auto &a : tl_asked)
    if (a == what) return;
  tl_asked.push_back(what);
  tl_requests.push_back(GuardRequest{fp, tensors_of(ts)});
}

// The fingerprint kernels run on a SIDE stream (one per thread and device, from torch's pool): they wait for an event recorded
// on the call's stream at this point - everything that may have written the tensors is in front of it - and then read
// BESIDE the call's own kernels instead of in front of them (8-16 streamed bytes per edge: +4 % on a 115 M-edge mh_spmm when
// they ran in line).  Nothing downstream depends on them until guard_settle looks at the verdicts, and the operator does not
// return before it has: the tensors they read are the caller's arguments, alive until then.
struct GuardSide {
  void *stream = nullptr;
  hipEvent_t entry = nullptr;
};
GuardSide &guard_side(int device) {
  static thread_local std::map<int, GuardSide> sides;
  GuardSide &g = sides[device];
  if (!g.stream) {
    g.stream = c10::hip::getStreamFromPoolMasqueradingAsCUDA(/*isHighPriority=*/false, (c10::DeviceIndex)device).stream();
    TORCH_CHECK(hipEventCreateWithFlags(&g.entry, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
  }
  return g;
}

void guard_flush() {
  if (tl_requests.empty()) return;
  std::vector<GuardRequest> reqs;
  reqs.swap(tl_requests);
  for (GuardRequest &r : reqs) {
    const at::Tensor &first = r.ts.front();
    const int device = (int)first.device().index();
    if ((int)tl_pending.size() >= kGuardSlots) guard_drain();
    Slot &s = slot_for(device);
    int64_t *slot = s.host + 8 + 2 * (s.guard_next++ % kGuardSlots);
    const int64_t seq = ++s.seq;
    void *main_stream = stream_of(first);
    void *stream = main_stream;
    if (g_opt.guard_side_stream) {
      GuardSide &side = guard_side(device);
      (void)guard_scratch(first, side.stream); // (zero-filled on the call's stream the first time: in front of the event below)
      TORCH_CHECK(hipEventRecord(side.entry, static_cast<hipStream_t>(main_stream)) == hipSuccess, "hipEventRecord failed");
      TORCH_CHECK(hipStreamWaitEvent(static_cast<hipStream_t>(side.stream), side.entry, 0) == hipSuccess, "hipStreamWaitEvent failed");
      stream = side.stream;
    }
    launch_fingerprint(r.ts, r.fp, true, slot, seq, stream);
    tl_pending.push_back(PendingVerdict{slot, seq, stream});
  }
}
struct GuardFlush { // declare FIRST in a function that looks products up: its destructor runs after the function's lock_guards'
  ~GuardFlush() {
    try {
      guard_flush();
    } catch (...) {
      tl_requests.clear();
      tl_guard_tripped = true; // (a question that could not be asked counts as a changed content: the call is repeated from the caller's bytes)
    }
  }
};

void clear_all_caches_locked();
// true: every product this operator call used was made from the bytes the tensors hold now
bool guard_settle();
// is `t` itself one of the remembered products (the transposed edge list handed to the backward pass)?  What is derived from
// those - the plan of the transposed graph, its weight in plan order - needs no fingerprint of its own: nobody but this file
// writes them, and the entry they belong to is checked against the caller's tensors in the same backward pass.
bool owned_product(const at::Tensor &t);

// ---- facts of an index tensor, keyed on its content identity ---------------------------------------------------------------
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
bool content_key(const at::Tensor &t, ContentKey *k) {
  if (t.is_inference() || !t.has_storage() || t.dim() > 2) return false; // inference tensors keep no version counter: never remembered
  k->storage = t.storage().unsafeGetStorageImpl();
  k->offset = t.storage_offset();
  k->numel = t.numel();
  k->version = t._version();
  k->dim = (int)t.dim();
  k->dtype = (int)t.scalar_type();
  for (int d = 0; d < 2; ++d) {
    k->size[d] = d < t.dim() ? t.size(d) : 1;
    k->stride[d] = d < t.dim() ? t.stride(d) : 0;
  }
  return true;
}
struct Facts {
  ContentKey key;
  c10::weak_intrusive_ptr<c10::StorageImpl> weak;
  int64_t rows;
  bool ascending;
  int64_t kmin, kmax;    // key range (sizes the sort of an index with descents)
  at::Tensor keys, perm; // stable sort of an index with descents (a few entries keep theirs)
  Produced made;         // ... and the event of that sort
  at::Tensor sort_fp;    // ... and the fingerprint of the index it sorted (guard_store)
};
std::list<Facts> g_facts; // most recent first, <= 16 entries
constexpr size_t kFactsMax = 16, kSortedKeep = 4;

struct FactsView {
  int64_t rows;
  bool ascending;
  bool cached;
  int64_t kmin, kmax;
};

// one pass: {index[-1], descents, min, max}
void probe_index(const at::Tensor &index, int64_t out4[4]) {
  TORCH_CHECK_INDEX(index.numel() > 0, "index -1 is out of bounds for dimension 0 with size 0");
  at::Tensor dev = at::empty({4}, index.options());
  void *st = stream_of(index);
  GEOT_CALL(geot_index_probe_range(index_ptr(index), index.numel(), dev.data_ptr<int64_t>(), st));
  Slot &s = slot_for(index.device().index());
  TORCH_CHECK(hipMemcpyAsync(s.host, dev.data_ptr<int64_t>(), 32, hipMemcpyDeviceToHost, static_cast<hipStream_t>(st)) == hipSuccess, "hipMemcpyAsync failed");
  TORCH_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(st)) == hipSuccess, "hipStreamSynchronize failed");
  std::memcpy(out4, s.host, 32);
}

void clear_all_caches_locked();

// Descent alarm (include/geot_hip.h, geot_set_alarm_word): the sorted kernels verify "ascending" as they go and repair a
// call whose index has descents on the device.  That only happens when a remembered fact was stale - the tensor was written
// behind its version counter (.data, DLPack, a raw pointer) - so every remembered fact is dropped here: the next call
// probes again and takes the sort path.  Checked at the start of every operator call (two pinned words).
void check_alarm(Slot &s) {
  if ((__atomic_load_n(&s.host[6], __ATOMIC_ACQUIRE) | __atomic_load_n(&s.host[7], __ATOMIC_ACQUIRE)) == 0) return;
  const int64_t repaired = __atomic_exchange_n(&s.host[6], (int64_t)0, __ATOMIC_ACQ_REL);
  const int64_t poisoned = __atomic_exchange_n(&s.host[7], (int64_t)0, __ATOMIC_ACQ_REL);
  {
    std::lock_guard<std::mutex> lk(g_mu);
    clear_all_caches_locked();
    ++g_stats.alarms;
  }
  TORCH_CHECK(!poisoned, "geot: an earlier call on this thread found DESCENTS in an index tensor that was ascending when it was probed: the "
              "tensor was written behind its version counter (.data, DLPack, a raw pointer).  That call used a reduction or dtype without "
              "float atomics to fall back on, so its output was filled with NaN.  The remembered facts have been dropped - repeat the call.");
  if (repaired)
    TORCH_WARN("geot: an index tensor was written behind its version counter (.data, DLPack, a raw pointer); the call that met it repaired "
               "itself on the device (zero-fill + float atomics, slow).  The remembered facts about index tensors have been dropped.");
}

// index: contiguous, 1-D, int64, on the GPU, non-empty checked inside
FactsView index_facts(const at::Tensor &index) {
  {
    Slot &s = slot_for(index.device().index());
    check_alarm(s);
    geot_set_alarm_word(s.host + 6); // (sticky per thread in the library; one slot per (thread, device))
  }
  ContentKey k;
  const bool keyed = g_opt.trust_version && content_key(index, &k);
  if (keyed) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto it = g_facts.begin(); it != g_facts.end(); ++it)
      if (it->key == k && !it->weak.expired()) {
        g_facts.splice(g_facts.begin(), g_facts, it);
        return {it->rows, it->ascending, true, it->kmin, it->kmax};
      }
  }
  TORCH_CHECK(!tl_capturing, "geot: this index tensor has not been seen before (or GEOT_TRUST_VERSION=0), and its row count / "
              "ordering cannot be read back while the stream is being captured into a graph.  Run the call once before the capture.");
  int64_t p[4] = {0, 0, 0, 0};
  probe_index(index, p);
  std::lock_guard<std::mutex> lk(g_mu);
  ++g_stats.probes;
  if (keyed) {
    g_facts.push_front(Facts{k, index.storage().getWeakStorageImpl(), p[0] + 1, p[1] == 0, p[2], p[3], {}, {}, {}});
    while (g_facts.size() > kFactsMax) g_facts.pop_back();
  }
  return {p[0] + 1, p[1] == 0, false, p[2], p[3]};
}

void remember_rows(const at::Tensor &index, int64_t rows) {
  ContentKey k;
  if (!content_key(index, &k)) return;
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto &f : g_facts)
    if (f.key == k) f.rows = rows;
}

// stable sort by key: keys that fit 32 bits go through geot_sort_index (radix passes over the bits in use only),
// anything else (negative keys, keys >= 2^32) through ATen's generic sort
std::pair<at::Tensor, at::Tensor> stable_sort_index(const at::Tensor &index, int64_t kmin, int64_t kmax) {
  const int64_t nnz = index.numel();
  if (geot_sort_supported(nnz, kmin, kmax)) {
    const size_t bytes = geot_sort_workspace_bytes(nnz);
    if (bytes) {
      at::Tensor keys = at::empty_like(index), perm = at::empty_like(index);
      at::Tensor ws = at::empty({(int64_t)bytes}, index.options().dtype(at::kByte));
      GEOT_CALL(geot_sort_index(index_ptr(index), nnz, kmax, keys.data_ptr<int64_t>(), perm.data_ptr<int64_t>(), ws.data_ptr(), bytes,
                                stream_of(index)));
      return {keys, perm};
    }
  }
  auto sorted = at::sort(index, /*stable=*/true, /*dim=*/0, /*descending=*/false);
  return {std::get<0>(sorted), std::get<1>(sorted)};
}

// (keys ascending, perm) of an index with descents
std::pair<at::Tensor, at::Tensor> sorted_form(const at::Tensor &index, int64_t kmin, int64_t kmax) {
  const GuardFlush flush_questions_;
  ContentKey k;
  const bool keyed = may_remember({&index}) && content_key(index, &k);
  if (keyed) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &f : g_facts)
      if (f.key == k && !f.weak.expired() && f.keys.defined()) {
        f.made.before_use(index, {&f.keys, &f.perm});
        guard_check(f.sort_fp, {&index});
        return {f.keys, f.perm};
      }
  }
  auto sorted = stable_sort_index(index, kmin, kmax);
  at::Tensor fp = keyed && !tl_capturing ? guard_store({&index}) : at::Tensor();
  std::lock_guard<std::mutex> lk(g_mu);
  ++g_stats.sorts;
  if (keyed && !tl_capturing) {
    size_t holders = 0;
    for (auto &f : g_facts) {
      if (f.key == k) {
        f.keys = sorted.first;
        f.perm = sorted.second;
        f.sort_fp = fp;
        f.made.mark(index);
      }
      if (f.keys.defined() && ++holders > kSortedKeep) f.keys = f.perm = at::Tensor();
    }
    enforce_cache_budget_locked();
  }
  return sorted;
}

// ---- int32 indices (the reference's Python wrappers cast to int32 for sddmm_coo_impl / csr_gws_impl,
// geot/gather_weight_scatter.py:10-11, geot/csr_gws.py) -> the int64 the kernels read, converted once per content
struct WidenedEntry {
  ContentKey key;
  WeakStorage narrow; // the caller's int32 tensor (weak: see WeakStorage)
  at::Tensor wide;
  Produced made;
  at::Tensor fp; // fingerprint of the narrow tensor (guard_store)
};
std::list<WidenedEntry> g_widened;

at::Tensor as_int64(const at::Tensor &t) {
  const GuardFlush flush_questions_;
  if (t.scalar_type() == at::kLong) return t.contiguous();
  ContentKey k;
  const bool keyed = may_remember({&t}) && content_key(t, &k);
  if (keyed) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto it = g_widened.begin(); it != g_widened.end(); ++it)
      if (it->key == k && !it->narrow.expired()) {
        g_widened.splice(g_widened.begin(), g_widened, it);
        g_widened.front().made.before_use(t, {&g_widened.front().wide});
        guard_check(g_widened.front().fp, {&t});
        return g_widened.front().wide;
      }
  }
  at::Tensor wide = t.to(at::kLong).contiguous();
  if (keyed && !tl_capturing) {
    at::Tensor fp = guard_store({&t});
    std::lock_guard<std::mutex> lk(g_mu);
    g_widened.push_front(WidenedEntry{k, weak_of(t), wide, {}, fp});
    g_widened.front().made.mark(t);
    while (g_widened.size() > 6) g_widened.pop_back();
    enforce_cache_budget_locked();
  }
  return wide;
}

constexpr int64_t kPublishMaxEdges = 1 << 20; // (larger calls keep the copy that completes under their kernels)

// ---- the row rule without stalling the GPU ---------------------------------------------------------------------------------
// launch(rows) allocates the output for `rows` rows and enqueues the kernels.  `guess` comes from the facts.
template <typename Launch> at::Tensor with_row_rule(const at::Tensor &index, int64_t guess, bool guess_is_fresh, Launch launch) {
  if (guess_is_fresh) return launch(guess); // this very call has just read index[-1] (the probe)
  if (g_opt.trust_version >= 2) return launch(guess); // opt-in: index[-1] is as trusted as the sortedness (same content key)
  if (tl_capturing) return launch(guess);             // a graph has static shapes: the remembered count is the contract
  void *st = stream_of(index);
  Slot &s = slot_for(index.device().index());
  const int64_t *last = index_ptr(index) + (index.numel() - 1);
  if (g_opt.speculate_rows && g_opt.publish_rows && index.numel() <= kPublishMaxEdges) {
    // launch-bound calls: no copy, no event - the first kernel of the launch writes index[-1] and a sequence number into
    // the pinned slot while it runs, the host spins on the sequence number (geot_publish_word)
    const int64_t seq = ++s.seq;
    GEOT_CALL(geot_publish_word(last, s.host + 4, seq));
    at::Tensor out;
    try {
      out = launch(guess);
    } catch (...) {
      geot_publish_pending();
      throw;
    }
    bool have = false;
    if (!geot_publish_pending()) { // taken by that launch
      for (int spin = 0; spin < 40000 && !have; ++spin) {
        have = __atomic_load_n(&s.host[5], __ATOMIC_ACQUIRE) == seq;
        if (!have) __builtin_ia32_pause();
      }
      if (!have) { // a long queue ahead of this call: wait properly
        TORCH_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(st)) == hipSuccess, "hipStreamSynchronize failed");
        have = __atomic_load_n(&s.host[5], __ATOMIC_ACQUIRE) == seq;
        TORCH_CHECK(have, "geot: the row-count word was not published by the kernel");
      }
      s.host[0] = s.host[4];
      std::lock_guard<std::mutex> lk(g_mu);
      ++g_stats.published;
    } else { // a path whose first kernel does not publish (small by the size rule above): read it back after the fact
      TORCH_CHECK(hipMemcpyAsync(s.host, last, 8, hipMemcpyDeviceToHost, static_cast<hipStream_t>(st)) == hipSuccess, "hipMemcpyAsync failed");
      TORCH_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(st)) == hipSuccess, "hipStreamSynchronize failed");
    }
    const int64_t rows = s.host[0] + 1;
    if (rows != guess) {
      {
        std::lock_guard<std::mutex> lk(g_mu);
        ++g_stats.row_mismatches;
      }
      remember_rows(index, rows);
      out = launch(rows);
    }
    return out;
  }
  TORCH_CHECK(hipMemcpyAsync(s.host, last, 8, hipMemcpyDeviceToHost, static_cast<hipStream_t>(st)) == hipSuccess, "hipMemcpyAsync failed");
  if (!g_opt.speculate_rows) {
    TORCH_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(st)) == hipSuccess, "hipStreamSynchronize failed");
    const int64_t rows = s.host[0] + 1;
    if (rows != guess) remember_rows(index, rows);
    return launch(rows);
  }
  TORCH_CHECK(hipEventRecord(s.ev, static_cast<hipStream_t>(st)) == hipSuccess, "hipEventRecord failed");
  at::Tensor out = launch(guess); // queued behind the copy: the copy completes while the kernels run
  TORCH_CHECK(hipEventSynchronize(s.ev) == hipSuccess, "hipEventSynchronize failed");
  const int64_t rows = s.host[0] + 1;
  if (rows != guess) { // the content changed under the same identity and version (a write through .data)
    {
      std::lock_guard<std::mutex> lk(g_mu);
      ++g_stats.row_mismatches;
    }
    remember_rows(index, rows);
    out = launch(rows);
  }
  return out;
}

// ---- dense graphs: source-blocked kernel (csrc/seg_slab.hip), Phase A -------------------------------------------------------
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

bool slab_worthwhile(int64_t nnz, int64_t out_rows, int64_t src_rows, int64_t rowbytes, int dtype = GEOT_F32) {
  if ((rowbytes != 256 && rowbytes != 512 && rowbytes != 1024) || nnz >= ((int64_t)1 << 31) || nnz < 8000000 || out_rows < 1 ||
      src_rows >= ((int64_t)1 << 31) || !geot_slab_full_chip()) // (a partitioned / CU-masked device: the rule below was not measured there)
    return false;
  const int64_t units = (int64_t)geot_slab_units() * (1024 / rowbytes);
  const int64_t rpg = geot_slab_rows_per_group_dtype(1, 1, dtype); // (16-bit storage: fp32 accumulators, half the rows per group)
  const int64_t rounds = std::max<int64_t>(1, (out_rows + rpg * units - 1) / (rpg * units));
  // uses of a source row per XCD and round; measured (profiles/r02/bench_slab_density_rule.txt, 120 M edges): 10 -> 1.50x,
  // 4.8 -> 1.34x, 2.8 -> 1.22x, 1.6 -> 1.09x, 0.8 -> 0.86x at 512-B rows; 8.8 -> 2.08x, 2.4 -> 1.62x, 1.8 -> 1.43x (Reddit2: 23 M
  // edges), 1.0 -> 1.13x at 1 KiB.  A graph routed here is only a CANDIDATE: its plan is tried against the per-edge kernels on
  // first use (plan_or_edges), so the threshold for 1-KiB rows sits where the plan starts to win, not where it wins clearly.
  return (double)nnz / rounds / 8.0 / (double)std::max<int64_t>(src_rows, 1) >= (rowbytes == 1024 ? 1.0 : 2.0);
}

// Phase A in ATen: the reference formulation of the plan (generic passes, one stable sort, a host loop over the virtual
// rows).  Serves CPU tensors (tests/test_slab_plan.py emulates the kernel on its output) and cross-checks the device
// builder below, which produces the same arrays bit for bit (tests/test_gpu_slab.py).  dst_index ascending.
std::shared_ptr<SlabPlanHolder> slab_build_aten(const at::Tensor &src_index, const at::Tensor &dst_index, int64_t out_rows, int64_t src_rows,
                                                int64_t rowbytes, int weight_mode, int64_t heads, int64_t slab_bytes, int64_t rows_per_group,
                                                int64_t units_override) {
  auto H = std::make_shared<SlabPlanHolder>();
  const int64_t nnz = dst_index.numel();
  const int64_t lanes = rowbytes / 16;
  const int64_t units = units_override > 0 ? units_override : (int64_t)geot_slab_units() * (64 / lanes);
  const int64_t R = rows_per_group > 0 ? rows_per_group : geot_slab_rows_per_group(weight_mode, heads);
  const auto lopt = dst_index.options();
  // row pointers of the ASCENDING dst_index by binary search (rows + 1 searches; a histogram would spend 23 ms of global
  // atomics on the hubs of a 115 M-edge graph - more than the rest of Phase A together)
  at::Tensor bounds = at::searchsorted(dst_index, at::arange(out_rows + 1, lopt));
  at::Tensor rowptr = bounds.slice(0, 0, out_rows).contiguous();
  at::Tensor counts = (bounds.slice(0, 1, out_rows + 1) - rowptr).contiguous();
  const int64_t nonempty = counts.gt(0).sum().item<int64_t>();
  const int64_t rounds0 = std::max<int64_t>(1, (nonempty + R * units - 1) / (R * units));
  const int64_t budget = std::max<int64_t>(256, (nnz + rounds0 * units - 1) / (rounds0 * units));
  const int64_t cap = std::max<int64_t>(64, budget / 2);
  at::Tensor nv_row = at::div(counts + (cap - 1), cap, "floor");
  at::Tensor vstart = at::cumsum(nv_row, 0) - nv_row;
  const int64_t V = nv_row.sum().item<int64_t>();
  at::Tensor v_row = at::repeat_interleave(nv_row, c10::optional<int64_t>(V)); // dst row of every virtual row
  at::Tensor v_piece = at::arange(V, lopt) - vstart.index_select(0, v_row);
  at::Tensor v_cnt = at::clamp_max(counts.index_select(0, v_row) - v_piece * cap, cap);
  // groups: greedy over consecutive virtual rows, <= R rows and <= budget edges
  at::Tensor v_cnt_h = v_cnt.cpu();
  const int64_t *vc = v_cnt_h.data_ptr<int64_t>();
  std::vector<int64_t> starts, gedges;
  for (int64_t i = 0; i < V;) {
    int64_t j = i, e = 0;
    while (j < V && j - i < R && (j == i || e + vc[j] <= budget)) e += vc[j++];
    starts.push_back(i);
    gedges.push_back(e);
    i = j;
  }
  const int64_t G = (int64_t)starts.size();
  std::vector<int64_t> order(G), pos_of_group(G), g_begin(G + 1, 0), g_nv_sorted(G), g_v0_sorted(G), nv_of_group(G);
  for (int64_t g = 0; g < G; ++g) {
    order[g] = g;
    nv_of_group[g] = (g + 1 < G ? starts[g + 1] : V) - starts[g];
  }
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return gedges[a] > gedges[b]; });
  for (int64_t p = 0; p < G; ++p) {
    pos_of_group[order[p]] = p;
    g_begin[p + 1] = g_begin[p] + gedges[order[p]];
    g_nv_sorted[p] = nv_of_group[order[p]];
    g_v0_sorted[p] = starts[order[p]];
  }
  auto to_dev = [&](const std::vector<int64_t> &v, at::ScalarType dt) {
    at::Tensor t = at::empty({(int64_t)std::max<size_t>(v.size(), 1)}, at::TensorOptions().dtype(at::kLong));
    if (!v.empty()) std::memcpy(t.data_ptr<int64_t>(), v.data(), v.size() * sizeof(int64_t));
    else t.zero_();
    return t.to(dst_index.device(), dt);
  };
  at::Tensor nv_of_group_t = to_dev(nv_of_group, at::kLong);
  at::Tensor group_of_vrow = G > 0 ? at::repeat_interleave(nv_of_group_t.slice(0, 0, G), c10::optional<int64_t>(V)) : at::empty({0}, lopt);
  at::Tensor start_of_group = to_dev(starts, at::kLong);
  at::Tensor pos_t = to_dev(pos_of_group, at::kLong);
  // per edge: virtual row, group position, slab, row in group -> one stable sort
  at::Tensor e_id = at::arange(nnz, lopt);
  at::Tensor vrow_e = vstart.index_select(0, dst_index) + at::div(e_id - rowptr.index_select(0, dst_index), cap, "floor");
  e_id = at::Tensor();
  at::Tensor gid_e = group_of_vrow.index_select(0, vrow_e);
  at::Tensor dl_e = vrow_e - start_of_group.index_select(0, gid_e);
  vrow_e = at::Tensor();
  int64_t slab_shift = 0;                       // slabs of 2^k source rows (the kernel finds an edge's slab with a shift)
  while (((int64_t)2 << slab_shift) * rowbytes <= slab_bytes) ++slab_shift;
  const int64_t slab_rows = (int64_t)1 << slab_shift;
  const int64_t n_slabs = std::max<int64_t>(1, (src_rows + slab_rows - 1) / slab_rows);
  at::Tensor key = (pos_t.index_select(0, gid_e) * n_slabs + at::div(src_index, slab_rows, "floor").clamp_(0, n_slabs - 1)) * R + dl_e;
  gid_e = at::Tensor();
  // (on the GPU the composite key's range is known: radix passes over its bits only)
  at::Tensor perm = key.is_cuda() ? stable_sort_index(key, 0, std::max<int64_t>(G, 1) * n_slabs * R).second
                                  : std::get<1>(at::sort(key, /*stable=*/true, 0, false));
  key = at::Tensor();
  at::Tensor e_src = src_index.index_select(0, perm).to(at::kInt);
  at::Tensor e_dl = dl_e.index_select(0, perm).to(at::kByte);
  at::Tensor e_perm = perm.to(at::kInt);
  perm = dl_e = at::Tensor();
  // outputs of the virtual rows: the dst row, or a carry slot for the pieces of a split row
  at::Tensor split_v = nv_row.index_select(0, v_row).gt(1);
  at::Tensor carry_slot = at::cumsum(split_v.to(at::kLong), 0) - 1;
  at::Tensor v_out = at::where(split_v, -(carry_slot + 1), v_row).contiguous();
  at::Tensor split_rows = at::nonzero(nv_row.gt(1)).flatten().contiguous();
  at::Tensor c_count = nv_row.index_select(0, split_rows).to(at::kInt).contiguous();
  at::Tensor c_first = split_rows.numel() ? carry_slot.index_select(0, vstart.index_select(0, split_rows)).contiguous() : split_rows;
  const int64_t n_carry = V > 0 ? split_v.sum().item<int64_t>() : 0;
  auto nonempty_t = [&](at::Tensor t) { return t.numel() ? t : at::zeros({1}, t.options()); };
  at::Tensor g_begin_t = to_dev(g_begin, at::kLong), g_v0_t = to_dev(g_v0_sorted, at::kInt), g_nv_t = to_dev(g_nv_sorted, at::kInt);
  at::Tensor v_row32 = v_row.to(at::kInt).contiguous();
  at::Tensor v_total = counts.index_select(0, v_row).to(at::kInt).contiguous();      // edges of the virtual row's whole dst row
  at::Tensor c_total = counts.index_select(0, split_rows).contiguous();
  H->keep = {nonempty_t(e_src), nonempty_t(e_dl), nonempty_t(e_perm), g_begin_t, g_v0_t, g_nv_t, nonempty_t(v_out),
             nonempty_t(split_rows), nonempty_t(c_first), nonempty_t(c_count), nonempty_t(v_row32), nonempty_t(v_total), nonempty_t(c_total)};
  geot_slab_plan &P = H->plan;
  P.e_src = H->keep[0].data_ptr<int32_t>();
  P.e_dl = H->keep[1].data_ptr<uint8_t>();
  P.e_perm = H->keep[2].data_ptr<int32_t>();
  P.g_begin = H->keep[3].data_ptr<int64_t>();
  P.g_vrow0 = H->keep[4].data_ptr<int32_t>();
  P.g_nv = H->keep[5].data_ptr<int32_t>();
  P.v_out = H->keep[6].data_ptr<int64_t>();
  P.c_row = H->keep[7].data_ptr<int64_t>();
  P.c_first = H->keep[8].data_ptr<int64_t>();
  P.c_count = H->keep[9].data_ptr<int32_t>();
  P.v_row = H->keep[10].data_ptr<int32_t>();
  P.v_total = H->keep[11].data_ptr<int32_t>();
  P.c_total = H->keep[12].data_ptr<int64_t>();
  P.n_groups = G;
  P.n_vrows = V;
  P.n_carry = n_carry;
  P.n_split = split_rows.numel();
  P.nnz = nnz;
  P.units = (int32_t)units;
  P.rows_per_group = (int32_t)R;
  P.slab_shift = (int32_t)slab_shift;
  P.n_slabs = (int32_t)std::min<int64_t>(n_slabs, INT32_MAX);
  H->rounds = (G + units - 1) / units;
  H->budget = budget;
  H->cap = cap;
  H->slabs = n_slabs;
  H->slab_rows = slab_rows;
  return H;
}

// Phase A on the device (csrc/seg_plan.hip): three calls into the library, two 8..64-byte read-backs (the sizes of the
// arrays allocated here), no host loop.  nullptr: the library declined (keys out of range, sort key beyond 32 bits).
std::shared_ptr<SlabPlanHolder> slab_build_device(const at::Tensor &src_index, const at::Tensor &dst_index, int64_t out_rows, int64_t src_rows,
                                                  int64_t rowbytes, int weight_mode, int64_t heads, int64_t slab_bytes, int64_t rows_per_group,
                                                  int64_t units_override) {
  auto H = std::make_shared<SlabPlanHolder>();
  const int64_t nnz = dst_index.numel();
  const int64_t lanes = rowbytes / 16;
  const int64_t units = units_override > 0 ? units_override : (int64_t)geot_slab_units() * (64 / lanes);
  const int64_t R = rows_per_group > 0 ? rows_per_group : geot_slab_rows_per_group(weight_mode, heads);
  void *st = stream_of(dst_index);
  geot_slab_plan_job job;
  std::memset(&job, 0, sizeof(job));
  job.src_index = index_ptr(src_index);
  job.dst_index = index_ptr(dst_index);
  job.nnz = nnz;
  job.out_rows = out_rows;
  job.src_rows = src_rows;
  job.rowbytes = rowbytes;
  job.slab_bytes = slab_bytes;
  job.units = units;
  job.rows_per_group = (int32_t)R;
  const auto bopt = dst_index.options().dtype(at::kByte), iopt = dst_index.options().dtype(at::kInt), lopt = dst_index.options();
  auto scratch = [&](int stage) { return at::empty({(int64_t)std::max<size_t>(geot_slab_plan_scratch_bytes(&job, stage), 256)}, bopt); };
  auto declined = [&](int rc) {
    if (rc == GEOT_EUNSUPPORTED) return true;
    TORCH_CHECK(rc == GEOT_OK, "geot slab plan failed (code ", rc, "): ", geot_last_error());
    return false;
  };
  at::Tensor s1 = scratch(1);
  if (declined(geot_slab_plan_rows(&job, s1.data_ptr(), s1.numel(), st))) return nullptr;
  const int64_t V = job.n_vrows, NS = job.n_split;
  auto table = [&](int64_t n, const at::TensorOptions &o) { return n > 0 ? at::empty({n}, o) : at::zeros({1}, o); };
  at::Tensor v_out = table(V, lopt), v_row = table(V, iopt), v_total = table(V, iopt);
  at::Tensor c_row = table(NS, lopt), c_first = table(NS, lopt), c_count = table(NS, iopt), c_total = table(NS, lopt);
  at::Tensor s2 = scratch(2);
  if (declined(geot_slab_plan_groups(&job, s1.data_ptr(), s2.data_ptr(), s2.numel(), v_out.data_ptr<int64_t>(), v_row.data_ptr<int32_t>(),
                                     v_total.data_ptr<int32_t>(), c_row.data_ptr<int64_t>(), c_first.data_ptr<int64_t>(), c_count.data_ptr<int32_t>(),
                                     c_total.data_ptr<int64_t>(), st)))
    return nullptr;
  const int64_t G = job.n_groups;
  at::Tensor g_begin = at::empty({G + 1}, lopt), g_v0 = at::empty({G}, iopt), g_nv = at::empty({G}, iopt);
  at::Tensor e_src = at::empty({nnz}, iopt), e_dl = at::empty({nnz}, bopt), e_perm = at::empty({nnz}, iopt);
  at::Tensor s3 = scratch(3);
  if (declined(geot_slab_plan_edges(&job, s1.data_ptr(), s2.data_ptr(), s3.data_ptr(), s3.numel(), g_begin.data_ptr<int64_t>(), g_v0.data_ptr<int32_t>(),
                                    g_nv.data_ptr<int32_t>(), e_src.data_ptr<int32_t>(), e_dl.data_ptr<uint8_t>(), e_perm.data_ptr<int32_t>(), st)))
    return nullptr;
  H->keep = {e_src, e_dl, e_perm, g_begin, g_v0, g_nv, v_out, c_row, c_first, c_count, v_row, v_total, c_total};
  geot_slab_plan &P = H->plan;
  P.e_src = e_src.data_ptr<int32_t>();
  P.e_dl = e_dl.data_ptr<uint8_t>();
  P.e_perm = e_perm.data_ptr<int32_t>();
  P.g_begin = g_begin.data_ptr<int64_t>();
  P.g_vrow0 = g_v0.data_ptr<int32_t>();
  P.g_nv = g_nv.data_ptr<int32_t>();
  P.v_out = v_out.data_ptr<int64_t>();
  P.c_row = c_row.data_ptr<int64_t>();
  P.c_first = c_first.data_ptr<int64_t>();
  P.c_count = c_count.data_ptr<int32_t>();
  P.v_row = v_row.data_ptr<int32_t>();
  P.v_total = v_total.data_ptr<int32_t>();
  P.c_total = c_total.data_ptr<int64_t>();
  P.n_groups = G;
  P.n_vrows = V;
  P.n_carry = job.n_carry;
  P.n_split = NS;
  P.nnz = nnz;
  P.units = (int32_t)units;
  P.rows_per_group = (int32_t)R;
  P.slab_shift = job.slab_shift;
  P.n_slabs = job.n_slabs;
  H->rounds = (G + units - 1) / units;
  H->budget = job.budget;
  H->cap = job.cap;
  H->slabs = job.n_slabs;
  H->slab_rows = (int64_t)1 << job.slab_shift;
  return H;
}

std::shared_ptr<SlabPlanHolder> slab_build(const at::Tensor &src_index, const at::Tensor &dst_index, int64_t out_rows, int64_t src_rows,
                                           int64_t rowbytes, int weight_mode, int64_t heads, int64_t slab_bytes, int64_t rows_per_group,
                                           int64_t units_override) {
  if (dst_index.is_cuda() && g_opt.slab_builder == 0 && dst_index.numel() > 0 && dst_index.numel() < ((int64_t)1 << 31) && out_rows > 0 &&
      out_rows < ((int64_t)1 << 31)) {
    if (auto H = slab_build_device(src_index, dst_index, out_rows, src_rows, rowbytes, weight_mode, heads, slab_bytes, rows_per_group, units_override))
      return H;
  }
  return slab_build_aten(src_index, dst_index, out_rows, src_rows, rowbytes, weight_mode, heads, slab_bytes, rows_per_group, units_override);
}

constexpr int64_t kSlabBytes = 2 << 20; // measured (profiles/r02/bench_slab.txt)

struct SlabEntry {
  ContentKey k1, k2;
  int64_t rows, src_rows, rowbytes, heads;
  int wmode;
  int rpg;   // rows per group the plan was built for: the only thing the weight mode / head count changes
  WeakStorage w1, w2; // the edge list the plan was built from (weak: the plan goes when the edge list dies)
  std::shared_ptr<SlabPlanHolder> plan;
};
std::list<SlabEntry> g_slab;
std::list<std::pair<ContentKey, ContentKey>> g_sightings; // edge lists seen once (no tensors held)

std::shared_ptr<SlabPlanHolder> slab_plan_for(const at::Tensor &si, const at::Tensor &di, int64_t rows, const at::Tensor &src,
                                              int wmode, int64_t heads) {
  const GuardFlush flush_questions_;
  const bool f32 = src.scalar_type() == at::kFloat;
  if (g_opt.slab_mode < 0 || rows < 1 || !(f32 || src.scalar_type() == at::kHalf || src.scalar_type() == at::kBFloat16)) return nullptr;
  const int dt = dtype_code(src, "slab");
  const int64_t rowbytes = (src.numel() / std::max<int64_t>(src.size(0), 1)) * src.element_size(), nnz = di.numel();
  // rows of 256 / 512 / 1024 bytes; 128-byte rows run too but were measured slower than the per-edge kernels (DESIGN.md
  // section 3.1d): only when the path is forced
  if ((rowbytes != 256 && rowbytes != 512 && rowbytes != 1024 && !(rowbytes == 128 && g_opt.slab_mode == 1)) || nnz == 0 ||
      nnz >= ((int64_t)1 << 31))
    return nullptr;
  if (g_opt.slab_mode != 1 && !slab_worthwhile(nnz, rows, src.size(0), rowbytes, dt)) return nullptr;
  ContentKey k1, k2;
  if (!may_remember({&si, &di}) || !content_key(si, &k1) || !content_key(di, &k2)) return nullptr;
  const int rpg = geot_slab_rows_per_group_dtype(wmode, heads, dt);
  {
    std::lock_guard<std::mutex> lk(g_mu);
    sweep_expired_locked();
    for (auto it = g_slab.begin(); it != g_slab.end(); ++it)
      if (it->k1 == k1 && it->k2 == k2 && it->rows == rows && it->src_rows == src.size(0) && it->rowbytes == rowbytes &&
          it->rpg == rpg && !it->w1.expired() && !it->w2.expired()) { // (a plan serves every weight mode with its R)
        g_slab.splice(g_slab.begin(), g_slab, it);
        g_slab.front().plan->made.before_use(src, g_slab.front().plan->keep);
        guard_check(g_slab.front().plan->fp, {&si, &di}); // (a rejected plan has released its fingerprint: the per-edge kernels read the caller's bytes)
        return g_slab.front().plan;
      }
    if (tl_capturing) return nullptr; // Phase A synchronises: never inside a capture (the per-edge kernels serve the call)
    if (g_opt.slab_mode != 1) { // first sighting of this edge list: only remember it - a one-shot call never pays for Phase A
      bool seen = false;
      for (auto &sg : g_sightings) seen |= (sg.first == k1 && sg.second == k2);
      if (!seen) {
        g_sightings.emplace_back(k1, k2);
        if (g_sightings.size() > 64) g_sightings.pop_front();
        return nullptr;
      }
    }
  }
  // (the build reads two small records back anyway: wait here, so that plan_us is the build and not the queue in front of it)
  if (di.is_cuda()) (void)hipStreamSynchronize(static_cast<hipStream_t>(stream_of(di)));
  const auto t0 = std::chrono::steady_clock::now();
  std::shared_ptr<SlabPlanHolder> plan;
  try {
    plan = slab_build(si, di, rows, src.size(0), rowbytes, wmode, heads, kSlabBytes, rpg, 0);
  } catch (const c10::Error &) {
    // Phase A needs ~80 bytes per edge of transient memory and keeps 9: if that does not fit, the per-edge kernels serve
    // the call (and every later one: the sighting is forgotten, a later call may try again)
    std::lock_guard<std::mutex> lk(g_mu);
    g_sightings.remove_if([&](const std::pair<ContentKey, ContentKey> &sg) { return sg.first == k1 && sg.second == k2; });
    return nullptr;
  }
  const auto us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
  plan->fp = owned_product(si) && owned_product(di) ? at::Tensor() : guard_store({&si, &di});
  plan->made.mark(src);
  std::lock_guard<std::mutex> lk(g_mu);
  ++g_stats.plans_built;
  g_stats.plan_us += us;
  g_slab.push_front(SlabEntry{k1, k2, rows, src.size(0), rowbytes, heads, wmode, rpg, weak_of(si), weak_of(di), plan});
  while ((int)g_slab.size() > g_opt.slab_keep) g_slab.pop_back();
  enforce_cache_budget_locked();
  return plan;
}

// (the persistent grids of a process take turns on a device INSIDE the library - geot_slab_spmm / geot_slab_sddmm, seg_slab.hip
// "SlabTurn" - so every caller of the C ABI gets it, not only this plugin)
// false: the plan's arrays have been released (a trial on another thread rejected it) - the caller runs the per-edge kernels
bool run_slab(SlabPlanHolder &H, const void *weight, int wmode, const at::Tensor &src, at::Tensor &out, int64_t heads, int64_t feat,
              int red = GEOT_REDUCE_SUM) {
  const std::vector<at::Tensor> pinned = H.pinned(); // this launch's own references: a release() meanwhile cannot free under the kernel
  if (pinned.empty()) return false;
  auto &ws = workspace(src, geot_slab_workspace_bytes(&H.plan, heads * feat));
  GEOT_CALL(geot_slab_spmm(&H.plan, weight, wmode, src.data_ptr(), out.data_ptr(), heads, feat, src.size(0), out.size(0), dtype_code(src, "slab"),
                           red, ws.data_ptr(), ws.numel(), stream_of(src)));
  H.launched_on(src, pinned);
  return true;
}

// One operator call over a graph that has a plan: the plan's kernels or the per-edge kernels (SlabPlanHolder::verdict).
// run_plan(o) / run_edges(o) enqueue the whole call into `o`; run_plan returns false when the plan's arrays are gone (released by
// a trial on another thread) - the per-edge kernels then serve the call.  kind: 0 forward reduction, 1 SDDMM.
//
// The TRIAL (first use of an undecided plan): both ways are run once UNTIMED first - the first launch of a kernel loads its
// code object, the workspace is allocated, a head-major weight is transposed, a static weight is not in plan order yet: none of
// that may decide a permanent verdict - and then kTrialReps alternating repetitions are timed with events on the call's stream;
// the best time of each side decides.  One trial at a time per plan (trial_mu): a thread that finds a trial in progress serves
// its call with the per-edge kernels and leaves the decision to the trying thread.  A plan that loses CLEARLY (> 10 %) is
// released; one that loses narrowly keeps its arrays and is tried once more after kRetrialAfter further calls.
constexpr int kTrialReps = 2, kRetrialAfter = 256;
template <typename RunPlan, typename RunEdges>
at::Tensor plan_or_edges(const std::shared_ptr<SlabPlanHolder> &plan, int kind, at::Tensor o, const at::Tensor &on, RunPlan run_plan, RunEdges run_edges) {
  int v = plan ? plan->verdict[kind].load() : 2;
  if (plan && g_opt.slab_mode == 1) v = 1;                          // forced: no trial (a released plan falls through to the per-edge kernels)
  if (v == 2 && plan && g_opt.slab_mode == 0 && !tl_capturing && plan->trials_done[kind].load() == 1 &&
      ++plan->uses_since_trial[kind] >= kRetrialAfter && !plan->pinned().empty())
    v = 0;                                                          // a narrow loss is looked at once more
  if (v == 0 && tl_capturing) v = 2;                                // an undecided plan is not tried inside a capture (the trial waits)
  if (v == 0) {
    std::unique_lock<std::mutex> trying(plan->trial_mu, std::try_to_lock);
    if (!trying.owns_lock()) v = 2;                                 // another thread is trying this plan right now
    else if (plan->verdict[kind].load() != 0 && plan->uses_since_trial[kind].load() < kRetrialAfter) v = plan->verdict[kind].load(); // decided meanwhile
    else {
      hipStream_t st = static_cast<hipStream_t>(stream_of(on));
      constexpr int kEv = 2 * kTrialReps + 1;
      hipEvent_t ev[kEv] = {};
      for (hipEvent_t &e : ev) TORCH_CHECK(hipEventCreate(&e) == hipSuccess, "hipEventCreate failed");
      auto destroy = [&]() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); };
      at::Tensor o2 = at::empty_like(o);
      try {
        bool plan_ok = run_plan(o);                                 // untimed: warms both paths
        run_edges(o2);
        TORCH_CHECK(hipEventRecord(ev[0], st) == hipSuccess, "hipEventRecord failed");
        for (int r = 0; r < kTrialReps && plan_ok; ++r) {
          plan_ok = run_plan(o);
          TORCH_CHECK(hipEventRecord(ev[2 * r + 1], st) == hipSuccess, "hipEventRecord failed");
          run_edges(o2);
          TORCH_CHECK(hipEventRecord(ev[2 * r + 2], st) == hipSuccess, "hipEventRecord failed");
        }
        if (!plan_ok) {                                             // (released under our feet: cannot happen while we hold trial_mu, but stay safe)
          destroy();
          run_edges(o);
          return o;
        }
        TORCH_CHECK(hipEventSynchronize(ev[kEv - 1]) == hipSuccess, "hipEventSynchronize failed");
        float t_plan = 1e30f, t_edges = 1e30f;
        for (int r = 0; r < kTrialReps; ++r) {
          float a = 0.f, b = 0.f;
          TORCH_CHECK(hipEventElapsedTime(&a, ev[2 * r], ev[2 * r + 1]) == hipSuccess && hipEventElapsedTime(&b, ev[2 * r + 1], ev[2 * r + 2]) == hipSuccess,
                      "hipEventElapsedTime failed");
          t_plan = std::min(t_plan, a);
          t_edges = std::min(t_edges, b);
        }
        destroy();
        plan->trial_ms[kind][0] = t_plan;
        plan->trial_ms[kind][1] = t_edges;
        const bool keep_plan = t_plan <= t_edges;
        const bool clear_loss = t_plan > 1.1f * t_edges;
        {
          std::lock_guard<std::mutex> lk(g_mu);
          ++g_stats.plan_trials;
          ++g_stats.slab_calls; // (operator calls that ran over a plan - a trial counts once, whatever it repeats)
          if (!keep_plan) ++g_stats.plans_rejected;
          g_stats.trial_plan_us = (int64_t)(t_plan * 1e3f);
          g_stats.trial_edges_us = (int64_t)(t_edges * 1e3f);
          plan->verdict[kind] = keep_plan ? 1 : 2;
          plan->uses_since_trial[kind] = 0;
          ++plan->trials_done[kind];
          // what makes the forward lose clearly makes the SDDMM lose: one decision, and the arrays go - unless the SDDMM has
          // already WON a trial of its own (then its launches keep the arrays they are reading)
          if (clear_loss && kind == 0 && plan->verdict[1].load() != 1) {
            plan->verdict[1] = 2;
            plan->trials_done[0] = plan->trials_done[1] = 2;        // (no arrays, no second look)
            plan->release();
          } else if (clear_loss) {
            plan->trials_done[kind] = 2;
          }
        }
        return keep_plan ? o : o2;
      } catch (...) {
        destroy();
        throw;
      }
    }
  }
  if (v == 1 && run_plan(o)) {
    std::lock_guard<std::mutex> lk(g_mu);
    ++g_stats.slab_calls;
    return o;
  }
  run_edges(o);
  return o;
}

// ---- index_scatter -------------------------------------------------------------------------------------------------------------
at::Tensor index_scatter_op(const int64_t dim, const at::Tensor &index_in, const at::Tensor &src, const c10::string_view reduce,
                            const bool sorted) {
  // checks of index_scatter_cuda (csrc/cuda/index_scatter_cuda.cu:90-94), same texts
  TORCH_CHECK(dim >= 0 && dim < src.dim(), "dim must be non-negative and less than input dimensions");
  TORCH_CHECK(index_in.dim() == 1, "index must be 1 dimensional");
  TORCH_CHECK(src.size(dim) == index_in.size(0), "index length must be equal to src dimension size");
  const int red = reduce_code(reduce);
  TORCH_CHECK_INDEX(index_in.numel() > 0, "index -1 is out of bounds for dimension 0 with size 0"); // index[-1] of the reference
  require_gpu("index_scatter", {&index_in, &src});
  const int dt = dtype_code(src, sorted ? "index_scatter_sorted" : "index_scatter_unsorted");
  GEOT_DEVICE_GUARD(src);
  at::Tensor moved = (dim == 0 ? src : src.movedim(dim, 0)).contiguous();
  at::Tensor index = index_in.contiguous();
  index_ptr(index);
  const int64_t nnz = index.numel(), feat = moved.numel() / nnz;
  const FactsView f = index_facts(index); // `sorted` is a promise the reference never checks; neither flag is trusted
  auto shape = moved.sizes().vec();
  std::pair<at::Tensor, at::Tensor> kp;
  const bool atomic_flush = !f.ascending && g_opt.unsorted_mode == 2 && red == GEOT_REDUCE_SUM && (dt == GEOT_F32 || dt == GEOT_F64);
  if (!f.ascending && !atomic_flush) kp = sorted_form(index, f.kmin, f.kmax);
  // rows = index[-1] + 1 is read back and verified on every call, whichever kernels serve it
  at::Tensor out = with_row_rule(index, f.rows, !f.cached, [&](int64_t rows) {
    shape[0] = rows;
    at::Tensor o = at::empty(shape, moved.options());
    auto &ws = workspace(src, geot_workspace_bytes(nnz, feat, rows, dt));
    if (f.ascending) {
      if (red == GEOT_REDUCE_SUM)
        GEOT_CALL(geot_index_scatter(index_ptr(index), moved.data_ptr(), o.data_ptr(), nnz, feat, rows, dt, 1, ws.data_ptr(), ws.numel(), stream_of(src)));
      else
        GEOT_CALL(geot_index_scatter_reduce(index_ptr(index), moved.data_ptr(), o.data_ptr(), nnz, feat, rows, dt, red, ws.data_ptr(), ws.numel(), stream_of(src)));
    } else if (atomic_flush) {
      // pre-reduced runs + float atomics into a zeroed dst (what the reference does for every flush)
      GEOT_CALL(geot_index_scatter(index_ptr(index), moved.data_ptr(), o.data_ptr(), nnz, feat, rows, dt, 0, ws.data_ptr(), ws.numel(), stream_of(src)));
    } else {
      // reduce over (sorted keys, permutation) with the gather-mode kernels: deterministic, any reduction and dtype;
      // rows stay index[-1]+1 (the reference's rule even for an unsorted index), keys beyond are ignored
      GEOT_CALL(geot_gather_reduce(index_ptr(kp.second), index_ptr(kp.first), nullptr, moved.data_ptr(), o.data_ptr(), nnz, feat, nnz, rows,
                                   dt, red, ws.data_ptr(), ws.numel(), stream_of(src)));
    }
    return o;
  });
  return dim == 0 ? out : out.movedim(0, dim);
}

// ---- index_scatter on CPU tensors -------------------------------------------------------------------------------------------
// The one operator the reference registers for the CPU key as well (csrc/index_scatter.cpp:11-24,53 ->
// csrc/cpu/index_scatter_cpu.cpp:25-155).  This is that key's kernel for CPU tensors - it is never reached by GPU
// tensors and is no fallback for them (without libgeot_hip.so the package does not import).  Semantics: the intended
// ones, dst[index[e]] (op)= src[e] (the reference reads src[index[e]]); per row strictly sequential in edge
// order with fp32 accumulation for 16-bit storage (index_scatter_cpu.cpp:78-113), rows = index[-1] + 1, rows without
// edges 0, NaN propagated by max / min as ATen does.  Rows are found without a segment table: every thread of
// at::parallel_for takes an edge range moved to row boundaries.  An index with descents (the reference refuses:
// "unsorted index is not supported yet", :151) is reduced over its stable sort, as on the GPU.
template <typename T, int RED> void cpu_reduce_rows(const int64_t *index, const T *src, const int64_t *perm, T *out, int64_t nnz, int64_t feat) {
  using acc_t = at::opmath_type<T>;
  const int64_t grain = std::max<int64_t>(1, 32768 / std::max<int64_t>(feat, 1));
  at::parallel_for(0, nnz, grain, [&](int64_t b, int64_t e) {
    while (b > 0 && b < nnz && index[b] == index[b - 1]) ++b;         // both ends move forward to the next row start:
    while (e > 0 && e < nnz && index[e] == index[e - 1]) ++e;         // neighbouring ranges meet at the same edge
    std::vector<acc_t> acc((size_t)feat);
    for (int64_t i = b; i < e;) {
      const int64_t row = index[i];
      int64_t j = i;
      const acc_t identity = RED == GEOT_REDUCE_PROD ? acc_t(1) : RED == GEOT_REDUCE_MAX ? -std::numeric_limits<acc_t>::infinity()
                             : RED == GEOT_REDUCE_MIN ? std::numeric_limits<acc_t>::infinity() : acc_t(0);
      std::fill(acc.begin(), acc.end(), identity);
      for (; j < nnz && index[j] == row; ++j) {
        const T *x = src + (perm ? perm[j] : j) * feat;
        for (int64_t k = 0; k < feat; ++k) {
          const acc_t v = static_cast<acc_t>(x[k]);
          if (RED == GEOT_REDUCE_SUM || RED == GEOT_REDUCE_MEAN) acc[k] += v;
          else if (RED == GEOT_REDUCE_PROD) acc[k] *= v;
          else if (RED == GEOT_REDUCE_MAX) acc[k] = (v != v) ? v : (acc[k] < v ? v : acc[k]);
          else acc[k] = (v != v) ? v : (v < acc[k] ? v : acc[k]);
        }
      }
      T *o = out + row * feat;
      const acc_t count = static_cast<acc_t>(j - i);
      for (int64_t k = 0; k < feat; ++k) o[k] = static_cast<T>(RED == GEOT_REDUCE_MEAN ? acc[k] / count : acc[k]);
      i = j;
    }
  });
}

template <typename T>
void cpu_reduce_dispatch(int red, const int64_t *index, const T *src, const int64_t *perm, T *out, int64_t nnz, int64_t feat) {
  switch (red) {
  case GEOT_REDUCE_SUM: return cpu_reduce_rows<T, GEOT_REDUCE_SUM>(index, src, perm, out, nnz, feat);
  case GEOT_REDUCE_MEAN: return cpu_reduce_rows<T, GEOT_REDUCE_MEAN>(index, src, perm, out, nnz, feat);
  case GEOT_REDUCE_MAX: return cpu_reduce_rows<T, GEOT_REDUCE_MAX>(index, src, perm, out, nnz, feat);
  case GEOT_REDUCE_MIN: return cpu_reduce_rows<T, GEOT_REDUCE_MIN>(index, src, perm, out, nnz, feat);
  default: return cpu_reduce_rows<T, GEOT_REDUCE_PROD>(index, src, perm, out, nnz, feat);
  }
}

at::Tensor index_scatter_cpu_op(const int64_t dim, const at::Tensor &index_in, const at::Tensor &src, const c10::string_view reduce,
                                const bool /*sorted: a hint, checked below*/) {
  TORCH_CHECK(dim >= 0 && dim < src.dim(), "dim must be non-negative and less than input dimensions");
  TORCH_CHECK(index_in.dim() == 1, "index must be 1 dimensional");
  TORCH_CHECK(src.size(dim) == index_in.size(0), "index length must be equal to src dimension size");
  const int red = reduce_code(reduce);
  TORCH_CHECK_INDEX(index_in.numel() > 0, "index -1 is out of bounds for dimension 0 with size 0");
  TORCH_CHECK(index_in.device().is_cpu() && src.device().is_cpu(), "all tensors must be on the same device");
  at::Tensor moved = (dim == 0 ? src : src.movedim(dim, 0)).contiguous();
  at::Tensor index = index_in.contiguous();
  const int64_t *ip = index_ptr(index);
  const int64_t nnz = index.numel(), feat = moved.numel() / nnz;
  const int64_t rows = ip[nnz - 1] + 1; // csrc/index_scatter.cpp:15
  TORCH_CHECK_INDEX(rows >= 0, "index out of range");
  bool ascending = ip[0] >= 0;
  for (int64_t i = 0; i + 1 < nnz && ascending; ++i) ascending = ip[i] <= ip[i + 1];
  at::Tensor keys = index, perm;
  if (!ascending) { // stable sort; rows stay index[-1] + 1, keys outside [0, rows) are ignored
    auto sorted = at::sort(index, /*stable=*/true, 0, false);
    at::Tensor k = std::get<0>(sorted), p = std::get<1>(sorted);
    at::Tensor keep = at::nonzero(k.ge(0).logical_and(k.lt(rows))).flatten();
    keys = k.index_select(0, keep).contiguous();
    perm = p.index_select(0, keep).contiguous();
  }
  auto shape = moved.sizes().vec();
  shape[0] = rows;
  at::Tensor out = at::zeros(shape, moved.options());
  const int64_t n = keys.numel();
  const int64_t *pp = perm.defined() ? perm.data_ptr<int64_t>() : nullptr;
  if (n > 0 && feat > 0) {
    switch (moved.scalar_type()) {
    case at::kFloat: cpu_reduce_dispatch<float>(red, keys.data_ptr<int64_t>(), moved.data_ptr<float>(), pp, out.data_ptr<float>(), n, feat); break;
    case at::kDouble: cpu_reduce_dispatch<double>(red, keys.data_ptr<int64_t>(), moved.data_ptr<double>(), pp, out.data_ptr<double>(), n, feat); break;
    case at::kHalf: cpu_reduce_dispatch<at::Half>(red, keys.data_ptr<int64_t>(), moved.data_ptr<at::Half>(), pp, out.data_ptr<at::Half>(), n, feat); break;
    case at::kBFloat16:
      cpu_reduce_dispatch<at::BFloat16>(red, keys.data_ptr<int64_t>(), moved.data_ptr<at::BFloat16>(), pp, out.data_ptr<at::BFloat16>(), n, feat);
      break;
    default: TORCH_CHECK(false, "\"index_scatter_sorted\" not implemented for '", toString(moved.scalar_type()), "'");
    }
  }
  return dim == 0 ? out : out.movedim(0, dim);
}

at::Tensor gather_rows_cpu_op(const at::Tensor &index, const at::Tensor &src) {
  TORCH_CHECK(index.dim() == 1 && src.dim() >= 1, "gather_rows: index must be 1 dimensional");
  return src.index_select(0, index);
}

// ---- gather ops -------------------------------------------------------------------------------------------------------------------
void check_gather(const at::Tensor &si, const at::Tensor &di, const at::Tensor &src, int64_t ndim) {
  TORCH_CHECK(si.dim() == 1 && di.dim() == 1, "src_index and dst_index must be 1 dimensional");
  TORCH_CHECK(src.dim() == ndim, "src must be ", ndim, " dimensional");
  TORCH_CHECK(si.size(0) == di.size(0), "src_index and dst_index must have the same length");
}

struct Edges {
  at::Tensor si, di, w;
  at::Tensor di_given; // the caller's dst_index (contiguous): the row rule reads ITS last element
  int64_t rows;       // index[-1] + 1 as remembered / probed
  bool fresh;         // rows was read in this very call
  bool permuted;      // di had descents: (si, di, w) are the stable sort by destination
};

// edges in ascending dst order: as given when dst_index is ascending (the reference's unchecked precondition, checked
// once per content here), else their stable sort by destination
Edges dst_ordered(const at::Tensor &si_in, const at::Tensor &di_in, const c10::optional<at::Tensor> &w_in, int64_t w_edge_dim) {
  Edges e;
  e.si = si_in.contiguous();
  e.di = di_in.contiguous();
  e.di_given = e.di;
  index_ptr(e.si);
  index_ptr(e.di);
  if (w_in.has_value() && w_in->defined()) e.w = w_in->contiguous();
  TORCH_CHECK_INDEX(e.di.numel() > 0, "index -1 is out of bounds for dimension 0 with size 0");
  const FactsView f = index_facts(e.di);
  e.rows = f.rows;
  e.fresh = !f.cached;
  e.permuted = !f.ascending;
  if (e.permuted) {
    auto kp = sorted_form(e.di, f.kmin, f.kmax);
    e.si = e.si.index_select(0, kp.second);
    if (e.w.defined()) e.w = e.w.index_select(w_edge_dim, kp.second).contiguous();
    e.di = kp.first;
  }
  return e;
}

// reduce: GEOT_REDUCE_*; weight optional; rows < 0 -> the row rule
at::Tensor gather_common(const char *op, const at::Tensor &si, const at::Tensor &di, const c10::optional<at::Tensor> &weight,
                         const at::Tensor &src, int red, int64_t rows_given) {
  check_gather(si, di, src, 2);
  const bool has_w = weight.has_value() && weight->defined();
  if (has_w) TORCH_CHECK(weight->dim() == 1 && weight->size(0) == di.size(0), "weight must be 1 dimensional with one value per edge");
  require_gpu(op, {&si, &di, &src, has_w ? &*weight : nullptr});
  const int dt = dtype_code(src, has_w ? "gather_weight_scatter_sorted" : "gather_scatter_sorted");
  if (has_w) TORCH_CHECK(weight->scalar_type() == src.scalar_type(), "expected weight of dtype ", toString(src.scalar_type()), " but found ",
                         toString(weight->scalar_type()));
  GEOT_DEVICE_GUARD(src);
  at::Tensor x = src.contiguous();
  Edges e = dst_ordered(si, di, weight, 0);
  const int64_t nnz = e.di.numel(), feat = x.size(1);
  auto launch = [&](int64_t rows) {
    at::Tensor o = at::empty({rows, feat}, x.options());
    auto run_edges = [&](at::Tensor &o) {
      auto &ws = workspace(x, geot_workspace_bytes(nnz, feat, rows, dt));
      if (red == GEOT_REDUCE_SUM && has_w)
        GEOT_CALL(geot_gather_weight_scatter(index_ptr(e.si), index_ptr(e.di), e.w.data_ptr(), x.data_ptr(), o.data_ptr(), nnz, feat, x.size(0), rows,
                                             dt, ws.data_ptr(), ws.numel(), stream_of(x)));
      else if (red == GEOT_REDUCE_SUM)
        GEOT_CALL(geot_gather_scatter(index_ptr(e.si), index_ptr(e.di), x.data_ptr(), o.data_ptr(), nnz, feat, x.size(0), rows, dt, ws.data_ptr(),
                                      ws.numel(), stream_of(x)));
      else
        GEOT_CALL(geot_gather_reduce(index_ptr(e.si), index_ptr(e.di), has_w ? e.w.data_ptr() : nullptr, x.data_ptr(), o.data_ptr(), nnz, feat,
                                     x.size(0), rows, dt, red, ws.data_ptr(), ws.numel(), stream_of(x)));
    };
    std::shared_ptr<SlabPlanHolder> plan;
    if (red != GEOT_REDUCE_PROD && !e.permuted)        // dense graphs: sum / mean / max / min on the source-blocked kernel
      plan = slab_plan_for(e.si, e.di, rows, x, has_w ? 1 : 0, 1);
    if (plan) {
      auto run_plan = [&](at::Tensor &o) -> bool {
        const void *wptr = has_w ? e.w.data_ptr() : nullptr;
        int wmode = has_w ? 1 : 0;
        at::Tensor w_planorder;                        // (keeps the permuted copy alive across the launch)
        ContentKey wk;
        const bool w_owned = has_w && owned_product(e.w); // (takes g_mu: before wmu)
        if (has_w && may_remember({&e.w}) && content_key(e.w, &wk)) {
          std::lock_guard<std::mutex> lk(plan->wmu);
          if (plan->w_planorder.defined() && plan->w_key == wk && plan->w_given && !plan->w_given->expired()) {
            w_planorder = plan->w_planorder;
            plan->w_made.before_use(x, {&w_planorder});
            guard_check(plan->w_fp, {&e.w});
          } else if (tl_capturing) {
            // (no new cache content during a capture)
          } else if (plan->w_seen_valid && plan->w_seen == wk && plan->keep.size() > 2) { // the same weight content again: permute it once
            plan->w_planorder = e.w.index_select(0, plan->keep[2]);
            plan->w_fp = w_owned ? at::Tensor() : guard_store({&e.w});
            plan->w_made.mark(x);
            plan->w_key = wk;
            plan->w_given = weak_of(e.w);              // (weak: pins the address under this key, not the data)
            w_planorder = plan->w_planorder;
          } else {
            plan->w_seen = wk;
            plan->w_seen_valid = true;
          }
          if (w_planorder.defined()) {
            wptr = w_planorder.data_ptr();
            wmode = 4;
          }
        }
        guard_flush(); // (the question about the weight, asked under wmu: launched now, in front of the plan's kernels)
        return run_slab(*plan, wptr, wmode, x, o, 1, feat, red);
      };
      return plan_or_edges(plan, 0, o, x, run_plan, run_edges);
    }
    run_edges(o);
    return o;
  };
  if (rows_given >= 0) return launch(rows_given);
  return with_row_rule(e.di_given, e.rows, e.fresh, launch);
}

at::Tensor gather_scatter_op(const at::Tensor &si, const at::Tensor &di, const at::Tensor &src) {
  return gather_common("gather_scatter", si, di, c10::nullopt, src, GEOT_REDUCE_SUM, -1);
}
at::Tensor gather_weight_scatter_op(const at::Tensor &si, const at::Tensor &di, const at::Tensor &weight, const at::Tensor &src) {
  return gather_common("gather_weight_scatter", si, di, weight, src, GEOT_REDUCE_SUM, -1);
}
at::Tensor gather_scatter_rows_op(const at::Tensor &si, const at::Tensor &di, const at::Tensor &src, int64_t rows) {
  TORCH_CHECK(rows >= 0, "rows must be non-negative");
  return gather_common("gather_scatter_rows", si, di, c10::nullopt, src, GEOT_REDUCE_SUM, rows);
}
at::Tensor gather_weight_scatter_rows_op(const at::Tensor &si, const at::Tensor &di, const at::Tensor &weight, const at::Tensor &src,
                                         int64_t rows) {
  TORCH_CHECK(rows >= 0, "rows must be non-negative");
  return gather_common("gather_weight_scatter_rows", si, di, weight, src, GEOT_REDUCE_SUM, rows);
}
// PyG call sites forward their `aggr` as the trailing reduce of the gather ops (models/conv/spmm.py:5-14)
at::Tensor gather_reduce_op(const at::Tensor &si, const at::Tensor &di, const c10::optional<at::Tensor> &weight, const at::Tensor &src,
                            const c10::string_view reduce) {
  return gather_common("gather_reduce", si, di, weight, src, reduce_code(reduce, /*pyg_add=*/true), -1);
}

at::Tensor mh_spmm_common(const at::Tensor &si, const at::Tensor &di, const at::Tensor &weight, const at::Tensor &src, int64_t rows_given,
                          bool edge_major_only) {
  check_gather(si, di, src, 3);
  const int64_t nnz = si.size(0);
  // layout pick of csrc/cuda/wrapper/mh_spmm_base.h:38-49 ([nnz, H] first, then [H, nnz])
  int layout;
  if (weight.dim() == 2 && weight.size(0) == nnz && weight.size(1) == src.size(1)) layout = GEOT_W_EDGE_MAJOR;
  else if (!edge_major_only && weight.dim() == 2 && weight.size(1) == nnz && weight.size(0) == src.size(1)) layout = GEOT_W_HEAD_MAJOR;
  else TORCH_CHECK(false, "Invalid weight size"); // csrc/cuda/wrapper/mh_spmm_base.h:49
  require_gpu("mh_spmm", {&si, &di, &weight, &src});
  const int dt = dtype_code(src, "mh_spmm_sorted");
  TORCH_CHECK(weight.scalar_type() == src.scalar_type(), "expected weight of dtype ", toString(src.scalar_type()), " but found ",
              toString(weight.scalar_type()));
  GEOT_DEVICE_GUARD(src);
  at::Tensor x = src.contiguous();
  Edges e = dst_ordered(si, di, weight, layout == GEOT_W_HEAD_MAJOR ? 1 : 0);
  const int64_t heads = x.size(1), feat = x.size(2);
  auto launch = [&](int64_t rows) {
    at::Tensor o = at::empty({rows, heads, feat}, x.options());
    auto run_edges = [&](at::Tensor &o) {
      auto &ws = workspace(x, geot_mh_workspace_bytes(nnz, heads, feat, rows, dt));
      GEOT_CALL(geot_mh_spmm(index_ptr(e.si), index_ptr(e.di), e.w.data_ptr(), x.data_ptr(), o.data_ptr(), nnz, heads, feat, x.size(0), rows, layout,
                             dt, ws.data_ptr(), ws.numel(), stream_of(x)));
    };
    if (!e.permuted && (feat * x.element_size()) % 16 == 0 && heads <= 16) {
      // the source-blocked kernel reads weights through the edge permutation: edge-major [nnz, H] is one 16-byte read per
      // edge, head-major [H, nnz] would be H scattered 4-byte reads (H x 64-byte sectors) - transpose it once instead
      // (a streaming pass, ~0.6 ms at 115 M edges x 4 heads)
      if (auto plan = slab_plan_for(e.si, e.di, rows, x, 2, heads)) {
        auto run_plan = [&](at::Tensor &o) -> bool {
          at::Tensor w_em = layout == GEOT_W_HEAD_MAJOR ? e.w.t().contiguous() : e.w;
          return run_slab(*plan, w_em.data_ptr(), 2, x, o, heads, feat);
        };
        return plan_or_edges(plan, 0, o, x, run_plan, run_edges);
      }
    }
    run_edges(o);
    return o;
  };
  if (rows_given >= 0) return launch(rows_given);
  return with_row_rule(e.di_given, e.rows, e.fresh, launch);
}

at::Tensor mh_spmm_op(const at::Tensor &si, const at::Tensor &di, const at::Tensor &weight, const at::Tensor &src, const c10::string_view reduce) {
  check_gather(si, di, src, 3);
  TORCH_CHECK_NOT_IMPLEMENTED(reduce_code(reduce) == GEOT_REDUCE_SUM, "mh_spmm: reduce='", reduce,
                              "' is not implemented on the HIP path (only 'sum').  Note the reference's GPU kernels ignore `reduce` and always sum.");
  return mh_spmm_common(si, di, weight, src, -1, false);
}
at::Tensor mh_spmm_rows_op(const at::Tensor &si, const at::Tensor &di, const at::Tensor &weight, const at::Tensor &src, int64_t rows) {
  TORCH_CHECK(rows >= 0, "rows must be non-negative");
  return mh_spmm_common(si, di, weight, src, rows, true);
}

at::Tensor sddmm_coo_op(const at::Tensor &si_in, const at::Tensor &di_in, const at::Tensor &m1_in, const at::Tensor &m2_in) {
  TORCH_CHECK(m1_in.dim() == 2 && m2_in.dim() == 2 && m1_in.size(1) == m2_in.size(1),
              "mat_1 and mat_2 must be 2 dimensional with the same feature dimension");
  require_gpu("sddmm_coo_impl", {&si_in, &di_in, &m1_in, &m2_in});
  GEOT_DEVICE_GUARD(m1_in);
  // the reference's Python wrapper hands over int32 indices (geot/gather_weight_scatter.py:10-11): both widths accepted
  at::Tensor si = as_int64(si_in), di = as_int64(di_in);
  at::Tensor m1 = m1_in.contiguous(), m2 = m2_in.contiguous();
  at::Tensor out = at::empty({di.size(0)}, m1.options());
  auto run_edges = [&](at::Tensor &o) {
    GEOT_CALL(geot_sddmm_coo(index_ptr(si), index_ptr(di), m1.data_ptr(), m2.data_ptr(), o.data_ptr(), di.size(0), m1.size(1), m1.size(0),
                             m2.size(0), dtype_code(m1, "sddmm_coo"), stream_of(m1)));
  };
  if (m1.scalar_type() != at::kDouble && m1.scalar_type() == m2.scalar_type() && di.numel() > 0 && m1.size(0) < ((int64_t)1 << 31)) {
    // a dense graph that has (or now earns) a source-blocked plan - the forward gather_weight_scatter's - and an
    // ascending dst_index (known from the facts): SDDMM over the plan, the gathered m2 rows re-used out of L2
    const int64_t rowbytes = m1.size(1) * m1.element_size();
    if ((rowbytes == 256 || rowbytes == 512 || rowbytes == 1024) &&
        (g_opt.slab_mode == 1 || (g_opt.slab_mode == 0 && slab_worthwhile(di.numel(), m1.size(0), m2.size(0), rowbytes, dtype_code(m1, "sddmm_coo")))) &&
        (tl_capturing || index_facts(di).ascending)) { // (under capture only an existing plan is used: built on an ascending di)
      if (auto plan = slab_plan_for(si, di, m1.size(0), m2, 1, 1)) {
        auto run_plan = [&](at::Tensor &o) -> bool {
          const std::vector<at::Tensor> pinned = plan->pinned();
          if (pinned.empty()) return false;
          auto &ws = workspace(m1, geot_slab_workspace_bytes(&plan->plan, m1.size(1)));
          GEOT_CALL(geot_slab_sddmm(&plan->plan, m1.data_ptr(), m2.data_ptr(), o.data_ptr(), m1.size(1), m1.size(0), m2.size(0),
                                    dtype_code(m1, "sddmm_coo"), ws.data_ptr(), ws.numel(), stream_of(m1)));
          plan->launched_on(m1, pinned);
          return true;
        };
        return plan_or_edges(plan, 1, out, m1, run_plan, run_edges);
      }
    }
  }
  run_edges(out);
  return out;
}

// CSR row pointers -> per-edge row ids, once per indptr content (the COO form is what every kernel consumes; with it a
// CSR call shares the index facts, the row-count handling and the source-blocked path of the COO ops)
struct ExpandedEntry {
  ContentKey key;
  WeakStorage indptr; // the caller's row pointers (weak)
  at::Tensor dst_index;
  Produced made;
  at::Tensor fp; // fingerprint of the row pointers (guard_store)
};
std::list<ExpandedEntry> g_expanded;

at::Tensor expand_indptr(const at::Tensor &indptr, int64_t nnz) {
  const GuardFlush flush_questions_;
  ContentKey k;
  const bool keyed = may_remember({&indptr}) && content_key(indptr, &k);
  if (keyed) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto it = g_expanded.begin(); it != g_expanded.end(); ++it)
      if (it->key == k && it->dst_index.numel() == nnz && !it->indptr.expired()) {
        g_expanded.splice(g_expanded.begin(), g_expanded, it);
        g_expanded.front().made.before_use(indptr, {&g_expanded.front().dst_index});
        guard_check(g_expanded.front().fp, {&indptr});
        return g_expanded.front().dst_index;
      }
  }
  const int64_t nrow = indptr.numel() - 1;
  at::Tensor counts = (indptr.slice(0, 1, nrow + 1) - indptr.slice(0, 0, nrow)).clamp_min(0);
  at::Tensor dst_index = at::repeat_interleave(counts, c10::optional<int64_t>(nnz));
  if (keyed && !tl_capturing) {
    at::Tensor fp = guard_store({&indptr});
    std::lock_guard<std::mutex> lk(g_mu);
    g_expanded.push_front(ExpandedEntry{k, weak_of(indptr), dst_index, {}, fp});
    g_expanded.front().made.mark(indptr);
    while (g_expanded.size() > 4) g_expanded.pop_back();
    enforce_cache_budget_locked();
  }
  return dst_index;
}

// csrc/csr_gws.cpp:24-35: any integer dtype for indptr / indices; indptr.size(0) output rows (the last one always zero)
at::Tensor csr_gws_op(const at::Tensor &indptr_in, const at::Tensor &indices_in, const at::Tensor &weight_in, const at::Tensor &src_in) {
  TORCH_CHECK(indptr_in.dim() == 1 && indices_in.dim() == 1, "indptr and indices must be 1 dimensional");
  TORCH_CHECK(src_in.dim() == 2, "src must be 2 dimensional");
  TORCH_CHECK(weight_in.dim() == 1 && weight_in.size(0) == indices_in.size(0), "weight must be 1 dimensional with one value per nonzero");
  require_gpu("csr_gws_impl", {&indptr_in, &indices_in, &weight_in, &src_in});
  TORCH_CHECK(weight_in.scalar_type() == src_in.scalar_type(), "expected weight of dtype ", toString(src_in.scalar_type()), " but found ",
              toString(weight_in.scalar_type()));
  GEOT_DEVICE_GUARD(src_in);
  at::Tensor indptr = as_int64(indptr_in), indices = as_int64(indices_in);
  if (g_opt.slab_mode >= 0 && indices.numel() > 0 && indptr.numel() >= 2 && src_in.scalar_type() != at::kDouble) {
    // a graph dense enough for the source-blocked kernel: go through the COO path (row ids expanded once per indptr)
    const int64_t rowbytes = src_in.size(1) * src_in.element_size();
    if (g_opt.slab_mode == 1 || slab_worthwhile(indices.numel(), indptr.size(0), src_in.size(0), rowbytes, dtype_code(src_in, "csr_gws")))
      return gather_common("csr_gws_impl", indices, expand_indptr(indptr, indices.numel()), weight_in, src_in, GEOT_REDUCE_SUM, indptr.size(0));
  }
  at::Tensor weight = weight_in.contiguous(), src = src_in.contiguous();
  const int64_t rows = indptr.size(0), nnz = indices.size(0), feat = src.size(1);
  at::Tensor out = at::empty({rows, feat}, src.options());
  const int dt = dtype_code(src, "csr_gws");
  auto &ws = workspace(src, geot_csr_workspace_bytes(nnz, feat, rows, dt));
  GEOT_CALL(geot_csr_gws(index_ptr(indptr), index_ptr(indices), weight.data_ptr(), src.data_ptr(), out.data_ptr(), rows - 1, nnz, feat, src.size(0),
                         rows, dt, ws.data_ptr(), ws.numel(), stream_of(src)));
  return out;
}

// backward of index_scatter: d/dsrc[e] = grad[index[e]] (the reference ships gather_eb_sorted_kernel,
// csrc/cuda/index_scatter_kernel.cuh:266-315, but never wires it to an op)
at::Tensor gather_rows_op(const at::Tensor &index_in, const at::Tensor &src_in) {
  TORCH_CHECK(index_in.dim() == 1 && src_in.dim() >= 1, "gather_rows: index must be 1 dimensional");
  require_gpu("gather_rows", {&index_in, &src_in});
  GEOT_DEVICE_GUARD(src_in);
  at::Tensor index = index_in.contiguous(), src = src_in.contiguous();
  auto shape = src.sizes().vec();
  shape[0] = index.size(0);
  at::Tensor out = at::empty(shape, src.options());
  const int64_t feat = src.numel() / std::max<int64_t>(src.size(0), 1);
  GEOT_CALL(geot_gather_rows(index_ptr(index), src.data_ptr(), out.data_ptr(), index.size(0), feat, src.size(0), dtype_code(src, "gather_rows"),
                             stream_of(src)));
  return out;
}

// The backward of the gather ops needs the edge list sorted by SOURCE (the transposed graph).  The reference re-sorts on
// every backward call (geot/gather_scatter.py:30-33); graphs are static, so it is kept per edge-list content.  The entry
// keeps its key tensors alive: a freed edge list's address can never be handed to a new one while the entry lives.
struct TransposedEntry {
  ContentKey k1, k2;
  WeakStorage w1, w2; // the edge list (weak)
  at::Tensor perm, si_sorted, di_perm;
  // the per-edge weight in transposed order, kept for the content it was made from (a static weight - a normalised
  // adjacency that does not require grad - is permuted once, not on every backward call)
  bool w_valid = false;
  ContentKey wkey{};
  c10::optional<WeakStorage> w_given;
  at::Tensor w_perm;
  Produced made, w_made;
  at::Tensor fp, w_fp; // fingerprints of the edge list / of the weight (guard_store)
  int64_t bytes() const { return nbytes_of(perm) + nbytes_of(si_sorted) + nbytes_of(di_perm) + nbytes_of(w_perm); }
};
std::list<TransposedEntry> g_transposed;

bool owned_product(const at::Tensor &t) {
  if (!t.defined() || !t.has_storage()) return false;
  const void *st = t.storage().unsafeGetStorageImpl();
  std::lock_guard<std::mutex> lk(g_mu);
  for (const TransposedEntry &e : g_transposed)
    for (const at::Tensor *p : {&e.perm, &e.si_sorted, &e.di_perm, &e.w_perm})
      if (p->defined() && p->storage().unsafeGetStorageImpl() == st) return true;
  return false;
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> transpose_edges_op(const at::Tensor &si, const at::Tensor &di) {
  const GuardFlush flush_questions_;
  require_gpu("transpose_edges", {&si, &di});
  GEOT_DEVICE_GUARD(si);
  ContentKey k1, k2;
  const bool keyed = g_opt.transpose_cache > 0 && may_remember({&si, &di}) && content_key(si, &k1) && content_key(di, &k2);
  if (keyed) {
    std::lock_guard<std::mutex> lk(g_mu);
    sweep_expired_locked();
    for (auto it = g_transposed.begin(); it != g_transposed.end(); ++it)
      if (it->k1 == k1 && it->k2 == k2) {
        g_transposed.splice(g_transposed.begin(), g_transposed, it);
        it->made.before_use(si, {&it->perm, &it->si_sorted, &it->di_perm});
        guard_check(it->fp, {&si, &di});
        return {it->perm, it->si_sorted, it->di_perm};
      }
  }
  at::Tensor sic = si.contiguous();
  index_ptr(sic);
  int64_t p[4] = {0, 0, -1, -1};                                         // (min -1: the generic sort)
  if (sic.numel() > 0 && !tl_capturing) probe_index(sic, p);              // the key range sizes the sort
  auto sorted = sic.numel() > 0 ? stable_sort_index(sic, p[2], p[3]) : std::make_pair(sic, sic);
  at::Tensor perm = sorted.second, di_perm = di.index_select(0, perm);
  at::Tensor fp = keyed && !tl_capturing ? guard_store({&si, &di}) : at::Tensor();
  std::lock_guard<std::mutex> lk(g_mu);
  ++g_stats.transposes;
  if (keyed && !tl_capturing) {
    g_transposed.push_front(TransposedEntry{k1, k2, weak_of(si), weak_of(di), perm, sorted.first, di_perm});
    g_transposed.front().fp = fp;
    g_transposed.front().made.mark(si);
    while ((int)g_transposed.size() > g_opt.transpose_cache) g_transposed.pop_back();
    enforce_cache_budget_locked();
  }
  return {perm, sorted.first, di_perm};
}

at::Tensor transposed_weight_op(const at::Tensor &si, const at::Tensor &di, const at::Tensor &weight) {
  const GuardFlush flush_questions_;
  require_gpu("transposed_weight", {&si, &di, &weight});
  TORCH_CHECK(weight.dim() >= 1 && weight.size(0) == si.size(0), "weight must have one entry per edge");
  GEOT_DEVICE_GUARD(si);
  auto tr = transpose_edges_op(si, di);           // (cached) permutation by source
  const at::Tensor &perm = std::get<0>(tr);
  ContentKey k1, k2, wk;
  const bool keyed = g_opt.transpose_cache > 0 && may_remember({&si, &di, &weight}) && content_key(si, &k1) && content_key(di, &k2) &&
                     content_key(weight, &wk) && !weight.requires_grad();
  if (keyed) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &e : g_transposed)
      if (e.k1 == k1 && e.k2 == k2 && e.w_valid && e.wkey == wk && e.w_given && !e.w_given->expired()) {
        e.w_made.before_use(weight, {&e.w_perm});
        guard_check(e.w_fp, {&weight});
        return e.w_perm;
      }
  }
  at::Tensor wp = weight.index_select(0, perm);
  if (keyed && !tl_capturing) {
    at::Tensor wfp = guard_store({&weight});
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &e : g_transposed)
      if (e.k1 == k1 && e.k2 == k2) {
        e.w_valid = true;
        e.wkey = wk;
        e.w_given = weak_of(weight);
        e.w_perm = wp;
        e.w_fp = wfp;
        e.w_made.mark(weight);
      }
  }
  return wp;
}

// both lookups of a weighted backward pass in ONE operator call (one read of the edge list for the content guard, one wait)
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> transpose_edges_weighted_op(const at::Tensor &si, const at::Tensor &di,
                                                                                       const at::Tensor &weight) {
  auto tr = transpose_edges_op(si, di);
  at::Tensor wp = transposed_weight_op(si, di, weight);
  return {std::get<0>(tr), std::get<1>(tr), std::get<2>(tr), wp};
}

void sweep_expired_locked() {
  g_widened.remove_if([](const WidenedEntry &e) { return e.narrow.expired(); });
  g_expanded.remove_if([](const ExpandedEntry &e) { return e.indptr.expired(); });
  g_transposed.remove_if([](const TransposedEntry &e) { return e.w1.expired() || e.w2.expired(); });
  g_slab.remove_if([](const SlabEntry &e) { return e.w1.expired() || e.w2.expired(); });
  for (auto &f : g_facts)
    if (f.weak.expired()) f.keys = f.perm = at::Tensor();
}

int64_t cache_bytes_locked() {
  int64_t b = 0;
  for (auto &f : g_facts) b += nbytes_of(f.keys) + nbytes_of(f.perm);
  for (auto &e : g_widened) b += nbytes_of(e.wide);
  for (auto &e : g_expanded) b += nbytes_of(e.dst_index);
  for (auto &e : g_transposed) b += e.bytes();
  for (auto &e : g_slab) b += e.plan->bytes();
  return b;
}

int64_t cache_budget_bytes() {
  if (g_opt.cache_mb > 0) return g_opt.cache_mb << 20;
  static const int64_t def = [] {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) {
      (void)hipGetLastError();
      return (int64_t)32 << 30;
    }
    return (int64_t)(total_b / 8);
  }();
  return def;
}

// all cached artefacts together stay under the budget: dead sources first, then the largest least-recently-used entry of any
// cache (the newest entry of each cache is the one just made or used: it goes last)
void enforce_cache_budget_locked() {
  sweep_expired_locked();
  const int64_t budget = cache_budget_bytes();
  for (int guard = 0; guard < 64 && cache_bytes_locked() > budget; ++guard) {
    int64_t best = 0;
    int which = -1;
    auto consider = [&](int id, int64_t b, size_t n) {
      if (n > 1 && b > best) { best = b; which = id; }
    };
    if (!g_widened.empty()) consider(0, nbytes_of(g_widened.back().wide), g_widened.size());
    if (!g_expanded.empty()) consider(1, nbytes_of(g_expanded.back().dst_index), g_expanded.size());
    if (!g_transposed.empty()) consider(2, g_transposed.back().bytes(), g_transposed.size());
    if (!g_slab.empty()) consider(3, g_slab.back().plan->bytes(), g_slab.size());
    Facts *oldest_sorted = nullptr;
    size_t sorted_holders = 0;
    for (auto &f : g_facts)
      if (f.keys.defined()) { oldest_sorted = &f; ++sorted_holders; }
    if (oldest_sorted) consider(4, nbytes_of(oldest_sorted->keys) + nbytes_of(oldest_sorted->perm), sorted_holders);
    if (which < 0) { // one entry per cache left: drop the largest of those too, whatever it is
      auto any = [&](int id, int64_t b) { if (b > best) { best = b; which = id; } };
      if (!g_widened.empty()) any(0, nbytes_of(g_widened.back().wide));
      if (!g_expanded.empty()) any(1, nbytes_of(g_expanded.back().dst_index));
      if (!g_transposed.empty()) any(2, g_transposed.back().bytes());
      if (!g_slab.empty()) any(3, g_slab.back().plan->bytes());
      if (oldest_sorted) any(4, nbytes_of(oldest_sorted->keys) + nbytes_of(oldest_sorted->perm));
      if (which < 0) break;
    }
    switch (which) {
    case 0: g_widened.pop_back(); break;
    case 1: g_expanded.pop_back(); break;
    case 2: g_transposed.pop_back(); break;
    case 3: g_slab.pop_back(); break;
    default: oldest_sorted->keys = oldest_sorted->perm = at::Tensor(); break;
    }
  }
}

void clear_all_caches_locked() {
  g_facts.clear();
  g_transposed.clear();
  g_slab.clear();
  g_sightings.clear();
  g_widened.clear();
  g_expanded.clear();
}

bool guard_settle() {
  try {
    guard_flush(); // (normally empty: every lookup flushed its own question)
  } catch (...) {
    tl_requests.clear();
    tl_guard_tripped = true;
  }
  const size_t asked = tl_asked.size();
  tl_asked.clear();
  guard_drain();
  const bool tripped = tl_guard_tripped;
  tl_guard_tripped = false;
  if (asked || tripped) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_stats.guard_checks += (int64_t)asked;
    if (tripped) {
      clear_all_caches_locked();
      ++g_stats.stale_products;
    }
  }
  return !tripped;
}

// every registered operator goes through this: run, look at the verdicts of the products the call used, and when one of
// them described bytes that are gone, run once more (every cache is empty by then: the call derives what it needs afresh)
template <auto Fn> struct Guarded;
template <typename R, typename... A, R (*Fn)(A...)> struct Guarded<Fn> {
  static R call(A... a) {
    try {
      R out = Fn(a...);
      if (guard_settle()) return out;
    } catch (...) {
      (void)guard_settle();
      throw;
    }
    TORCH_WARN("geot: a tensor this call depends on (an index, row pointers or a static edge weight) was written behind its version "
               "counter (.data, DLPack, a raw pointer): what had been derived from its earlier content was dropped and the call was run "
               "again from the bytes it holds now.");
    try {
      R out = Fn(a...);
      (void)guard_settle();
      return out;
    } catch (...) {
      (void)guard_settle();
      throw;
    }
  }
};
#define GUARDED(fn) (&Guarded<&fn>::call)

// ---- introspection for tests / tools ------------------------------------------------------------------------------------------
int64_t host_option_op(c10::string_view name, int64_t value) {
  std::lock_guard<std::mutex> lk(g_mu);
  int *p = nullptr;
  if (name == "speculate_rows") p = &g_opt.speculate_rows;
  else if (name == "trust_version") p = &g_opt.trust_version;
  else if (name == "unsorted_mode") p = &g_opt.unsorted_mode;
  else if (name == "slab_mode") p = &g_opt.slab_mode;
  else if (name == "transpose_cache") p = &g_opt.transpose_cache;
  else if (name == "slab_keep") p = &g_opt.slab_keep;
  else if (name == "publish_rows") p = &g_opt.publish_rows;
  else if (name == "slab_builder") p = &g_opt.slab_builder;
  else if (name == "content_guard") p = &g_opt.content_guard;
  else if (name == "guard_side_stream") p = &g_opt.guard_side_stream;
  else if (name == "clear_caches") {
    clear_all_caches_locked();
    return 0;
  } else if (name == "cache_mb") {
    const int64_t old = g_opt.cache_mb;
    if (value != INT64_MIN) {
      g_opt.cache_mb = value;
      enforce_cache_budget_locked();
    }
    return old;
  }
  TORCH_CHECK(p, "unknown host option ", name);
  const int old = *p;
  if (value != INT64_MIN) *p = (int)value;
  if (name == "transpose_cache")
    while ((int)g_transposed.size() > std::max(g_opt.transpose_cache, 0)) g_transposed.pop_back();
  return old;
}

std::vector<int64_t> host_stats_op() {
  std::lock_guard<std::mutex> lk(g_mu);
  sweep_expired_locked();
  return {g_stats.probes, g_stats.row_mismatches, g_stats.sorts, g_stats.transposes, g_stats.plans_built, g_stats.slab_calls, g_stats.plan_us,
          (int64_t)g_facts.size(), (int64_t)g_transposed.size(), (int64_t)g_slab.size(), g_stats.published, g_stats.alarms,
          cache_bytes_locked(), g_stats.stale_products, g_stats.guard_checks, g_stats.plan_trials, g_stats.plans_rejected,
          g_stats.trial_plan_us, g_stats.trial_edges_us};
}

// Phase A of the source-blocked kernel as an op (works on CPU tensors too: the tests emulate the kernel on its output).
// Returns [e_src, e_dl, e_perm, g_begin, g_vrow0, g_nv, v_out, c_row, c_first, c_count, scalars(int64[12])]
std::vector<at::Tensor> slab_plan_op(const at::Tensor &si, const at::Tensor &di, int64_t rows, int64_t src_rows, int64_t rowbytes,
                                     int64_t weight_mode, int64_t heads, int64_t slab_bytes, int64_t rows_per_group, int64_t units) {
  TORCH_CHECK(si.dim() == 1 && di.dim() == 1 && si.numel() == di.numel(), "slab_plan: 1-D edge lists of equal length");
  auto H = slab_build(si.contiguous(), di.contiguous(), rows, src_rows, rowbytes, (int)weight_mode, heads,
                      slab_bytes > 0 ? slab_bytes : kSlabBytes, rows_per_group, units);
  std::vector<at::Tensor> out = H->keep;
  at::Tensor sc = at::empty({13}, at::TensorOptions().dtype(at::kLong));
  int64_t *s = sc.data_ptr<int64_t>();
  s[0] = H->plan.n_groups; s[1] = H->plan.n_vrows; s[2] = H->plan.n_carry; s[3] = H->plan.n_split; s[4] = H->plan.nnz;
  s[5] = H->plan.units; s[6] = H->plan.rows_per_group; s[7] = H->rounds; s[8] = H->budget; s[9] = H->cap; s[10] = H->slabs; s[11] = H->slab_rows; s[12] = H->plan.slab_shift;
  out.push_back(sc);
  return out;
}

bool slab_worthwhile_op(int64_t nnz, int64_t rows, int64_t src_rows, int64_t rowbytes) { return slab_worthwhile(nnz, rows, src_rows, rowbytes, GEOT_F32); }

} // namespace

// WHAT THIS PLUGIN DEFINES.  Exactly the operators the reference's csrc/*.cpp define - index_scatter
// (csrc/index_scatter.cpp:43-47), gather_scatter_impl (csrc/gather_scatter.cpp:16-17), gather_weight_scatter_impl and
// sddmm_coo_impl (csrc/gather_weight_scatter.cpp:12-16), csr_gws_impl (csrc/csr_gws.cpp:12-13), mh_spmm (a catch-all
// def in the reference, csrc/mh_spmm.cpp:23; named here) - plus helpers under names the reference does not use.
// geot::gather_scatter / gather_weight_scatter / csr_gws are NOT defined here: the reference defines them in Python
// (torch.library.custom_op, geot/gather_scatter.py:7, gather_weight_scatter.py:15, csr_gws.py:25), so a definition in
// this library would collide with the reference's unmodified wrappers ("Tried to register an operator ... multiple
// times").  Whichever Python layer sits on top defines them - geot_amd/ops.py, or the reference's own files - and this
// library only IMPLEMENTS them for the device keys (TORCH_LIBRARY_IMPL below; an impl may precede or follow its def):
// GPU tensors then reach the C++ kernels straight from the dispatcher, no Python hop, under either Python layer.
TORCH_LIBRARY_FRAGMENT(geot, m) {
  m.def("index_scatter(int dim, Tensor index, Tensor src, str reduce, bool sorted) -> Tensor");
  m.def("gather_scatter_impl(Tensor src_index, Tensor dst_index, Tensor src) -> Tensor");
  m.def("gather_weight_scatter_impl(Tensor src_index, Tensor dst_index, Tensor weight, Tensor src) -> Tensor");
  m.def("sddmm_coo_impl(Tensor src_index, Tensor dst_index, Tensor mat_1, Tensor mat_2) -> Tensor");
  m.def("csr_gws_impl(Tensor indptr, Tensor indices, Tensor weight, Tensor src) -> Tensor");
  m.def("mh_spmm(Tensor src_index, Tensor dst_index, Tensor weight, Tensor src, str reduce) -> Tensor");
  // helpers (names the reference does not use)
  m.def("gather_reduce(Tensor src_index, Tensor dst_index, Tensor? weight, Tensor src, str reduce) -> Tensor");
  m.def("gather_scatter_rows(Tensor src_index, Tensor dst_index, Tensor src, SymInt rows) -> Tensor");
  m.def("gather_weight_scatter_rows(Tensor src_index, Tensor dst_index, Tensor weight, Tensor src, SymInt rows) -> Tensor");
  m.def("mh_spmm_rows(Tensor src_index, Tensor dst_index, Tensor weight, Tensor src, SymInt rows) -> Tensor");
  m.def("gather_rows(Tensor index, Tensor src) -> Tensor");
  m.def("transpose_edges(Tensor src_index, Tensor dst_index) -> (Tensor, Tensor, Tensor)");
  m.def("transposed_weight(Tensor src_index, Tensor dst_index, Tensor weight) -> Tensor");
  m.def("transpose_edges_weighted(Tensor src_index, Tensor dst_index, Tensor weight) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("_host_option(str name, int value) -> int", host_option_op);
  m.def("_host_stats() -> int[]", host_stats_op);
  m.def("_slab_plan(Tensor src_index, Tensor dst_index, int rows, int src_rows, int rowbytes, int weight_mode, int heads, int slab_bytes, "
        "int rows_per_group, int units) -> Tensor[]", slab_plan_op);
  m.def("_slab_worthwhile(int nnz, int rows, int src_rows, int rowbytes) -> bool", slab_worthwhile_op);
}

// The GPU key ("CUDA" is what a ROCm build of PyTorch calls it).
#define GEOT_IMPLS(m)                                                                \
  m.impl("index_scatter", GUARDED(index_scatter_op));                                         \
  m.impl("gather_scatter_impl", GUARDED(gather_scatter_op));                                  \
  m.impl("gather_weight_scatter_impl", GUARDED(gather_weight_scatter_op));                    \
  m.impl("sddmm_coo_impl", GUARDED(sddmm_coo_op));                                            \
  m.impl("csr_gws_impl", GUARDED(csr_gws_op));                                                \
  m.impl("mh_spmm", GUARDED(mh_spmm_op));                                                     \
  m.impl("gather_scatter", GUARDED(gather_scatter_op));                                       \
  m.impl("gather_weight_scatter", GUARDED(gather_weight_scatter_op));                         \
  m.impl("csr_gws", GUARDED(csr_gws_op));                                                     \
  m.impl("gather_reduce", GUARDED(gather_reduce_op));                                         \
  m.impl("gather_scatter_rows", GUARDED(gather_scatter_rows_op));                             \
  m.impl("gather_weight_scatter_rows", GUARDED(gather_weight_scatter_rows_op));               \
  m.impl("mh_spmm_rows", GUARDED(mh_spmm_rows_op));                                           \
  m.impl("gather_rows", GUARDED(gather_rows_op));                                             \
  m.impl("transpose_edges", GUARDED(transpose_edges_op));                                     \
  m.impl("transposed_weight", GUARDED(transposed_weight_op));                               \
  m.impl("transpose_edges_weighted", GUARDED(transpose_edges_weighted_op))

TORCH_LIBRARY_IMPL(geot, CUDA, m) { GEOT_IMPLS(m); }
#undef GEOT_IMPLS
// CPU key: index_scatter computes (the reference registers a CPU kernel for it and for nothing else); the other
// operators run their argument checks (reference texts) and then refuse CPU tensors, as the reference has no CPU kernel
TORCH_LIBRARY_IMPL(geot, CPU, m) {
  m.impl("index_scatter", index_scatter_cpu_op);
  m.impl("gather_scatter_impl", gather_scatter_op);
  m.impl("gather_weight_scatter_impl", gather_weight_scatter_op);
  m.impl("sddmm_coo_impl", sddmm_coo_op);
  m.impl("csr_gws_impl", csr_gws_op);
  m.impl("mh_spmm", mh_spmm_op);
  m.impl("gather_scatter", gather_scatter_op);
  m.impl("gather_weight_scatter", gather_weight_scatter_op);
  m.impl("csr_gws", csr_gws_op);
  m.impl("gather_reduce", gather_reduce_op);
  m.impl("gather_scatter_rows", gather_scatter_rows_op);
  m.impl("gather_weight_scatter_rows", gather_weight_scatter_rows_op);
  m.impl("mh_spmm_rows", mh_spmm_rows_op);
  m.impl("gather_rows", gather_rows_cpu_op); // (the backward of the CPU index_scatter)
  m.impl("transpose_edges", transpose_edges_op);
  m.impl("transposed_weight", transposed_weight_op);
  m.impl("transpose_edges_weighted", transpose_edges_weighted_op);
}
