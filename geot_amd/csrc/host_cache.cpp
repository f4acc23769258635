// host_cache.cpp -- what the host layer remembers about the caller's index tensors, and how it stays true (see host.h).
#include "host.h"

namespace geot_host {

at::Tensor take_rows(const at::Tensor &values, const at::Tensor &index) {
  const auto st = values.scalar_type();
  const bool lib = values.is_cuda() && values.is_contiguous() && values.dim() >= 1 && values.size(0) > 0 && index.is_cuda() && index.dim() == 1 &&
                   index.scalar_type() == at::kLong && index.is_contiguous() &&
                   (st == at::kFloat || st == at::kDouble || st == at::kHalf || st == at::kBFloat16);
  if (!lib) return values.index_select(0, index);
  auto shape = values.sizes().vec();
  shape[0] = index.numel();
  at::Tensor out = at::empty(shape, values.options());
  if (index.numel() == 0) return out;
  const int64_t feat = values.numel() / values.size(0);
  const int rc = geot_gather_rows(index.data_ptr<int64_t>(), values.data_ptr(), out.data_ptr(), index.numel(), feat, values.size(0),
                                  dtype_code(values, "take_rows"), stream_of(values));
  TORCH_CHECK(rc == GEOT_OK, "geot_gather_rows failed (code ", rc, "): ", geot_last_error());
  return out;
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

// The fingerprint kernels go on the CALL'S stream, in front of its own kernels (the lookups come first).  Running them on a side
// stream BESIDE the call's kernels was measured and rejected (round 4, profiles/r04/bench_content_guard_side_stream.txt): a
// bandwidth-bound read beside bandwidth-bound kernels saves nothing (gws forward + backward at 40 M edges: +14.8 % either way),
// and beside the persistent source-blocked kernel it breaks the lockstep (configs[3]: 8.0 -> 14.4 ms).
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
    void *stream = stream_of(first);
    launch_fingerprint(r.ts, r.fp, true, slot, seq, stream);
    tl_pending.push_back(PendingVerdict{slot, seq, stream});
  }
}
GuardFlush::~GuardFlush() { // (declared FIRST in a function that looks products up: runs after the function's lock_guards')
  try {
    guard_flush();
  } catch (...) {
    tl_requests.clear();
    tl_guard_tripped = true; // (a question that could not be asked counts as a changed content: the call is repeated from the caller's bytes)
  }
}

void clear_all_caches_locked();
// true: every product this operator call used was made from the bytes the tensors hold now
bool guard_settle();
// is `t` itself one of the remembered products (the transposed edge list handed to the backward pass)?  What is derived from
// those - the plan of the transposed graph, its weight in plan order - needs no fingerprint of its own: nobody but this file
// writes them, and the entry they belong to is checked against the caller's tensors in the same backward pass.
bool owned_product(const at::Tensor &t);


// ---- facts of an index tensor, keyed on its content identity ---------------------------------------------------------------
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
// (the caches are never destroyed: at process exit their tensors would be freed by static destructors, after graphs / streams /
//  the allocator may already be half gone - an exit-time SIGSEGV was seen with captured graphs alive; the OS takes the memory back)
std::list<Facts> &g_facts = *new std::list<Facts>; // most recent first, <= 16 entries
constexpr size_t kFactsMax = 16, kSortedKeep = 4;


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
std::list<WidenedEntry> &g_widened = *new std::list<WidenedEntry>;

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


// CSR row pointers -> per-edge row ids, once per indptr content (the COO form is what every kernel consumes; with it a
// CSR call shares the index facts, the row-count handling and the source-blocked path of the COO ops)
struct ExpandedEntry {
  ContentKey key;
  WeakStorage indptr; // the caller's row pointers (weak)
  at::Tensor dst_index;
  Produced made;
  at::Tensor fp; // fingerprint of the row pointers (guard_store)
};
std::list<ExpandedEntry> &g_expanded = *new std::list<ExpandedEntry>;

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
std::list<TransposedEntry> &g_transposed = *new std::list<TransposedEntry>;

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
  at::Tensor wp = take_rows(weight, perm);
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
  g_declined.clear();
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
  else if (name == "slab_min_coverage_pct") p = &g_opt.slab_min_coverage_pct;
  else if (name == "slab_bytes") p = &g_opt.slab_bytes;
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
          g_stats.trial_plan_us, g_stats.trial_edges_us, g_stats.plans_declined, g_stats.last_coverage_permille};
}


} // namespace geot_host
