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
#include "host.h"

namespace {
using namespace geot_host;

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
      struct Events {                                               // (RAII: a failure part-way through creates no leak, every exit destroys)
        hipEvent_t e[kEv] = {};
        ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
      } evs;
      hipEvent_t *ev = evs.e;
      for (int i = 0; i < kEv; ++i) TORCH_CHECK(hipEventCreate(&ev[i]) == hipSuccess, "hipEventCreate failed");
      at::Tensor o2 = at::empty_like(o);
      {
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
    const at::Tensor ws = workspace(src, geot_workspace_bytes(nnz, feat, rows, dt));
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
    if (e.w.defined()) e.w = w_edge_dim == 0 ? take_rows(e.w.contiguous(), kp.second.contiguous()) : e.w.index_select(w_edge_dim, kp.second).contiguous();
    e.di = kp.first;
  }
  return e;
}

// a per-edge weight ([nnz] or [nnz, heads], contiguous) in the plan's edge order: the library's own gather (int32 plan positions;
// torch's index_select with an int32 index returned garbage for [nnz, 4] rows from 70 M edges on - found by the configs[3] full-size
// test the day the host layer began to keep multi-head weights in plan order); shapes the library does not take go through int64 indices
at::Tensor to_plan_order(SlabPlanHolder &plan, const at::Tensor &w, int64_t heads, const at::Tensor &on) {
  at::Tensor out = at::empty_like(w);
  const int rc = geot_slab_to_plan_order(&plan.plan, w.data_ptr(), out.data_ptr(), heads, dtype_code(w, "slab"), stream_of(on));
  if (rc == GEOT_EUNSUPPORTED) return w.index_select(0, plan.keep[2].to(at::kLong));
  TORCH_CHECK(rc == GEOT_OK, "geot_slab_to_plan_order failed (code ", rc, "): ", geot_last_error());
  return out;
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
  // Rows that are not whole 16-byte vectors (Reddit's raw features: F = 602; Cora 1433; F = 130) run the ragged-lane kernels
  // (seg_tile_rag_kernel: 16-byte accesses off the 16-byte grid, ~0.85-0.9 of the aligned rate).  For WIDE rows (>= 1 KiB) whose edges
  // dominate the node table, padding the table to whole 64-byte lines once per call (one streaming pass over N rows in, one over K rows
  // out) is still a little faster: F = 601 9.25 -> 8.65 ms, 602 9.46 -> 8.69 ms at 23 M edges on 233 k nodes (profiles/r04/bench_odd_feat.txt);
  // narrower rows keep the ragged kernel as it is (F = 130: 2.02 ms against 2.18 padded).
  const int64_t feat_given = x.size(1), esize = (int64_t)x.element_size(), lane_elems = 16 / esize;
  const int64_t unit = 64 / esize;
  const bool pad = x.dim() == 2 && feat_given % lane_elems != 0 && feat_given * esize >= 1024 && e.di.numel() >= 16 * std::max<int64_t>(x.size(0), 1);
  if (pad) x = at::constant_pad_nd(x, {0, (unit - feat_given % unit) % unit}, 0);
  const int64_t nnz = e.di.numel(), feat = x.size(1);
  auto launch = [&](int64_t rows) {
    at::Tensor o = at::empty({rows, feat}, x.options());
    auto run_edges = [&](at::Tensor &o) {
      const at::Tensor ws = workspace(x, geot_workspace_bytes(nnz, feat, rows, dt));
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
      plan = slab_plan_for(e.si, e.di, rows, x, has_w ? 1 : 0, 1, red);
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
            plan->w_planorder = to_plan_order(*plan, e.w, 1, x);
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
  at::Tensor out = rows_given >= 0 ? launch(rows_given) : with_row_rule(e.di_given, e.rows, e.fresh, launch);
  return pad ? out.narrow(1, 0, feat_given).contiguous() : out;
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
      const at::Tensor ws = workspace(x, geot_mh_workspace_bytes(nnz, heads, feat, rows, dt));
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
          // a STATIC weight (the same content on the second call: attention coefficients of a model that is being served, a fixed
          // multi-head adjacency) is brought into the plan's edge order once and then read without the permutation (weight mode 5:
          // 7.2 -> 6.1 ms at configs[3]) - the multi-head form of what gather_common does for a normalised adjacency
          const void *wptr = w_em.data_ptr();
          int wmode = 2;
          at::Tensor w_planorder;
          ContentKey wk;
          if (layout == GEOT_W_EDGE_MAJOR) {
            const bool w_owned = owned_product(e.w); // (takes g_mu: before wmu)
            if (may_remember({&e.w}) && content_key(e.w, &wk)) {
              std::lock_guard<std::mutex> lk(plan->wmu);
              if (plan->w_planorder.defined() && plan->w_key == wk && plan->w_given && !plan->w_given->expired()) {
                w_planorder = plan->w_planorder;
                plan->w_made.before_use(x, {&w_planorder});
                guard_check(plan->w_fp, {&e.w});
              } else if (tl_capturing) {
                // (no new cache content during a capture)
              } else if (plan->w_seen_valid && plan->w_seen == wk && plan->keep.size() > 2) {
                plan->w_planorder = to_plan_order(*plan, e.w, heads, x);
                plan->w_fp = w_owned ? at::Tensor() : guard_store({&e.w});
                plan->w_made.mark(x);
                plan->w_key = wk;
                plan->w_given = weak_of(e.w);
                w_planorder = plan->w_planorder;
              } else {
                plan->w_seen = wk;
                plan->w_seen_valid = true;
              }
              if (w_planorder.defined()) {
                wptr = w_planorder.data_ptr();
                wmode = 5;
              }
            }
            guard_flush();
          }
          return run_slab(*plan, wptr, wmode, x, o, heads, feat);
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
          // The persistent kernel leaves its results in the PLAN's edge order (8 per 32-byte piece) in a scratch tensor and a second
          // kernel brings them into edge order group by group through LDS: written straight to out[original edge id] every 4-byte
          // result is a partial write of its own (3.6 GB written for 0.46 GB of results at 115 M edges).
          const at::Tensor ws = workspace(m1, geot_slab_workspace_bytes(&plan->plan, m1.size(1)));
          const at::Tensor staging = at::empty_like(o);
          GEOT_CALL(geot_slab_sddmm_staged(&plan->plan, m1.data_ptr(), m2.data_ptr(), o.data_ptr(), staging.data_ptr(), m1.size(1), m1.size(0),
                                           m2.size(0), dtype_code(m1, "sddmm_coo"), ws.data_ptr(), ws.numel(), stream_of(m1)));
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

// d/dweight of mh_spmm: out(e, h) = <mat_1[dst_index[e], h, :], mat_2[src_index[e], h, :]>, laid out like the weight it is the
// gradient of - [nnz, H] or (head_major) [H, nnz].  The reference's mh_spmm has no backward (geot/mh_spmm.py:4-12); the pattern is
// sddmm_coo_impl of the single-head op (geot/gather_weight_scatter.py:8-12,46-50).
at::Tensor mh_sddmm_op(const at::Tensor &si_in, const at::Tensor &di_in, const at::Tensor &m1_in, const at::Tensor &m2_in, bool head_major) {
  TORCH_CHECK(si_in.dim() == 1 && di_in.dim() == 1 && si_in.size(0) == di_in.size(0), "src_index and dst_index must be 1 dimensional");
  TORCH_CHECK(m1_in.dim() == 3 && m2_in.dim() == 3 && m1_in.size(1) == m2_in.size(1) && m1_in.size(2) == m2_in.size(2),
              "mat_1 and mat_2 must be 3 dimensional with the same heads and feature dimensions");
  require_gpu("mh_sddmm", {&si_in, &di_in, &m1_in, &m2_in});
  TORCH_CHECK(m1_in.scalar_type() == m2_in.scalar_type(), "expected mat_2 of dtype ", toString(m1_in.scalar_type()), " but found ",
              toString(m2_in.scalar_type()));
  GEOT_DEVICE_GUARD(m1_in);
  at::Tensor si = as_int64(si_in), di = as_int64(di_in);
  at::Tensor m1 = m1_in.contiguous(), m2 = m2_in.contiguous();
  const int64_t nnz = di.size(0), heads = m1.size(1), feat = m1.size(2);
  if (nnz == 0 || heads == 0) return head_major ? at::empty({heads, nnz}, m1.options()) : at::empty({nnz, heads}, m1.options());
  const int dt = dtype_code(m1, "mh_sddmm");
  auto per_edge = [&](at::Tensor &o, bool hm) {
    GEOT_CALL(geot_mh_sddmm_coo(index_ptr(si), index_ptr(di), m1.data_ptr(), m2.data_ptr(), o.data_ptr(), nnz, heads, feat, m1.size(0), m2.size(0),
                                hm ? GEOT_W_HEAD_MAJOR : GEOT_W_EDGE_MAJOR, dt, stream_of(m1)));
  };
  // a dense graph that has (or now earns) the source-blocked plan of its forward mh_spmm: the SDDMM over that plan, the gathered
  // rows re-used out of L2 (results edge-major; a head-major caller gets them transposed)
  const int64_t rowbytes = heads * feat * m1.element_size(), ebytes = heads * m1.element_size();
  if (m1.scalar_type() != at::kDouble && (heads == 1 || heads == 2 || heads == 4 || heads == 8) && ebytes <= 16 &&
      (rowbytes == 256 || rowbytes == 512 || rowbytes == 1024) && (feat * m1.element_size()) % 16 == 0 && m1.size(0) < ((int64_t)1 << 31) &&
      (g_opt.slab_mode == 1 || (g_opt.slab_mode == 0 && slab_worthwhile(nnz, m1.size(0), m2.size(0), rowbytes, dt))) &&
      (tl_capturing || index_facts(di).ascending)) {
    if (auto plan = slab_plan_for(si, di, m1.size(0), m2, 2, heads)) {
      at::Tensor out = at::empty({nnz, heads}, m1.options());
      auto run_plan = [&](at::Tensor &o) -> bool {
        const std::vector<at::Tensor> pinned = plan->pinned();
        if (pinned.empty()) return false;
        const at::Tensor ws = workspace(m1, geot_slab_workspace_bytes(&plan->plan, heads * feat));
        const at::Tensor staging = at::empty_like(o);
        GEOT_CALL(geot_slab_mh_sddmm(&plan->plan, m1.data_ptr(), m2.data_ptr(), o.data_ptr(), staging.data_ptr(), heads, feat, m1.size(0), m2.size(0), dt,
                                     ws.data_ptr(), ws.numel(), stream_of(m1)));
        plan->launched_on(m1, pinned);
        return true;
      };
      auto run_edges = [&](at::Tensor &o) { per_edge(o, false); };
      out = plan_or_edges(plan, 1, out, m1, run_plan, run_edges);
      return head_major ? out.t().contiguous() : out;
    }
  }
  at::Tensor out = head_major ? at::empty({heads, nnz}, m1.options()) : at::empty({nnz, heads}, m1.options());
  per_edge(out, head_major);
  return out;
}


// Backward of the max / min aggregation of the gather ops: (grad_src, grad_weight) - the gradient of out[d, f] goes to the messages that
// attain it, divided evenly among ties (torch.scatter_reduce's rule).  No counterpart in the reference (its wrappers differentiate the
// sum only, geot/gather_scatter.py:21-39).
std::tuple<at::Tensor, at::Tensor> gather_select_backward_op(const at::Tensor &si_in, const at::Tensor &di_in, const c10::optional<at::Tensor> &weight,
                                                             const at::Tensor &src, const at::Tensor &out, const at::Tensor &grad) {
  check_gather(si_in, di_in, src, 2);
  const bool has_w = weight.has_value() && weight->defined();
  TORCH_CHECK(out.dim() == 2 && grad.dim() == 2 && out.sizes() == grad.sizes() && out.size(1) == src.size(1), "out and grad must be [rows, feat] of the forward result");
  require_gpu("gather_select_backward", {&si_in, &di_in, &src, &out, &grad, has_w ? &*weight : nullptr});
  TORCH_CHECK(src.scalar_type() == at::kFloat || src.scalar_type() == at::kDouble,
              "gather_scatter / gather_weight_scatter: the backward of reduce='max' / 'min' needs float32 or float64 (float atomics), got ", toString(src.scalar_type()));
  TORCH_CHECK(out.scalar_type() == src.scalar_type() && grad.scalar_type() == src.scalar_type() && (!has_w || weight->scalar_type() == src.scalar_type()),
              "gather_select_backward: one dtype for src, out, grad and weight");
  GEOT_DEVICE_GUARD(src);
  at::Tensor si = as_int64(si_in), di = as_int64(di_in);
  at::Tensor x = src.contiguous(), o = out.contiguous(), g = grad.contiguous();
  at::Tensor w = has_w ? weight->contiguous() : at::Tensor();
  const int64_t nnz = di.numel(), feat = x.size(1);
  at::Tensor gsrc = at::empty_like(x), ties = at::empty_like(o);
  at::Tensor gw = has_w ? at::empty({nnz}, x.options()) : at::empty({0}, x.options());
  GEOT_CALL(geot_gather_select_backward(index_ptr(si), index_ptr(di), has_w ? w.data_ptr() : nullptr, x.data_ptr(), o.data_ptr(), g.data_ptr(), ties.data_ptr(),
                                        gsrc.data_ptr(), has_w ? gw.data_ptr() : nullptr, nnz, feat, x.size(0), o.size(0), dtype_code(x, "gather_select_backward"),
                                        stream_of(x)));
  return std::make_tuple(gsrc, gw);
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
  const at::Tensor ws = workspace(src, geot_csr_workspace_bytes(nnz, feat, rows, dt));
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
  m.def("mh_sddmm(Tensor src_index, Tensor dst_index, Tensor mat_1, Tensor mat_2, bool head_major) -> Tensor");
  m.def("gather_select_backward(Tensor src_index, Tensor dst_index, Tensor? weight, Tensor src, Tensor out, Tensor grad) -> (Tensor, Tensor)");
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
  m.impl("mh_sddmm", GUARDED(mh_sddmm_op));                                                   \
  m.impl("gather_select_backward", GUARDED(gather_select_backward_op));                       \
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
  m.impl("mh_sddmm", mh_sddmm_op);
  m.impl("gather_select_backward", gather_select_backward_op);
  m.impl("transpose_edges", transpose_edges_op);
  m.impl("transposed_weight", transposed_weight_op);
  m.impl("transpose_edges_weighted", transpose_edges_weighted_op);
}
