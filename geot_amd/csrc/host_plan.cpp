// host_plan.cpp -- dense graphs: Phase A of the source-blocked kernel (csrc/seg_slab.hip), its routing rule and its cache (see host.h).
#include "host.h"

namespace geot_host {

// ---- dense graphs: source-blocked kernel (csrc/seg_slab.hip), Phase A -------------------------------------------------------

bool slab_worthwhile(int64_t nnz, int64_t out_rows, int64_t src_rows, int64_t rowbytes, int dtype) {
  if ((rowbytes != 256 && rowbytes != 512 && rowbytes != 1024) || nnz >= ((int64_t)1 << 31) || nnz < 8000000 || out_rows < 1 ||
      src_rows >= ((int64_t)1 << 31) || !geot_slab_full_chip()) // (a partitioned / CU-masked device: the rule below was not measured there)
    return false;
  // (what the plan of this shape will be built with: a unit is a wave for every row of >= 256 bytes - geot_slab_units_for - and the
  //  rows per group follow the kernel that runs it; 16-bit storage keeps fp32 accumulators: half the rows per group)
  const int64_t units = (int64_t)geot_slab_units_for(1, rowbytes);
  const int64_t rpg = geot_slab_rows_per_group_shape(1, 1, dtype, rowbytes);
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
  const int64_t units = units_override > 0 ? units_override : (int64_t)geot_slab_units_for(weight_mode, rowbytes);
  const int64_t R = rows_per_group > 0 ? rows_per_group : geot_slab_rows_per_group_shape(weight_mode, heads, GEOT_F32, rowbytes);
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
  // (a split row's edges go to its pieces interleaved - edge j to piece j % nv - so that every piece samples the row's whole source
  //  range also when the sources are sorted inside the row; see plan_vrows_kernel in seg_plan.hip)
  at::Tensor nv_of_v = nv_row.index_select(0, v_row), cnt_of_v = counts.index_select(0, v_row);
  at::Tensor v_cnt = at::div(cnt_of_v, nv_of_v, "floor") + v_piece.lt(at::remainder(cnt_of_v, nv_of_v)).to(at::kLong);
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
  at::Tensor vrow_e = vstart.index_select(0, dst_index) + at::remainder(e_id - rowptr.index_select(0, dst_index), nv_row.index_select(0, dst_index).clamp_min(1));
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
  const int64_t units = units_override > 0 ? units_override : (int64_t)geot_slab_units_for(weight_mode, rowbytes);
  const int64_t R = rows_per_group > 0 ? rows_per_group : geot_slab_rows_per_group_shape(weight_mode, heads, GEOT_F32, rowbytes);
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

constexpr int64_t kSlabBytes = 2 << 20; // measured (profiles/r02/bench_slab.txt; jointly with window and workgroups per CU: profiles/r04/sweep_slab_*.txt)
// ... and 1 MiB (with a window of 3 slabs, seg_slab.hip) under multi-head weights on 1-KiB rows, since the row loop got its scalar
// bases (round 4, profiles/r04/sweep_slab_*_v2.txt: mh fp32 7.21 vs 7.37 ms).  Everything else stays at 2 MiB: plans of one weight or
// none are within 1.5 % of their best there on rows of 256 / 512 bytes, and at 1 KiB the forward would gain 2 % (gws F=256 6.63 vs
// 6.76) where the SDDMM of its backward, which runs over the same plan, loses 4 % (8.21 vs 7.90); multi-head plans on rows of 512 / 256
// bytes (seg_slab_wrow_kernel): bf16 H=4 x F=64 5.24 ms at 2 MiB / window 2 against 5.41 at 1 MiB / window 3
int64_t slab_bytes_rule(int64_t rowbytes, int wmode) { return g_opt.slab_bytes > 0 ? g_opt.slab_bytes : ((wmode >= 2 && rowbytes >= 1024) ? kSlabBytes / 2 : kSlabBytes); }

// Does the graph have LOCALITY?  The source-blocked kernel pays off when the groups in flight sweep the WHOLE source table together;
// on a graph whose sources sit near their destinations every group lives in its own few slabs, the per-edge kernels serve it out of
// L2 as it is (their tile order is XCD-contiguous), and the plan loses 3-25x (profiles/r03/slab_vs_per_edge_by_locality.txt).
// Probed on 64 blocks of consecutive edges (dst order; a block ~ the edges of one group): distinct source slabs per block
// against what uniform sources would touch.  Uniform / power-law sources: ~1.0; sources within +-100 000 of 233 k rows: 0.86
// (plan and per-edge kernels level - the trial decides); +-20 000: 0.18, +-2 000: 0.03 (declined here: no Phase A, no trial).
// One small read-back, once per edge list.
double slab_source_coverage(const at::Tensor &si, const at::Tensor &di, int64_t src_rows, int64_t rowbytes, int64_t block_edges) {
  const int64_t nnz = si.numel();
  int64_t slab_rows = 1;
  while (2 * slab_rows * rowbytes <= kSlabBytes) slab_rows *= 2;
  const int64_t n_slabs = (src_rows + slab_rows - 1) / slab_rows;
  const int64_t L = std::min<int64_t>(std::max<int64_t>(block_edges, 256), 4096), B = 64;
  if (n_slabs < 8 || nnz < 4 * B * L) return 1.0;
  const auto lopt = si.options();
  at::Tensor starts = at::arange(B, lopt) * ((nnz - L) / B);
  // a block that falls inside a row of more than L edges (a hub) is spread over that whole row, as Phase A's interleaved pieces are:
  // on a list whose sources ascend inside every row, L CONSECUTIVE edges of a hub would cover one narrow source range and read
  // as locality the graph does not have
  at::Tensor key = di.index_select(0, starts);
  at::Tensor lo = at::searchsorted(di, key, /*out_int32=*/false, /*right=*/false), hi = at::searchsorted(di, key, false, true);
  at::Tensor len = hi - lo;
  at::Tensor hub = len.gt(L);
  at::Tensor base = at::where(hub, lo, starts).unsqueeze(1);
  at::Tensor step = at::where(hub, len.to(at::kDouble) / (double)L, at::ones_like(len).to(at::kDouble)).unsqueeze(1);
  at::Tensor idx = (base + (at::arange(L, lopt).to(at::kDouble).unsqueeze(0) * step).floor().to(at::kLong)).clamp_max(nnz - 1).flatten();
  at::Tensor slabs = at::div(si.index_select(0, idx), slab_rows, "floor").view({B, L});
  at::Tensor sorted = std::get<0>(slabs.sort(1));
  at::Tensor distinct = sorted.slice(1, 1, L).ne(sorted.slice(1, 0, L - 1)).sum(1) + 1;
  const double mean = distinct.to(at::kDouble).mean().item<double>();
  const double expect = (double)n_slabs * (1.0 - std::pow(1.0 - 1.0 / (double)n_slabs, (double)L));
  return mean / std::max(expect, 1.0);
}

std::list<SlabEntry> &g_slab = *new std::list<SlabEntry>; // (never destroyed: see g_facts)
std::list<std::pair<ContentKey, ContentKey>> g_sightings; // edge lists seen once (no tensors held)
std::list<std::pair<ContentKey, ContentKey>> g_declined;  // edge lists with locality (slab_source_coverage): per-edge kernels, no plan
std::list<std::pair<ContentKey, ContentKey>> g_building;  // edge lists whose plan a thread is building right now (the others keep the per-edge kernels)

std::shared_ptr<SlabPlanHolder> slab_plan_for(const at::Tensor &si, const at::Tensor &di, int64_t rows, const at::Tensor &src,
                                              int wmode, int64_t heads, int red) {
  const GuardFlush flush_questions_;
  const bool f32 = src.scalar_type() == at::kFloat;
  if (g_opt.slab_mode < 0 || rows < 1 || !(f32 || src.scalar_type() == at::kHalf || src.scalar_type() == at::kBFloat16)) return nullptr;
  const int dt = dtype_code(src, "slab");
  const int64_t rowbytes = (src.numel() / std::max<int64_t>(src.size(0), 1)) * src.element_size(), nnz = di.numel();
  // rows of 256 / 512 / 1024 bytes; 128-byte rows run too but were measured slower than the per-edge kernels (DESIGN.md
  // section 3.1d): only when the path is forced
  if ((rowbytes != 256 && rowbytes != 512 && rowbytes != 1024 && !(rowbytes == 128 && g_opt.slab_mode == 1)) || nnz == 0 ||
      nnz >= ((int64_t)1 << 31) || src.size(0) * rowbytes >= ((int64_t)1 << 32))   // (32-bit row offsets in the kernel: a table below 4 GiB)
    return nullptr;
  if (g_opt.slab_mode != 1 && !slab_worthwhile(nnz, rows, src.size(0), rowbytes, dt)) return nullptr;
  ContentKey k1, k2;
  if (!may_remember({&si, &di}) || !content_key(si, &k1) || !content_key(di, &k2)) return nullptr;
  // The plan's cut.  16-bit SUMS and MEANS over 256- / 512-byte rows with one weight per edge or none take the multi-head rule's cut -
  // waves, at most 16 rows per group - because that is the plan the matrix-core kernels run (seg_slab_spmm_mfma_kernel /
  // seg_slab_sddmm_mfma_kernel; bf16 at configs[3]'s graph, weighted / no weight / mean / the SDDMM of the backward pass: F = 128
  // 1.78 / 1.72 / 1.74 / 2.37 ms against the lane-group kernels' 2.30 / 2.17 / 2.18 / 3.27, F = 256 3.26 / 3.17 / 3.19 / 3.79 against
  // 4.03 / 3.88 / 3.91 / 5.48 - profiles/r06/slab_cases__rows_of_256_bytes.txt, slab_cases__rows_of_512_bytes_one_head.txt); max / min
  // keep lane groups (no matrix-core form, and the row-per-wave kernel that would run a wave-cut plan is slower than the lane-group one)
  const int cut_wmode = (!f32 && (rowbytes == 256 || rowbytes == 512) && (red == GEOT_REDUCE_SUM || red == GEOT_REDUCE_MEAN) && (wmode == 0 || wmode == 1)) ? 2 : wmode;
  const int rpg = geot_slab_rows_per_group_shape(cut_wmode, heads, dt, rowbytes);
  const int64_t units = geot_slab_units_for(cut_wmode, rowbytes);
  {
    std::lock_guard<std::mutex> lk(g_mu);
    sweep_expired_locked();
    for (auto it = g_slab.begin(); it != g_slab.end(); ++it)
      if (it->k1 == k1 && it->k2 == k2 && it->rows == rows && it->src_rows == src.size(0) && it->rowbytes == rowbytes &&
          it->rpg == rpg && it->units == units && !it->w1.expired() && !it->w2.expired()) { // (a plan serves every weight mode with its R and its units)
        g_slab.splice(g_slab.begin(), g_slab, it);
        g_slab.front().plan->made.before_use(src, g_slab.front().plan->keep);
        guard_check(g_slab.front().plan->fp, {&si, &di}); // (a rejected plan has released its fingerprint: the per-edge kernels read the caller's bytes)
        return g_slab.front().plan;
      }
    if (tl_capturing) return nullptr; // Phase A synchronises: never inside a capture (the per-edge kernels serve the call)
    if (g_opt.slab_mode != 1) { // first sighting of this edge list: only remember it - a one-shot call never pays for Phase A
      for (auto &dc : g_declined)
        if (dc.first == k1 && dc.second == k2) return nullptr;
    }
    for (auto &bd : g_building) // one builder per edge list: Phase A takes milliseconds and ~80 bytes per edge of transient memory
      if (bd.first == k1 && bd.second == k2) return nullptr;
    if (g_opt.slab_mode != 1) {
      bool seen = false;
      for (auto &sg : g_sightings) seen |= (sg.first == k1 && sg.second == k2);
      if (!seen) {
        g_sightings.emplace_back(k1, k2);
        if (g_sightings.size() > 64) g_sightings.pop_front();
        return nullptr;
      }
    }
    g_building.emplace_back(k1, k2); // this thread probes / builds; concurrent callers keep the per-edge kernels until the plan is there
  }
  struct Building { // (leaves the list however this function is left)
    const ContentKey &a, &b;
    ~Building() {
      std::lock_guard<std::mutex> lk(g_mu);
      g_building.remove_if([&](const std::pair<ContentKey, ContentKey> &x) { return x.first == a && x.second == b; });
    }
  } building{k1, k2};
  if (g_opt.slab_mode != 1 && g_opt.slab_min_coverage_pct > 0 && si.is_cuda()) { // a graph with locality keeps the per-edge kernels (and pays neither Phase A nor a trial)
    const int64_t rounds = std::max<int64_t>(1, (rows + rpg * units - 1) / (rpg * units));
    const double cover = slab_source_coverage(si, di, src.size(0), rowbytes, nnz / (rounds * units));
    std::lock_guard<std::mutex> lk(g_mu);
    g_stats.last_coverage_permille = (int64_t)(cover * 1000.0);
    if (cover * 100.0 < (double)g_opt.slab_min_coverage_pct) {
      ++g_stats.plans_declined;
      g_declined.emplace_back(k1, k2);
      if (g_declined.size() > 64) g_declined.pop_front();
      return nullptr;
    }
  }
  // (the build reads two small records back anyway: wait here, so that plan_us is the build and not the queue in front of it)
  if (di.is_cuda()) (void)hipStreamSynchronize(static_cast<hipStream_t>(stream_of(di)));
  const auto t0 = std::chrono::steady_clock::now();
  std::shared_ptr<SlabPlanHolder> plan;
  try {
    plan = slab_build(si, di, rows, src.size(0), rowbytes, wmode, heads, slab_bytes_rule(rowbytes, wmode), rpg, units);
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
  g_slab.push_front(SlabEntry{k1, k2, rows, src.size(0), rowbytes, heads, wmode, rpg, units, weak_of(si), weak_of(di), plan});
  while ((int)g_slab.size() > g_opt.slab_keep) g_slab.pop_back();
  enforce_cache_budget_locked();
  return plan;
}

// (the persistent grids of a process take turns on a device INSIDE the library - geot_slab_spmm / geot_slab_sddmm, seg_slab.hip
// "SlabTurn" - so every caller of the C ABI gets it, not only this plugin)
// false: the plan's arrays have been released (a trial on another thread rejected it) - the caller runs the per-edge kernels
bool run_slab(SlabPlanHolder &H, const void *weight, int wmode, const at::Tensor &src, at::Tensor &out, int64_t heads, int64_t feat, int red) {
  const std::vector<at::Tensor> pinned = H.pinned(); // this launch's own references: a release() meanwhile cannot free under the kernel
  if (pinned.empty()) return false;
  // (room for the call's weights in plan order: edge-order weights are staged inside the kernel instead of read through the permutation)
  const at::Tensor ws = workspace(src, geot_slab_workspace_bytes_staged(&H.plan, heads * feat, wmode, heads, dtype_code(src, "slab")));
  GEOT_CALL(geot_slab_spmm(&H.plan, weight, wmode, src.data_ptr(), out.data_ptr(), heads, feat, src.size(0), out.size(0), dtype_code(src, "slab"),
                           red, ws.data_ptr(), ws.numel(), stream_of(src)));
  H.launched_on(src, pinned);
  return true;
}


// Phase A of the source-blocked kernel as an op (works on CPU tensors too: the tests emulate the kernel on its output).
// Returns [e_src, e_dl, e_perm, g_begin, g_vrow0, g_nv, v_out, c_row, c_first, c_count, scalars(int64[12])]
std::vector<at::Tensor> slab_plan_op(const at::Tensor &si, const at::Tensor &di, int64_t rows, int64_t src_rows, int64_t rowbytes,
                                     int64_t weight_mode, int64_t heads, int64_t slab_bytes, int64_t rows_per_group, int64_t units) {
  TORCH_CHECK(si.dim() == 1 && di.dim() == 1 && si.numel() == di.numel(), "slab_plan: 1-D edge lists of equal length");
  auto H = slab_build(si.contiguous(), di.contiguous(), rows, src_rows, rowbytes, (int)weight_mode, heads,
                      slab_bytes > 0 ? slab_bytes : slab_bytes_rule(rowbytes, (int)weight_mode), rows_per_group, units);
  std::vector<at::Tensor> out = H->keep;
  at::Tensor sc = at::empty({13}, at::TensorOptions().dtype(at::kLong));
  int64_t *s = sc.data_ptr<int64_t>();
  s[0] = H->plan.n_groups; s[1] = H->plan.n_vrows; s[2] = H->plan.n_carry; s[3] = H->plan.n_split; s[4] = H->plan.nnz;
  s[5] = H->plan.units; s[6] = H->plan.rows_per_group; s[7] = H->rounds; s[8] = H->budget; s[9] = H->cap; s[10] = H->slabs; s[11] = H->slab_rows; s[12] = H->plan.slab_shift;
  out.push_back(sc);
  return out;
}

bool slab_worthwhile_op(int64_t nnz, int64_t rows, int64_t src_rows, int64_t rowbytes) { return slab_worthwhile(nnz, rows, src_rows, rowbytes, GEOT_F32); }


} // namespace geot_host
