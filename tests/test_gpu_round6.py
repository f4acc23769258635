"""Round 6 (`-m gpu`): the 16-bit multi-head SpMM on the matrix cores (seg_slab_spmm_mfma_kernel) and the IEEE isolation of every
source-blocked kernel.

* the matrix-core SpMM against float64 sums for every shape / weight layout it serves, exact equality with the row-per-wave kernel on
  integer data (any slip in the operand maps - which lane holds which edge of which row, the transposed LDS reads, the selector
  masks - shows as a wrong integer, not as rounding);
* a source table with Inf / NaN rows: only the destination rows that really have such a source are affected, for EVERY kernel that
  runs a plan (the matrix-core SpMM multiplies other rows' sources by a selector's zero - 0 x Inf would be a NaN next door - so its
  launch is gated on a finite table and a vector-ALU twin takes such calls: this test is what holds that construction to account).

Reference semantics: csrc/cuda/mh_spmm_kernel.cuh:28-111 (dst[d, h, :] += w[e, h] * src[s, h, :]), test/test_mh_spmm.py:4-10.
"""
import numpy as np
import pytest
import torch

from conftest import powerlaw_index

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def geot():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import geot_amd
    return geot_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _dense_graph(rng, nodes, nnz, hub=20):
    di = powerlaw_index(nnz, nodes, nodes + 1)
    di[: nnz // hub] = di[nnz // hub]                                   # a hub that is split (carry slots)
    di = np.sort(di)
    di[di == 7] = 8                                                     # a destination without edges
    di[-1] = nodes - 1
    si = rng.integers(0, nodes, nnz).astype(np.int64)
    return si, di


def _ref(d_si, d_di, w, v, nodes):
    """float64 sums; w [nnz, H] or None, v [nodes, H, F]."""
    msg = v.double()[d_si]
    if w is not None:
        msg = msg * w.double()[:, :, None]
    return torch.zeros(nodes, *v.shape[1:], device="cuda", dtype=torch.float64).index_add_(0, d_di, msg)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("H,Fh", [(4, 64), (1, 256), (2, 128), (8, 32), (1, 128), (2, 64), (4, 32), (8, 16), (8, 64), (4, 128), (2, 256)])   # rows of 512, 256 and 1024 bytes
def test_matrix_core_spmm_against_float64_and_the_row_per_wave_kernel(geot, dtype, H, Fh):
    """Every weight layout the kernel serves (edge-major through the permutation, head-major, plan order, none), a hub split into
    pieces (carry rows), rows without edges, out-of-range sources (contribute nothing): float64 sums within the storage type's
    rounding; integer data: EQUAL to the row-per-wave kernel's bits and to the exact sums."""
    from geot_amd import slab
    rng = np.random.default_rng(97 * H + Fh)
    nodes, nnz = 2500, 300_000
    si, di = _dense_graph(rng, nodes, nnz)
    si[rng.integers(0, nnz, 50)] = nodes + 7                            # out-of-range sources
    d_si, d_di = dev(si), dev(di)
    ok = dev(si < nodes)
    rowbytes = H * Fh * 2
    R = slab.rows_per_group(2, H, dtype, rowbytes)                      # (the rule keeps 16-bit plans within a matrix-core operand's 16 rows)
    plan = slab.build_plan(d_si, d_di, nodes, nodes, rowbytes, 2, H, rows_per_group=R)
    assert plan.meta["split_rows"] >= 1 and plan.meta["rows_per_group"] <= 16
    e_perm = plan.tensors["e_perm"].long()
    tol = 2.0 ** -9 if dtype == torch.float16 else 2.0 ** -6

    def run(weight, mode, v, mfma):
        geot.hip.set_option("slab_spmm_mfma", mfma)
        o = torch.full((nodes, H, Fh), float("nan"), device="cuda", dtype=dtype)
        slab.slab_spmm_out(plan, weight, mode, v, o, H, Fh, stage_weights=False)
        assert ("seg_slab_spmm_mfma_kernel" in geot.hip.last_kernel()) == bool(mfma), geot.hip.last_kernel()
        return o

    try:
        # (a) real-valued data against float64
        v = torch.from_numpy(rng.random((nodes, H, Fh), dtype=np.float32) - 0.3).to(dtype).cuda()
        w = torch.from_numpy(rng.random((nnz, H), dtype=np.float32) - 0.3).to(dtype).cuda()
        wz = w * ok[:, None]
        ref = _ref(d_si.clamp(max=nodes - 1), d_di, wz, v, nodes)
        mag = _ref(d_si.clamp(max=nodes - 1), d_di, wz.abs(), v.abs(), nodes)          # per element: the sum of |contributions|
        forms = [("edge-major", w, 2), ("head-major", w.t().contiguous(), 3), ("plan order", w[e_perm].contiguous(), 5)]
        outs = []
        for name, weight, mode in forms:
            o = run(weight, mode, v, 1)
            assert bool(((o.double() - ref).abs() <= tol * mag + 1e-30).all()), (name, float(((o.double() - ref).abs() / (mag + 1e-30)).max()))
            outs.append(o)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])          # one order of additions, whatever the layout
        assert torch.equal(run(w, 2, v, 1), outs[0])                                    # fixed by the plan, not by timing
        if H in (4, 8):                                                                 # 8 / 16 bytes of 16-bit heads an edge: brought into plan order a
            o_st = torch.full((nodes, H, Fh), float("nan"), device="cuda", dtype=dtype)  # group at a time through LDS first (slab_stage_kernel) - the same bits
            slab.slab_spmm_out(plan, w, 2, v, o_st, H, Fh)                              # (stage_weights=True: the workspace has room)
            assert torch.equal(o_st, outs[0])
        ones = ok[:, None].to(dtype).expand(nnz, H)
        ref0, mag0 = _ref(d_si.clamp(max=nodes - 1), d_di, ones, v, nodes), _ref(d_si.clamp(max=nodes - 1), d_di, ones, v.abs(), nodes)
        o0 = run(None, 0, v, 1)
        assert bool(((o0.double() - ref0).abs() <= tol * mag0 + 1e-30).all())
        # (b) integer data: exact in fp32, so both kernels round the same number
        vi = torch.from_numpy(rng.integers(-1, 2, (nodes, H, Fh)).astype(np.float32)).to(dtype).cuda()
        wi = torch.from_numpy(rng.integers(-2, 3, (nnz, H)).astype(np.float32)).to(dtype).cuda()
        exact = _ref(d_si.clamp(max=nodes - 1), d_di, wi * ok[:, None], vi, nodes)
        a = run(wi, 2, vi, 1)
        b = run(wi, 2, vi, 0)
        assert torch.equal(a, exact.to(dtype))                                          # an out-of-range source contributes NOTHING here
        # (the row-per-wave kernel reads row 0 for such an edge - indices out of range are outside the contract, csrc/gather_scatter.cpp:25-34
        # checks nothing; both are memory-safe): the two kernels agree bit for bit on every row that has no such edge
        clean_rows = torch.ones(nodes, dtype=torch.bool, device="cuda")
        clean_rows[d_di[~ok]] = False
        assert torch.equal(a[clean_rows], b[clean_rows]), float((a.double() - b.double())[clean_rows].abs().max())
    finally:
        geot.hip.set_option("slab_spmm_mfma", 1)


def _poison(v, rows, dtype):
    v = v.clone()
    v[rows[0]] = float("inf")
    v[rows[1]] = float("-inf")
    v[rows[2]] = float("nan")
    v[rows[3], ..., 5] = float("inf")                                   # a single element of a row
    return v.to(dtype)


@pytest.mark.parametrize("case", ["mh bf16 512", "mh f16 512", "mh f32 1024", "mh f32 512", "gws f32 512", "gws bf16 256", "gs f32 512", "mh bf16 1024", "mh f16 256"])
def test_nonfinite_sources_touch_only_their_own_destinations(geot, case):
    """Inf / -Inf / NaN rows (and a row with one Inf element) in the source table: a destination row is non-finite exactly where the
    float64 reference says so, and every other row equals the result computed from the table with those rows zeroed - bit for bit on
    the same kernel where that kernel does not depend on the data (every kernel but the gated matrix-core SpMM, whose finite twin is
    compared within rounding).  Covers seg_slab_kernel (lane groups, whole-wave rows), seg_slab_wrow_kernel, seg_slab_twin1k_kernel and the gated pairs."""
    from geot_amd import slab
    kind, tname, rowbytes = case.split()
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[tname]
    rowbytes = int(rowbytes)
    esz = 2 if dtype != torch.float32 else 4
    H = 4 if kind == "mh" else 1
    Fh = rowbytes // esz // H
    rng = np.random.default_rng(len(case) + rowbytes)
    nodes, nnz = 2500, 300_000
    si, di = _dense_graph(rng, nodes, nnz)
    bad = rng.choice(nodes, 4, replace=False)
    clean_row = int(np.setdiff1d(np.arange(nodes), bad)[0])
    si[np.isin(si, bad)] = clean_row                                    # keep the poisoned rows RARE: three edges each
    for b in bad:
        si[rng.integers(0, nnz, 3)] = b
    d_si, d_di = dev(si), dev(di)
    wmode = {"mh": 2, "gws": 1, "gs": 0}[kind]
    plan = slab.build_plan(d_si, d_di, nodes, nodes, rowbytes, wmode, H, rows_per_group=min(16, slab.rows_per_group(wmode, H, dtype, rowbytes)))
    v32 = torch.from_numpy(rng.random((nodes, H, Fh), dtype=np.float32) + 0.5).cuda()
    w = None if kind == "gs" else (torch.from_numpy(rng.random((nnz, H), dtype=np.float32) + 0.5).to(dtype).cuda())
    v_bad = _poison(v32, bad, dtype)
    v_zero = v32.clone()
    v_zero[bad] = 0
    v_zero = v_zero.to(dtype)
    shape = (nodes, H, Fh)

    def run(v):
        o = torch.full(shape, 7.0, device="cuda", dtype=dtype)
        weight = None if w is None else (w if kind == "mh" else w.view(-1))
        slab.slab_spmm_out(plan, weight, wmode, v.view(nodes, H * Fh) if kind != "mh" else v, o.view(nodes, H * Fh) if kind != "mh" else o,
                           H, Fh, stage_weights=False)
        return o, geot.hip.last_kernel()

    got, kernel_bad = run(v_bad)
    clean, kernel_clean = run(v_zero)
    ref = _ref(d_si, d_di, None if w is None else w.float(), v_bad.float(), nodes)
    touched = torch.zeros(nodes, dtype=torch.bool, device="cuda")
    touched[d_di[torch.isin(d_si, dev(bad))]] = True
    assert 0 < int(touched.sum()) < nodes // 4
    # where the reference is non-finite so is the result, with the reference's kind (NaN / +Inf / -Inf); nowhere else
    g64 = got.double()
    assert torch.equal(torch.isnan(g64), torch.isnan(ref)), case
    inf_ref = torch.isinf(ref)                                           # (finite sums stay below 4e4: nothing overflows the storage type)
    assert torch.equal(torch.isinf(g64), inf_ref) and torch.equal(torch.sign(g64[inf_ref]), torch.sign(ref[inf_ref]))
    assert not bool((~torch.isfinite(g64))[~touched].any())
    # rows no poisoned source reaches: the values of the clean table's run
    if "spmm_mfma" in kernel_clean:                                      # the finite call ran on the matrix cores, the poisoned one on the twin
        assert "spmm_mfma" in kernel_bad                                 # (the label names the pair's first kernel; the gate picked the twin)
        tol = (2.0 ** -6 if dtype == torch.bfloat16 else 2.0 ** -9) * float(clean.double().abs().max())
        assert float((g64[~touched] - clean.double()[~touched]).abs().max()) <= tol
        geot.hip.set_option("slab_spmm_mfma", 0)
        try:
            clean_twin, k2 = run(v_zero)
            assert ("twin1k" if rowbytes == 1024 else "wrow") in k2, k2
        finally:
            geot.hip.set_option("slab_spmm_mfma", 1)
        assert torch.equal(got[~touched], clean_twin[~touched])          # the twin IS the row-per-wave kernel (rows of 1 KiB: seg_slab_twin1k_kernel)
    else:
        assert torch.equal(got[~touched], clean[~touched]), case


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_nonfinite_rows_and_the_plan_sddmm(geot, dtype):
    """The SDDMMs over a plan (row-per-wave and matrix cores): a score is non-finite exactly for the edges whose own source or
    destination row is - nothing is contaminated by a neighbour in the tile (each product sums its OWN edge's features only)."""
    from geot_amd import slab
    rng = np.random.default_rng(77)
    nodes, nnz, H = 2500, 300_000, 4
    Fh = 64 if dtype == torch.bfloat16 else 32                           # 512-byte rows
    si, di = _dense_graph(rng, nodes, nnz)
    bad = rng.choice(np.arange(10, nodes - 10), 4, replace=False)
    d_si, d_di = dev(si), dev(di)
    plan = slab.build_plan(d_si, d_di, nodes, nodes, 512, 2, H, rows_per_group=min(16, slab.rows_per_group(2, H, dtype, 512)))
    q = (torch.rand(nodes, H, Fh, device="cuda") + 0.5).to(dtype)
    k = _poison(torch.rand(nodes, H, Fh, device="cuda") + 0.5, bad, dtype)
    s = torch.empty(nnz, H, device="cuda", dtype=dtype)
    slab.slab_mh_sddmm_out(plan, q, k, s)
    assert ("mfma" in geot.hip.last_kernel()) == (dtype == torch.bfloat16)
    ref = (q.double()[d_di] * k.double()[d_si]).sum(-1)
    hit = torch.isin(d_si, dev(bad))
    assert torch.equal(~torch.isfinite(s.double()), ~torch.isfinite(ref)) and not bool((~torch.isfinite(s.double()))[~hit].any())
    fin = torch.isfinite(ref)
    tol = 2.0 ** -6 if dtype == torch.bfloat16 else 2e-5
    assert float(((s.double() - ref)[fin]).abs().max()) <= tol * float(ref[fin].abs().max())


@pytest.mark.parametrize("H,Fh", [(4, 64), (8, 64)])                   # rows of 512 bytes, rows of 1 KiB (two passes)
def test_matrix_core_spmm_through_the_operator_and_the_handle(geot, H, Fh):
    """The drop-in operator (mh_spmm on a dense graph: plan on the second sighting) and geot_amd.Graph reach the matrix-core kernel for
    bf16 H=4 x F=64 and H=8 x F=64, forward and backward (d/dsrc runs the same kernel over the transposed list's plan, d/dweight the
    matrix-core SDDMM)."""
    from geot_amd import ops
    nodes, nnz = 4000, 600_000
    rng = np.random.default_rng(5)
    si, di = _dense_graph(rng, nodes, nnz)
    d_si, d_di = dev(si), dev(di)
    x = (torch.rand(nodes, H, Fh, device="cuda") - 0.3).bfloat16().requires_grad_()
    w = (torch.rand(nnz, H, device="cuda") - 0.3).bfloat16().requires_grad_()
    ref = _ref(d_si, d_di, w.detach(), x.detach(), nodes)
    g = geot.Graph(d_si, d_di, num_src=nodes, num_dst=nodes, slab_mode="always")
    y = g.mh_spmm(w, x)
    assert "seg_slab_spmm_mfma_kernel" in geot.hip.last_kernel(), geot.hip.last_kernel()
    assert float((y.double() - ref).abs().max()) <= 2.0 ** -6 * float(ref.abs().max())
    up = torch.rand_like(y)
    gx, gw = torch.autograd.grad(y, [x, w], up)
    xr, wr = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
    r = torch.zeros(nodes, H, Fh, device="cuda", dtype=torch.float64).index_add_(0, d_di, xr[d_si] * wr[:, :, None])
    rgx, rgw = torch.autograd.grad(r, [xr, wr], up.double())
    assert float((gx.double() - rgx).abs().max()) <= 2.0 ** -6 * float(rgx.abs().max())
    assert float((gw.double() - rgw).abs().max()) <= 2.0 ** -5 * float(rgw.abs().max())
    old = ops.set_option("slab_mode", "always")
    try:
        ops.clear_caches()
        for _ in range(3):
            y2 = geot.mh_spmm(d_si, d_di, w.detach(), x.detach())
        assert "seg_slab_spmm_mfma_kernel" in geot.hip.last_kernel(), geot.hip.last_kernel()
        assert float((y2.double() - ref).abs().max()) <= 2.0 ** -6 * float(ref.abs().max())
    finally:
        ops.set_option("slab_mode", old)
        ops.clear_caches()


@pytest.mark.parametrize("F", [128, 256])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_single_head_16_bit_sums_reach_the_matrix_cores(geot, dtype, F):
    """16-bit F = 128 / 256 (256- / 512-byte rows) on a dense graph, through the drop-in operators and through geot_amd.Graph: sums - weighted,
    unweighted, and the SDDMM of the weighted op's backward - run on plans cut into WAVES (the multi-head cut: csrc/host_plan.cpp
    slab_plan_for) by the matrix-core kernels, and so does the mean (the row's sum divided by its edge count where the row is written);
    max keeps its lane-group plan and vector-ALU kernel; fp32 keeps lane groups.  Values against float64."""
    from geot_amd import ops
    nodes, nnz = 4000, 600_000
    rng = np.random.default_rng(11 + F)
    si, di = _dense_graph(rng, nodes, nnz)
    d_si, d_di = dev(si), dev(di)
    x = (torch.rand(nodes, F, device="cuda") - 0.3).to(dtype)
    w = (torch.rand(nnz, device="cuda") - 0.3).to(dtype)
    tol = 2.0 ** -6 if dtype == torch.bfloat16 else 2.0 ** -9
    ref_w = torch.zeros(nodes, F, device="cuda", dtype=torch.float64).index_add_(0, d_di, x.double()[d_si] * w.double()[:, None])
    ref_1 = torch.zeros(nodes, F, device="cuda", dtype=torch.float64).index_add_(0, d_di, x.double()[d_si])
    cnt = torch.bincount(d_di, minlength=nodes).clamp(min=1).double()[:, None]

    def close(y, ref):
        return float((y.double() - ref).abs().max()) <= tol * float(ref.abs().max())

    old = ops.set_option("slab_mode", "always")
    try:
        ops.clear_caches()
        for _ in range(3):
            y = geot.gather_weight_scatter(d_si, d_di, w, x)
        assert "seg_slab_spmm_mfma_kernel" in geot.hip.last_kernel() and f", {2 * F}>" in geot.hip.last_kernel(), geot.hip.last_kernel()
        assert close(y, ref_w)
        for _ in range(2):
            y = geot.gather_scatter(d_si, d_di, x, "sum")
        assert "seg_slab_spmm_mfma_kernel" in geot.hip.last_kernel(), geot.hip.last_kernel()
        assert close(y, ref_1)
        for _ in range(3):
            y = geot.gather_scatter(d_si, d_di, x, "mean")
        assert "seg_slab_spmm_mfma_kernel" in geot.hip.last_kernel() and "mean" in geot.hip.last_kernel(), geot.hip.last_kernel()
        assert close(y, ref_1 / cnt)
        y_mean_w = geot.gather_weight_scatter(d_si, d_di, w, x, "mean")
        assert "mean" in geot.hip.last_kernel() and close(y_mean_w, ref_w / cnt)
        geot.hip.set_option("slab_spmm_mfma", 0)                             # the vector-ALU twin on the same plan: the same division
        try:
            y_twin = geot.gather_scatter(d_si, d_di, x, "mean")
            assert "seg_slab_wrow_kernel" in geot.hip.last_kernel(), geot.hip.last_kernel()
        finally:
            geot.hip.set_option("slab_spmm_mfma", 1)
        assert close(y_twin, ref_1 / cnt)
        for _ in range(3):
            y = geot.gather_scatter(d_si, d_di, x, "max")
        assert geot.hip.last_kernel().startswith("seg_slab_kernel<"), geot.hip.last_kernel()       # max / min: lane groups, vector ALUs
        big = torch.full((nodes, F), float("-inf"), device="cuda", dtype=torch.float64)
        ref_max = big.scatter_reduce(0, d_di[:, None].expand(-1, F), x.double()[d_si], "amax", include_self=True)
        has = torch.bincount(d_di, minlength=nodes) > 0
        assert torch.equal(y.double()[has], ref_max[has])
        # the weighted op's backward: d/dweight is the SDDMM over the same wave-cut plan
        xg, wg = x.clone().requires_grad_(), w.clone().requires_grad_()
        up = torch.rand(nodes, F, device="cuda").to(dtype)
        gx, gw = torch.autograd.grad(geot.gather_weight_scatter(d_si, d_di, wg, xg), [xg, wg], up)
        assert close(gw, (up.double()[d_di] * x.double()[d_si]).sum(-1))
        sc = ops.sddmm_coo_impl(d_si, d_di, up, x)
        assert "seg_slab_sddmm_mfma_kernel" in geot.hip.last_kernel(), geot.hip.last_kernel()
        assert close(sc, (up.double()[d_di] * x.double()[d_si]).sum(-1))
        x32 = x.float()
        for _ in range(3):
            y = geot.gather_scatter(d_si, d_di, x32[:, :F // 2].contiguous(), "sum")         # fp32 rows of the same width: lane groups
        assert geot.hip.last_kernel().startswith("seg_slab_kernel<float"), geot.hip.last_kernel()
    finally:
        ops.set_option("slab_mode", old)
        ops.clear_caches()
    g = geot.Graph(d_si, d_di, num_src=nodes, num_dst=nodes, slab_mode="always")
    y = g.gather_weight_scatter(w, x)
    assert "seg_slab_spmm_mfma_kernel" in geot.hip.last_kernel(), geot.hip.last_kernel()
    assert close(y, ref_w)


def _device_powerlaw(nnz, keys, seed):
    """bench.py's generator on the device: full-size index arrays without a host round trip."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    from bench import powerlaw_index as gen
    return gen(nnz, keys, seed, torch.device("cuda"))


@pytest.mark.parametrize("order,H,F", [("edge", 4, 64), ("plan", 4, 64), ("plan", 8, 64)])
def test_cfg4_full_size_bf16_properties(geot, order, H, F):
    """BASELINE.json configs[3] at FULL size in bf16 storage (232 965 nodes, 114 615 892 edges, H=4 x F=64 - and eight heads of 64: rows of
    1 KiB, the two-pass form over 16-row plans with the lockstep at full scale): the matrix-core SpMM and the matrix-core score SDDMM
    as geot_amd.Graph dispatches them, weights in edge order and in plan order.  Size-independent properties (rows without edges
    exactly zero, determinism, the checksum of checksums in float64 within bf16's rounding of the OUTPUT) and sampled rows - the hub,
    the ends, 150 random rows - against the float64 sum at 2^-7 relative."""
    nodes, nnz = 232_965, 114_615_892
    di = _device_powerlaw(nnz, nodes, 11)
    g = torch.Generator(device="cuda")
    g.manual_seed(12)
    si = torch.randint(0, nodes, (nnz,), device="cuda", generator=g)
    w = (torch.rand(nnz, H, device="cuda", generator=g) + 0.25).bfloat16()
    x = (torch.rand(nodes, H, F, device="cuda", generator=g) + 0.25).bfloat16()
    handle = geot.Graph(si, di, num_src=nodes, num_dst=nodes, slab_mode="always")
    wv = handle.plan_order(w, x) if order == "plan" else w
    out = handle.mh_spmm(wv, x)
    assert "seg_slab_spmm_mfma_kernel" in geot.hip.last_kernel(), geot.hip.last_kernel()
    assert out.shape == (nodes, H, F) and out.dtype == torch.bfloat16
    assert torch.equal(out, handle.mh_spmm(wv, x))                       # deterministic
    counts = torch.bincount(di, minlength=nodes)
    assert out[counts == 0].float().abs().sum().item() == 0              # rows without edges: exactly zero
    offs = torch.cumsum(counts, 0) - counts
    gen = torch.Generator().manual_seed(3)
    pick = [int(counts.argmax()), 0, nodes - 1, nodes // 2] + torch.randint(0, nodes, (150,), generator=gen).tolist()
    for k in pick:                                                       # comparator of test/test_mh_spmm.py:4-10, in float64
        e = slice(int(offs[k]), int(offs[k] + counts[k]))
        want = (x[si[e]].double() * w[e].double()[:, :, None]).sum(0)
        assert float((out[k].double() - want).abs().max()) <= 2.0 ** -7 * float(want.abs().max()) + 1e-30, k
    # checksum of checksums: column sums of the result == (per-node, per-head total weight) contracted with x; every output element
    # carries one bf16 rounding (relative 2^-9, random sign): the column sums over 233 k rows agree far inside 2^-9
    cw = torch.zeros(nodes, H, dtype=torch.float64, device="cuda").index_add_(0, si, w.double())
    want = torch.einsum("nh,nhf->hf", cw, x.double())
    got = out.double().sum(0)
    assert float(((got - want) / want).abs().max()) <= 2.0 ** -11, float(((got - want) / want).abs().max())
    # the scores of an attention layer over the same plan (matrix cores), sampled edges against float64
    q = (torch.rand(nodes, H, F, device="cuda", generator=g) / 4).bfloat16()
    s = handle.mh_sddmm(q, x, plan_order=(order == "plan"))
    assert "seg_slab_sddmm_mfma_kernel" in geot.hip.last_kernel(), geot.hip.last_kernel()
    s_edge = s.edge_order() if order == "plan" else s
    eid = torch.randint(0, nnz, (20_000,), device="cuda", generator=g)
    ref = (q[di[eid]].double() * x[si[eid]].double()).sum(-1)
    assert float((s_edge[eid].double() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    del handle
    torch.cuda.empty_cache()


def test_cfg4_full_size_eight_heads_through_the_operators_with_autograd(geot):
    """configs[3]'s graph at full size with EIGHT bf16 heads of 64 (rows of 1 KiB, 16 bytes of weights an edge) through the drop-in
    operator and its backward pass: the host layer's own permutations of per-edge values at this size (plan order, the transposed
    list: take_rows / geot_slab_to_plan_order - torch's gathers of such tensors are not to be trusted here, see graph._rows_at), the
    two-pass matrix-core SpMM over both lists' plans and the matrix-core SDDMM.  Sampled rows / edges against float64."""
    from geot_amd import ops
    nodes, nnz, H, F = 232_965, 114_615_892, 8, 64
    di = _device_powerlaw(nnz, nodes, 11)
    g = torch.Generator(device="cuda")
    g.manual_seed(12)
    si = torch.randint(0, nodes, (nnz,), device="cuda", generator=g)
    w = (torch.rand(nnz, H, device="cuda", generator=g) + 0.25).bfloat16().requires_grad_()
    x = (torch.rand(nodes, H, F, device="cuda", generator=g) + 0.25).bfloat16().requires_grad_()
    up = (torch.rand(nodes, H, F, device="cuda", generator=g) / 8).bfloat16()
    old = ops.set_option("slab_mode", "always")
    try:
        ops.clear_caches()
        for _ in range(2):
            y = geot.mh_spmm(si, di, w.detach(), x.detach())             # (the second sighting builds the plan)
        y = geot.mh_spmm(si, di, w, x)
        assert "seg_slab_spmm_mfma_kernel" in geot.hip.last_kernel() and "1024" in geot.hip.last_kernel(), geot.hip.last_kernel()
        gx, gw = torch.autograd.grad(y, [x, w], up)
    finally:
        ops.set_option("slab_mode", old)
        ops.clear_caches()
    assert bool(torch.isfinite(y.float()).all()) and bool(torch.isfinite(gx.float()).all()) and bool(torch.isfinite(gw.float()).all())
    counts = torch.bincount(di, minlength=nodes)
    offs = torch.cumsum(counts, 0) - counts
    wd, xd = w.detach(), x.detach()
    for k in [int(counts.argmax()), 0, nodes - 1] + torch.randint(0, nodes, (40,), generator=torch.Generator().manual_seed(5)).tolist():
        e = slice(int(offs[k]), int(offs[k] + counts[k]))
        want = (xd[si[e]].double() * wd[e].double()[:, :, None]).sum(0)
        assert float((y[k].double() - want).abs().max()) <= 2.0 ** -7 * float(want.abs().max()) + 1e-30, k
    eid = torch.randint(0, nnz, (20_000,), device="cuda", generator=g)                 # d/dweight[e, h] = <up[dst], x[src]>
    ref = (up[di[eid]].double() * xd[si[eid]].double()).sum(-1)
    assert float((gw[eid].double() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    for s_row in torch.randint(0, nodes, (4,), generator=torch.Generator().manual_seed(6)).tolist():   # d/dx[s] = sum over the edges out of s
        e = torch.nonzero(si == s_row).flatten()
        want = (up[di[e]].double() * wd[e].double()[:, :, None]).sum(0)
        assert float((gx[s_row].double() - want).abs().max()) <= 2.0 ** -6 * float(want.abs().max()) + 1e-30, s_row
    del y, gx, gw, w, x, up
    torch.cuda.empty_cache()


def test_cfg3_full_size_bf16_properties(geot):
    """BASELINE.json configs[2] at full size in bf16 storage (gws, 2.45 M nodes, 123.7 M edges, F=128; the per-edge tile kernel with fp32
    accumulation): rows without edges, sampled rows against float64 at 2^-7, the checksum of checksums."""
    nodes, nnz, F = 2_449_029, 123_718_280, 128
    di = _device_powerlaw(nnz, nodes, 7)
    g = torch.Generator(device="cuda")
    g.manual_seed(8)
    si = torch.randint(0, nodes, (nnz,), device="cuda", generator=g)
    w = (torch.rand(nnz, device="cuda", generator=g) + 0.25).bfloat16()
    x = (torch.rand(nodes, F, device="cuda", generator=g) + 0.25).bfloat16()
    out = geot.gather_weight_scatter(si, di, w, x)
    assert out.shape == (nodes, F) and out.dtype == torch.bfloat16 and "__bf16" in geot.hip.last_kernel()
    counts = torch.bincount(di, minlength=nodes)
    assert out[counts == 0].float().abs().sum().item() == 0
    offs = torch.cumsum(counts, 0) - counts
    gen = torch.Generator().manual_seed(4)
    pick = [int(counts.argmax()), 0, nodes - 1, nodes // 2] + torch.randint(0, nodes, (150,), generator=gen).tolist()
    for k in pick:
        e = slice(int(offs[k]), int(offs[k] + counts[k]))
        want = (x[si[e]].double() * w[e].double()[:, None]).sum(0)
        assert float((out[k].double() - want).abs().max()) <= 2.0 ** -7 * float(want.abs().max()) + 1e-30, k
    cw = torch.zeros(nodes, dtype=torch.float64, device="cuda").index_add_(0, si, w.double())
    want = cw @ x.double()
    got = out.double().sum(0)
    assert float(((got - want) / want).abs().max()) <= 2.0 ** -11


def test_gated_pair_inside_a_captured_graph(geot):
    """The matrix-core SpMM and its vector-ALU twin are BOTH enqueued and a word written on the stream picks one: inside a captured HIP
    graph the choice is therefore made at every REPLAY from the table's content of that moment - finite data runs on the matrix cores, a
    NaN written into the source table before the next replay sends that replay to the twin, and only the NaN's destinations change."""
    from geot_amd import slab
    rng = np.random.default_rng(12)
    nodes, nnz, H, Fh = 2500, 300_000, 4, 64
    si, di = _dense_graph(rng, nodes, nnz)
    d_si, d_di = dev(si), dev(di)
    plan = slab.build_plan(d_si, d_di, nodes, nodes, 512, 2, H, rows_per_group=min(16, slab.rows_per_group(2, H, torch.bfloat16, 512)))
    v = (torch.rand(nodes, H, Fh, device="cuda") + 0.5).bfloat16()
    w = (torch.rand(nnz, H, device="cuda") + 0.5).bfloat16()
    out = torch.empty(nodes, H, Fh, device="cuda", dtype=torch.bfloat16)
    slab.slab_spmm_out(plan, w, 2, v, out, H, Fh, stage_weights=False)          # (eager once: workspace, first-launch costs)
    eager = out.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        slab.slab_spmm_out(plan, w, 2, v, out, H, Fh, stage_weights=False)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        slab.slab_spmm_out(plan, w, 2, v, out, H, Fh, stage_weights=False)
    out.fill_(7)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    bad_row = int(si[1234])
    v[bad_row, 1, 3] = float("nan")                                              # the table changes between replays
    g.replay()
    torch.cuda.synchronize()
    touched = torch.zeros(nodes, dtype=torch.bool, device="cuda")
    touched[d_di[d_si == bad_row]] = True
    assert bool(torch.isnan(out[touched][:, 1, 3]).all()) and int(touched.sum()) > 0
    assert not bool(torch.isnan(out[~touched]).any())
    ref = _ref(d_si, d_di, w, v, nodes)
    fin = torch.isfinite(ref)
    err = torch.where(fin, (out.double() - ref).abs(), torch.zeros_like(ref))
    assert float(err.max()) <= 2.0 ** -6 * float(ref[fin].abs().max())
    v[bad_row, 1, 3] = 1.0
    g.replay()
    torch.cuda.synchronize()
    assert not bool(torch.isnan(out).any())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rows_without_edges_stay_zero_on_every_replay(geot, dtype):
    """A captured source-blocked call zero-fills its output (rows without edges rely on it) with a KERNEL: the hipMemsetAsync node it
    used to capture wrote a repeating 16-byte pattern - another kernel's argument block - into the output from the SECOND replay of a
    graph captured through PyTorch on, whenever an ordinary kernel ran between the replays (found in round 6 by the test above; every
    earlier capture test replayed once, or had an edge in every row).  Five replays with a fill of the output in between, 127 rows
    without edges: the eager result every time."""
    from geot_amd import slab
    rng = np.random.default_rng(21)
    nodes, nnz, H, Fh = 2500, 300_000, 4, (64 if dtype == torch.bfloat16 else 32)
    si, di = _dense_graph(rng, nodes, nnz)
    d_si, d_di = dev(si), dev(di)
    assert int((torch.bincount(d_di, minlength=nodes) == 0).sum()) > 100
    plan = slab.build_plan(d_si, d_di, nodes, nodes, 512, 2, H, rows_per_group=min(16, slab.rows_per_group(2, H, dtype, 512)))
    v = (torch.rand(nodes, H, Fh, device="cuda") + 0.5).to(dtype)
    w = (torch.rand(nnz, H, device="cuda") + 0.5).to(dtype)
    out = torch.empty(nodes, H, Fh, device="cuda", dtype=dtype)
    slab.slab_spmm_out(plan, w, 2, v, out, H, Fh, stage_weights=False)
    eager = out.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        slab.slab_spmm_out(plan, w, 2, v, out, H, Fh, stage_weights=False)
    for rep in range(5):
        out.fill_(7)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager), (rep, int((out != eager).sum()))
