"""Round-5 cases (`-m gpu`).

* the descent guard catches a descent THROUGH an ignored key ([a, out-of-range, a] under sorted = 1): the call
  repairs itself instead of writing row `a` from two separate runs;
* autograd of the operator surface beyond `sum`: `gather_scatter` / `gather_weight_scatter` with reduce='mean' against dense
  float64 autograd; max / min / prod refuse loudly; `mh_spmm` backward (d/dsrc over the transposed list, d/dweight by the
  per-head SDDMM) against the dense formula of the reference's test (test/test_mh_spmm.py:4-10), both weight layouts;
* ...

Reference semantics: csrc/util/check.cuh:78-111, geot/gather_weight_scatter.py:31-51 (the backward pattern).
"""
import warnings

import numpy as np
import pytest
import torch

from conftest import ROOT, powerlaw_index  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def geot():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import geot_amd
    return geot_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("F", [64, 3, 2])
@pytest.mark.parametrize("middle", ["above", "negative"])
def test_descent_through_an_ignored_key_is_caught(geot, F, middle):
    """[.., a, out-of-range, a, ..] with sorted = 1 through the C ABI: the out-of-range key is ignored, but the run of `a` is cut
    in two - the atomic-free kernels would write row a twice (the second run overwriting the first).  The guard sees a descent with
    one in-range side and the call repairs itself (sum over fp32: the reference's own atomic formulation)."""
    from geot_amd import hip, ops
    rng = np.random.default_rng(3)
    K, nnz = 500, 40_000
    index = np.sort(rng.integers(0, K, nnz)).astype(np.int64)
    index[-1] = K - 1
    # plant the pattern in the middle of a run, away from tile boundaries and at one
    for pos in (1234, 20_000, 512 * 40):
        a = index[pos - 1]
        index[pos] = K + 3 if middle == "above" else -5
        index[pos + 1] = a
    t_index = dev(index)
    src = torch.rand(nnz, F, device="cuda")
    geot.index_scatter(0, torch.rand(8, 4, device="cuda"), torch.arange(8, device="cuda"))   # (sets this thread's alarm word)
    alarms = ops.stats()["alarms"]
    out = torch.full((K, F), 7.0, device="cuda")
    hip.index_scatter_out(t_index, src, out, sorted=True)
    torch.cuda.synchronize()
    keep = torch.from_numpy((index >= 0) & (index < K)).cuda()
    want = torch.zeros(K, F, device="cuda", dtype=torch.float64).index_add_(0, t_index[keep], src[keep].double())
    assert torch.allclose(out.double(), want, rtol=1e-5, atol=1e-5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        geot.index_scatter(0, torch.rand(8, 4, device="cuda"), torch.arange(8, device="cuda"))   # the host layer hears the alarm here
    assert ops.stats()["alarms"] == alarms + 1


def _graph(rng, nodes, nnz, empty_rows=True):
    di = np.sort(rng.integers(0, nodes, nnz)).astype(np.int64)
    if empty_rows:
        di[di == 3] = 4                                   # a destination without edges
    di[-1] = nodes - 1
    si = rng.integers(0, nodes - 2, nnz).astype(np.int64)  # the last two source nodes have no out-edge
    return dev(si), dev(di)


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_mean_aggregation_is_differentiable(geot, weighted, dtype):
    """geot.gather_scatter(..., 'mean') / gather_weight_scatter(..., 'mean') route to geot::gather_reduce, which had a fake but no
    autograd registration (missing #2): a training loop got autograd's fallback warning and no gradient.  Now
    the sum backward on grad / deg; checked against plain-torch autograd of the same formula in float64."""
    rng = np.random.default_rng(17)
    nodes, nnz, F = 300, 9000, 24
    si, di = _graph(rng, nodes, nnz)
    x = torch.rand(nodes, F, device="cuda", dtype=dtype, requires_grad=True)
    w = torch.rand(nnz, device="cuda", dtype=dtype, requires_grad=True) if weighted else None
    out = geot.gather_weight_scatter(si, di, w, x, "mean") if weighted else geot.gather_scatter(si, di, x, "mean")
    up = torch.rand_like(out)
    grads = torch.autograd.grad(out, [x] + ([w] if weighted else []), up)
    xr = x.detach().double().requires_grad_()
    wr = w.detach().double().requires_grad_() if weighted else None
    msg = xr[si] * (wr[:, None] if weighted else 1.0)
    deg = torch.zeros(nodes, device="cuda", dtype=torch.float64).index_add_(0, di, torch.ones(nnz, device="cuda", dtype=torch.float64))
    ref = torch.zeros(nodes, F, device="cuda", dtype=torch.float64).index_add_(0, di, msg) / deg.clamp(min=1)[:, None]
    tol = 1e-12 if dtype == torch.float64 else 1e-5
    assert torch.allclose(out.double(), ref, rtol=tol, atol=tol)
    refg = torch.autograd.grad(ref, [xr] + ([wr] if weighted else []), up.double())
    for g, r in zip(grads, refg):
        assert g.shape == r.shape and g.dtype == dtype
        assert torch.allclose(g.double(), r, rtol=tol * 10, atol=tol * 10 * float(r.abs().max()))
    assert float(grads[0][-2:].abs().max()) == 0.0          # source nodes without out-edges: gradient rows exist and are zero


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("reduce", ["max", "min"])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_max_min_aggregation_gradients_follow_torch_scatter_reduce(geot, reduce, weighted, dtype):
    """The gradient of a max / min aggregation goes to the messages that attain it, evenly among ties - torch.scatter_reduce's rule.
    Data with MANY exact ties (features quantised to a few values, some weights equal) against plain-torch autograd of
    scatter_reduce(amax / amin, include_self=False) over the materialised messages."""
    rng = np.random.default_rng(21)
    nodes, nnz, F = 300, 9000, 24
    si, di = _graph(rng, nodes, nnz)
    x = (torch.randint(0, 4, (nodes, F), device="cuda").to(dtype) / 2 - 0.5).requires_grad_()
    w = (torch.randint(1, 3, (nnz,), device="cuda").to(dtype) / 2).requires_grad_() if weighted else None
    out = geot.gather_weight_scatter(si, di, w, x, reduce) if weighted else geot.gather_scatter(si, di, x, reduce)
    up = torch.rand_like(out)
    grads = torch.autograd.grad(out, [x] + ([w] if weighted else []), up)
    xr = x.detach().double().requires_grad_()
    wr = w.detach().double().requires_grad_() if weighted else None
    msg = xr[si] * (wr[:, None] if weighted else 1.0)
    # (self = a value no message takes: scatter_reduce's backward counts `self == result` among the ties even with include_self=False,
    # so a zero-filled self would take a share wherever the extremum is 0.0)
    ref = torch.full((nodes, F), 7.0, device="cuda", dtype=torch.float64).scatter_reduce(0, di[:, None].expand(-1, F), msg, "amax" if reduce == "max" else "amin",
                                                                                          include_self=False)
    has_edges = torch.zeros(nodes, device="cuda", dtype=torch.bool).index_fill_(0, di, True)
    ref = torch.where(has_edges[:, None], ref, torch.zeros_like(ref))          # rows without edges: 0, as the forward writes them
    assert torch.equal(out.double(), ref)                      # selections: exact
    refg = torch.autograd.grad(ref, [xr] + ([wr] if weighted else []), up.double())
    tol = 1e-12 if dtype == torch.float64 else 1e-5
    for g, r in zip(grads, refg):
        assert g.shape == r.shape and g.dtype == dtype
        assert torch.allclose(g.double(), r, rtol=tol * 10, atol=tol * 10 * float(r.abs().max()))


def test_prod_aggregation_refuses_a_backward_pass(geot):
    """... and what has no gradient here says so (like index_scatter's non-sum reductions, ops._is_backward) instead of autograd's
    silent fallback; 16-bit storage has no float atomic for the max / min backward and says so too."""
    rng = np.random.default_rng(18)
    si, di = _graph(rng, 50, 700)
    x = torch.rand(50, 8, device="cuda", requires_grad=True)
    out = geot.gather_scatter(si, di, x, "prod")
    with pytest.raises(NotImplementedError, match="backward is implemented for reduce='sum', 'mean', 'max' and 'min' only"):
        out.sum().backward()
    xh = torch.rand(50, 8, device="cuda").bfloat16().requires_grad_()
    with pytest.raises(RuntimeError, match="needs float32 or float64"):
        geot.gather_scatter(si, di, xh, "max").float().sum().backward()


def _mh_dense(si, di, w_em, x, rows):
    """test/test_mh_spmm.py:4-10 of the reference: index_select the sources, scale per head, index_add into the destinations."""
    msg = x[si] * w_em[:, :, None]
    return torch.zeros(rows, *x.shape[1:], device=x.device, dtype=x.dtype).index_add_(0, di, msg)


@pytest.mark.parametrize("head_major", [False, True])
@pytest.mark.parametrize("dtype,H,F", [(torch.float64, 4, 8), (torch.float32, 4, 64), (torch.float32, 3, 20), (torch.float32, 8, 16)])
def test_mh_spmm_backward_against_the_dense_formula(geot, head_major, dtype, H, F):
    """d/dsrc = mh_spmm over the transposed list with the weights in its order, d/dweight = the per-head SDDMM, in the layout the
    weight came in ([nnz, H] or [H, nnz]); float64 against plain-torch autograd of the reference test's formula."""
    rng = np.random.default_rng(19 + H)
    nodes, nnz = 400, 12_000
    si, di = _graph(rng, nodes, nnz)
    x = torch.rand(nodes, H, F, device="cuda", dtype=dtype, requires_grad=True)
    w_em = torch.rand(nnz, H, device="cuda", dtype=dtype)
    w = (w_em.t().contiguous() if head_major else w_em.clone()).requires_grad_()
    out = geot.mh_spmm(si, di, w, x)
    up = torch.rand_like(out)
    gx, gw = torch.autograd.grad(out, [x, w], up)
    xr, wr = x.detach().double().requires_grad_(), w_em.double().requires_grad_()
    ref = _mh_dense(si, di, wr, xr, nodes)
    rgx, rgw = torch.autograd.grad(ref, [xr, wr], up.double())
    if head_major:
        rgw = rgw.t()
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    assert torch.allclose(out.double(), ref, rtol=tol, atol=tol * float(ref.abs().max()))
    assert gx.shape == x.shape and gw.shape == w.shape
    assert torch.allclose(gx.double(), rgx, rtol=tol, atol=tol * float(rgx.abs().max()))
    assert torch.allclose(gw.double(), rgw, rtol=tol, atol=tol * float(rgw.abs().max()))


def test_mh_sddmm_c_abi_and_16bit(geot):
    """The C ABI entry point with raw pointers (geot_mh_sddmm_coo), both layouts, fp32 / bf16 / f16 storage (fp32 dot products),
    out-of-range endpoints give 0."""
    from geot_amd import hip
    rng = np.random.default_rng(23)
    nodes, nnz, H, F = 500, 30_000, 4, 32
    si, di = _graph(rng, nodes, nnz, empty_rows=False)
    si[5] = nodes + 7                                      # out-of-range source: the dot is 0
    for dt, tol in ((torch.float32, 1e-5), (torch.bfloat16, 2.0 ** -7), (torch.float16, 2.0 ** -10)):
        a = torch.rand(nodes, H, F, device="cuda").to(dt)
        b = torch.rand(nodes, H, F, device="cuda").to(dt)
        ok = (si < nodes)
        ref = (a.double()[di] * b.double()[si.clamp(max=nodes - 1)]).sum(-1) * ok[:, None]
        for hm in (False, True):
            out = torch.full((H, nnz) if hm else (nnz, H), 7.0, device="cuda", dtype=dt)
            hip.mh_sddmm_coo_out(si, di, a, b, out, hm)
            got = out.t() if hm else out
            assert torch.allclose(got.double(), ref, rtol=tol, atol=tol * float(ref.abs().max())), (dt, hm)


# ---- the source-blocked kernels, one row per wave-instruction for every weight mode (rows of 512 / 256 bytes) -----------------------
def _dense_graph(rng, nodes, nnz):
    di = powerlaw_index(nnz, nodes, nodes + 1)
    di[: nnz // 20] = di[nnz // 20]                                     # a hub that is split (carry slots)
    di = np.sort(di)
    di[di == 7] = 8                                                     # a destination without edges
    si = rng.integers(0, nodes, nnz).astype(np.int64)
    return si, di


@pytest.mark.parametrize("dtype,F", [(torch.float32, 128), (torch.float32, 64), (torch.bfloat16, 256), (torch.float16, 128)])
def test_row_per_wave_kernels_every_weight_mode_and_reduction(geot, oracle, dtype, F, request):
    """seg_slab_wrow_kernel for weight modes 0 / 1 / 4 - a plan cut into WAVES runs one row per wave-instruction under any weight
    mode (the library's own rule keeps lane groups for these modes: measured faster, profiles/r05/slab_cases__row_per_wave_*): sum /
    mean / max / min against the oracle in float64, split hub, empty row, deterministic; edge-order weights staged inside the kernel
    and read through the permutation give the same bits."""
    from geot_amd import slab
    geot.hip.set_option("slab_spmm_mfma", 0)                            # (this test is about the row-per-wave kernel; the matrix-core kernel
    request.addfinalizer(lambda: geot.hip.set_option("slab_spmm_mfma", 1))   #  that serves 16-bit 512-byte plans by default: tests/test_gpu_round6.py)
    rng = np.random.default_rng(F)
    nodes, nnz = 3000, 400_000
    si, di = _dense_graph(rng, nodes, nnz)
    esz = 4 if dtype == torch.float32 else 2
    ulp = 1e-5 if dtype == torch.float32 else (2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7)
    scale = 1.0 / 16 if dtype == torch.float16 else 1.0
    x = torch.from_numpy(rng.random((nodes, F), dtype=np.float32) * scale).to(dtype)
    w = torch.from_numpy(rng.random(nnz, dtype=np.float32)).to(dtype)
    d_si, d_di, d_x, d_w = dev(si), dev(di), x.cuda(), w.cuda()
    R = slab.rows_per_group(1, 1, dtype, F * esz)
    plan = slab.build_plan(d_si, d_di, nodes, nodes, F * esz, 1, 1, rows_per_group=R, units=slab._lib.load().geot_slab_units())
    assert plan.meta["units"] == slab._lib.load().geot_slab_units() and plan.meta["split_rows"] >= 1
    out = torch.empty(nodes, F, device="cuda", dtype=dtype)

    def check(got, hi, what):
        got = got.float().cpu().numpy()
        bound = ulp * np.abs(hi) + 1e-6 if dtype != torch.float32 else 1e-5 * np.abs(hi).max()
        assert np.all(np.abs(got - hi) <= bound), (what, float(np.max(np.abs(got - hi))))
        assert np.all(got[hi == 0] == 0), what

    xf, wf = x.float().numpy(), w.float().numpy()
    slab.slab_spmm_out(plan, d_w, 1, d_x, out, 1, F)
    assert "seg_slab_wrow_kernel" in geot.hip.last_kernel(), geot.hip.last_kernel()
    check(out, oracle.gather_weight_scatter(si, di, wf, xf, rows=nodes, acc64=True), "gws")
    again = torch.empty_like(out)
    slab.slab_spmm_out(plan, d_w, 1, d_x, again, 1, F)
    assert torch.equal(out, again)
    slab.slab_spmm_out(plan, d_w, 1, d_x, again, 1, F, stage_weights=False)   # weights read through e_perm in the row loop: same bits
    assert torch.equal(out, again)
    wp = d_w[plan.tensors["e_perm"].long()].contiguous()                # the weight in plan order (mode 4): same bits
    slab.slab_spmm_out(plan, wp, 4, d_x, again, 1, F)
    assert torch.equal(out, again)
    slab.slab_spmm_out(plan, None, 0, d_x, out, 1, F)
    check(out, oracle.gather_scatter(si, di, xf, rows=nodes, acc64=True), "gs")
    # the library's own plan for these modes (lane groups): staged and permuted weight reads agree there too
    lg = slab.build_plan(d_si, d_di, nodes, nodes, F * esz, 1, 1, rows_per_group=slab.rows_per_group(1, 1, dtype, F * esz))
    assert lg.meta["units"] == slab._lib.load().geot_slab_units_for(1, F * esz)
    a, b = torch.empty_like(out), torch.empty_like(out)
    slab.slab_spmm_out(lg, d_w, 1, d_x, a, 1, F)
    assert "seg_slab_kernel" in geot.hip.last_kernel(), geot.hip.last_kernel()
    slab.slab_spmm_out(lg, d_w, 1, d_x, b, 1, F, stage_weights=False)
    assert torch.equal(a, b)
    check(a, oracle.gather_weight_scatter(si, di, wf, xf, rows=nodes, acc64=True), "gws, lane groups, staged weights")
    for red, tred in (("max", "amax"), ("min", "amin"), ("mean", "mean")):
        for weighted in (False, True):
            slab.slab_spmm_out(plan, d_w if weighted else None, 1 if weighted else 0, d_x, out, 1, F, reduce=red)
            msg = d_x[d_si].float() * (d_w.float()[:, None] if weighted else 1.0)
            want = torch.zeros(nodes, F, device="cuda").scatter_reduce(0, d_di[:, None].expand(-1, F), msg, tred, include_self=False)
            check(out, want.cpu().numpy().astype(np.float64), (red, weighted))


@pytest.mark.parametrize("dtype,H,Fh", [(torch.float32, 1, 32), (torch.bfloat16, 1, 64), (torch.float32, 4, 8), (torch.float16, 2, 32),
                                         (torch.float32, 1, 64), (torch.float32, 4, 64)])
def test_weights_in_every_order_on_every_chunk_length(geot, dtype, H, Fh):
    """The three ways a weight reaches the row loop - through e_perm inside it, staged by the pre-pass, given in plan order - on plans
    whose chunks are ONE batch long (rows of 128 bytes: 8 edges a chunk, where the edge-order weights of the next chunk cannot be
    fetched 'a batch later'), two (256 bytes) and eight (whole-wave rows): equal bits, and the float64 sums."""
    from geot_amd import slab
    rng = np.random.default_rng(7 * H + Fh)
    nodes, nnz = 2000, 250_000
    si, di = _dense_graph(rng, nodes, nnz)
    esz = 4 if dtype == torch.float32 else 2
    rowbytes = H * Fh * esz
    mh = H > 1
    x = torch.from_numpy(rng.random((nodes, H, Fh), dtype=np.float32)).to(dtype).cuda()
    w = torch.from_numpy(rng.random((nnz, H) if mh else (nnz,), dtype=np.float32)).to(dtype).cuda()
    d_si, d_di = dev(si), dev(di)
    wm = 2 if mh else 1
    plan = slab.build_plan(d_si, d_di, nodes, nodes, rowbytes, wm, H, rows_per_group=slab.rows_per_group(wm, H, dtype, rowbytes))
    outs = []
    for weight, mode, kw in ((w, wm, dict(stage_weights=False)), (w, wm, {}), (w[plan.tensors["e_perm"].long()].contiguous(), wm + 3, {})):
        o = torch.full((nodes, H, Fh), float("nan"), device="cuda", dtype=dtype)
        slab.slab_spmm_out(plan, weight, mode, x if mh else x.view(nodes, Fh), o if mh else o.view(nodes, Fh), H, Fh, **kw)
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), geot.hip.last_kernel()
    wd = w.double() if mh else w.double()[:, None]
    ref = torch.zeros(nodes, H, Fh, device="cuda", dtype=torch.float64).index_add_(0, d_di, x.double()[d_si] * wd[:, :, None])
    tol = 1e-5 if dtype == torch.float32 else (2.0 ** -9 if dtype == torch.float16 else 2.0 ** -6)
    assert float((outs[0].double() - ref).abs().max()) <= tol * float(ref.abs().max())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("H,Fh", [(4, 64), (1, 256), (2, 128), (8, 32), (1, 128), (2, 64), (4, 32), (8, 64), (4, 128), (2, 256)])   # rows of 512 and (round 6) of 256 and 1024 bytes
def test_matrix_core_sddmm_is_exact_on_integer_data(geot, dtype, H, Fh):
    """seg_slab_sddmm_mfma_kernel (16-bit multi-head SDDMM over a plan of 512-byte rows, v_mfma_f32_16x16x32): features in {-1, 0, 1}, so every dot product is an integer below 2^8 - exact in fp32 and in the 16-bit result -
    and any slip in the operand maps (which lane holds which features of which edge / row, where D[m][dl(m)] sits, the padded LDS
    image) shows as a wrong integer, not as rounding.  Plans with at most 16 rows per group (one 16-column operand), a hub split into
    pieces, rows without edges, out-of-range sources; results in plan order and in edge order; against float64 and against the
    row-per-wave kernel (option slab_sddmm_mfma = 0)."""
    from geot_amd import slab
    rng = np.random.default_rng(31 * H + Fh)
    nodes, nnz = 2500, 300_000
    si, di = _dense_graph(rng, nodes, nnz)
    si[rng.integers(0, nnz, 50)] = nodes + 7                            # out-of-range sources: the dot is 0
    rowbytes = H * Fh * 2
    q = torch.from_numpy(rng.integers(-1, 2, (nodes, H, Fh)).astype(np.float32)).to(dtype).cuda()
    k = torch.from_numpy(rng.integers(-1, 2, (nodes, H, Fh)).astype(np.float32)).to(dtype).cuda()
    d_si, d_di = dev(si), dev(di)
    R = slab.rows_per_group(2, H, dtype, rowbytes)                      # (the rule keeps 16-bit plans within a matrix-core operand's 16 rows)
    plan = slab.build_plan(d_si, d_di, nodes, nodes, rowbytes, 2, H, rows_per_group=R)
    assert plan.meta["split_rows"] >= 1 and plan.meta["rows_per_group"] <= 16
    ok = dev((si < nodes))
    want = (q.double()[d_di] * k.double()[d_si.clamp(max=nodes - 1)]).sum(-1) * ok[:, None]
    assert float(want.abs().max()) <= 256
    outs = {}
    try:
        for mfma in (1, 0):
            geot.hip.set_option("slab_sddmm_mfma", mfma)
            if not mfma and rowbytes == 1024:                           # a plan of 16 rows of 1 KiB is cut for the matrix cores: no vector-ALU form
                with pytest.raises(RuntimeError, match="matrix-core"):
                    slab.slab_mh_sddmm_out(plan, q, k, None)
                outs[0] = outs[1]
                continue
            s_plan = slab.slab_mh_sddmm_out(plan, q, k, None)
            assert ("seg_slab_sddmm_mfma_kernel" in geot.hip.last_kernel()) == bool(mfma), geot.hip.last_kernel()
            s_edge = torch.full((nnz, H), float("nan"), device="cuda", dtype=dtype)
            slab.slab_mh_sddmm_out(plan, q, k, s_edge)
            assert torch.equal(s_plan, s_edge[plan.tensors["e_perm"].long()])
            assert torch.equal(s_edge.double(), want), (mfma, float((s_edge.double() - want).abs().max()))
            outs[mfma] = s_edge
    finally:
        geot.hip.set_option("slab_sddmm_mfma", 1)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("dtype,H,Fh", [(torch.float32, 4, 64), (torch.float32, 4, 32), (torch.bfloat16, 4, 64), (torch.float32, 2, 32),
                                         (torch.float16, 8, 16)])
def test_plan_order_attention_pipeline(geot, dtype, H, Fh):
    """SDDMM -> (anything elementwise) -> SpMM without ever leaving the plan's edge order: geot_slab_mh_sddmm with out = NULL leaves
    the scores in plan order, geot_slab_spmm's weight_mode 5 reads them back without the permutation.  Equal to the edge-order
    pipeline (the unstaged scores, weight_mode 2) bit for bit, and to float64 within rounding."""
    from geot_amd import slab
    rng = np.random.default_rng(H * Fh)
    nodes, nnz = 2500, 300_000
    si, di = _dense_graph(rng, nodes, nnz)
    esz = 4 if dtype == torch.float32 else 2
    rowbytes = H * Fh * esz
    q = (torch.from_numpy(rng.standard_normal((nodes, H, Fh)).astype(np.float32)) / 8).to(dtype).cuda()
    k = (torch.from_numpy(rng.standard_normal((nodes, H, Fh)).astype(np.float32)) / 8).to(dtype).cuda()
    v = torch.from_numpy(rng.random((nodes, H, Fh), dtype=np.float32)).to(dtype).cuda()
    d_si, d_di = dev(si), dev(di)
    plan = slab.build_plan(d_si, d_di, nodes, nodes, rowbytes, 2, H, rows_per_group=slab.rows_per_group(2, H, dtype, rowbytes))
    assert plan.meta["split_rows"] >= 1
    # scores in edge order (staged + unstaged) against float64 and against the per-edge kernel
    s_edge = torch.full((nnz, H), float("nan"), device="cuda", dtype=dtype)
    slab.slab_mh_sddmm_out(plan, q, k, s_edge)
    want = (q.double()[d_di] * k.double()[d_si]).sum(-1)
    tol = 1e-5 if dtype == torch.float32 else (2.0 ** -9 if dtype == torch.float16 else 2.0 ** -6)
    assert float((s_edge.double() - want).abs().max()) <= tol * max(1.0, float(want.abs().max()))
    per_edge = torch.empty_like(s_edge)
    geot.hip.mh_sddmm_coo_out(d_si, d_di, q, k, per_edge, False)
    assert float((per_edge.double() - want).abs().max()) <= tol * max(1.0, float(want.abs().max()))
    # the same scores left in plan order
    s_plan = slab.slab_mh_sddmm_out(plan, q, k, None)
    assert torch.equal(s_plan, s_edge[plan.tensors["e_perm"].long()])
    # elementwise work in plan order, then the SpMM in both orders: bit-equal
    a_plan, a_edge = torch.exp(s_plan.float()).to(dtype), torch.exp(s_edge.float()).to(dtype)
    out_plan = torch.empty(nodes, H, Fh, device="cuda", dtype=dtype)
    out_edge = torch.empty_like(out_plan)
    slab.slab_spmm_out(plan, a_plan, 5, v, out_plan, H, Fh)
    slab.slab_spmm_out(plan, a_edge, 2, v, out_edge, H, Fh)
    assert torch.equal(out_plan, out_edge)
    slab.slab_spmm_out(plan, a_edge, 2, v, out_edge, H, Fh, stage_weights=False)
    assert torch.equal(out_plan, out_edge)
    ref = torch.zeros(nodes, H, Fh, device="cuda", dtype=torch.float64).index_add_(0, d_di, v.double()[d_si] * a_edge.double()[:, :, None])
    otol = 1e-5 if dtype == torch.float32 else (2.0 ** -9 if dtype == torch.float16 else 2.0 ** -6)
    assert float((out_plan.double() - ref).abs().max()) <= otol * float(ref.abs().max())


@pytest.mark.parametrize("dtype,rows_log2", [(torch.float32, 22), (torch.bfloat16, 23)])
def test_tables_just_below_4_gib_use_the_top_of_the_32_bit_row_offsets(geot, dtype, rows_log2):
    """The wave-row kernels address a gathered row as (buffer base) + a 32-bit scalar row offset + the lane's offset, and an edge's
    (source row, row in group) travel as one word: a table of 2^22 - 1 rows of 1 KiB (fp32 H=4 x F=64) / 2^23 - 1 rows of 512 bytes
    (bf16) is 4 GiB minus one row - the largest the source-blocked kernels accept - with most sources in its last rows (bit 31 of the
    offset set, the top bits of the packed word used).  SpMM and multi-head SDDMM (the matrix-core kernel for bf16) against float64;
    and the library refuses one row more."""
    from geot_amd import slab
    H, Fh = 4, 64
    nodes = (1 << rows_log2) - 1
    esz = 4 if dtype == torch.float32 else 2
    rowbytes = H * Fh * esz
    assert nodes * rowbytes < 2 ** 32 <= (nodes + 1) * rowbytes
    rng = np.random.default_rng(rows_log2)
    nnz, rows_used = 400_000, 1500
    di = np.sort(rng.integers(0, rows_used, nnz)).astype(np.int64)
    si = np.where(rng.random(nnz) < 0.8, rng.integers(nodes - 40_000, nodes, nnz), rng.integers(0, nodes, nnz)).astype(np.int64)
    si[:64] = nodes - 1                                                 # the very last row
    d_si, d_di = dev(si), dev(di)
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand(nodes, H, Fh, device="cuda", generator=gen).to(dtype)
    w = torch.rand(nnz, H, device="cuda", generator=gen).to(dtype)
    plan = slab.build_plan(d_si, d_di, rows_used, nodes, rowbytes, 2, H, rows_per_group=slab.rows_per_group(2, H, dtype, rowbytes))
    out = torch.full((rows_used, H, Fh), float("nan"), device="cuda", dtype=dtype)
    slab.slab_spmm_out(plan, w, 2, x, out, H, Fh)
    ref = torch.zeros(rows_used, H, Fh, device="cuda", dtype=torch.float64).index_add_(0, d_di, x[d_si].double() * w.double()[:, :, None])
    tol = 1e-5 if dtype == torch.float32 else 2.0 ** -6
    assert float((out.double() - ref).abs().max()) <= tol * float(ref.abs().max()), geot.hip.last_kernel()
    q = (torch.rand(rows_used, H, Fh, device="cuda", generator=gen) - 0.5).to(dtype)
    s_edge = torch.full((nnz, H), float("nan"), device="cuda", dtype=dtype)
    slab.slab_mh_sddmm_out(plan, q, x, s_edge)
    if dtype == torch.bfloat16:
        assert "seg_slab_sddmm_mfma_kernel" in geot.hip.last_kernel(), geot.hip.last_kernel()
    want = (q.double()[d_di] * x[d_si].double()).sum(-1)
    assert float((s_edge.double() - want).abs().max()) <= tol * max(1.0, float(want.abs().max()))
    del x
    x1 = torch.empty(nodes + 1, H, Fh, device="cuda", dtype=dtype)         # one row more: 4 GiB exactly - refused, not wrapped
    with pytest.raises(RuntimeError, match="4 GiB"):
        slab.slab_spmm_out(plan, w, 2, x1, out, H, Fh)



def test_hang_hunt_quick(tmp_path):
    """tools/hang_hunt.py stays in the suite's orbit: ten fresh-process runs of round 3's three-thread
    scenario (three workers on one dense graph, persistent source-blocked grids in flight) under the watchdog, five with the
    library's turn-taking and five without - every run finishes, none is killed."""
    import json
    import os
    import subprocess
    import sys
    for turn in (1, 0):
        out = tmp_path / f"turn{turn}"
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hang_hunt.py"), "--scenario", "threads", "--runs", "5", "--slab-turn", str(turn),
                            "--T", "60", "--iters", "10", "--out", str(out)], capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
        summary = json.load(open(out / f"threads_turn{turn}_guard1_always_summary.json"))
        assert summary["hung"] is None and summary.get("failed") is None and len(summary["runs"]) == 5, summary
        assert all(r["rc"] == 0 and not r["hung"] for r in summary["runs"]), summary["runs"]
