"""geot_amd.Graph (`-m gpu`): the opt-in static-graph handle gives the operators' results - forward and backward - without the host
layer's caches and content guard, and keeps attention scores in the plan's edge order from the SDDMM to the SpMM.

Reference semantics: csrc/util/check.cuh:90-111 (gather / gws), test/test_mh_spmm.py:4-10 (multi-head), backward pattern
geot/gather_weight_scatter.py:31-51.
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


def _graph(nodes, nnz, seed, hub=True):
    rng = np.random.default_rng(seed)
    di = powerlaw_index(nnz, nodes, seed)
    if hub:
        di[: nnz // 20] = di[nnz // 20]
        di = np.sort(di)
    di[di == 7] = 8
    di[-1] = nodes - 1
    si = rng.integers(0, nodes - 3, nnz).astype(np.int64)          # the last source nodes have no out-edge
    return dev(si), dev(di)


@pytest.mark.parametrize("mode", ["never", "always"])
def test_handle_gives_the_operators_results_forward_and_backward(geot, mode):
    """gather_scatter / gather_weight_scatter / mh_spmm / sddmm through the handle against plain-torch float64 autograd - on the per-edge
    kernels (slab_mode 'never') and over the handle's own plans (forward + transposed: 'always')."""
    nodes, nnz, F, H = 3000, 400_000, 128, 4
    si, di = _graph(nodes, nnz, 5)
    g = geot.Graph(si, di, num_src=nodes, num_dst=nodes, slab_mode=mode)
    x = torch.rand(nodes, F, device="cuda", requires_grad=True)
    w = torch.rand(nnz, device="cuda", requires_grad=True)
    xr, wr = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()

    def dense(si, di, w, x, rows):
        return torch.zeros(rows, *x.shape[1:], device="cuda", dtype=x.dtype).index_add_(0, di, x[si] * w.view(-1, *([1] * (x.dim() - 1))) if w.dim() == 1 else x[si] * w[:, :, None])

    y = g.gather_weight_scatter(w, x)
    up = torch.rand_like(y)
    gx, gw = torch.autograd.grad(y, [x, w], up)
    ref = dense(si, di, wr, xr, nodes)
    rgx, rgw = torch.autograd.grad(ref, [xr, wr], up.double())
    for a, b in ((y, ref), (gx, rgx), (gw, rgw)):
        assert a.shape == b.shape
        assert float((a.double() - b).abs().max()) <= 2e-5 * float(b.abs().max())
    # the operator of the same name agrees
    y_op = geot.gather_weight_scatter(si, di, w.detach(), x.detach())
    assert float((y.detach() - y_op).abs().max()) <= 2e-5 * float(y_op.abs().max())
    # no weight, mean
    ym = g.gather_scatter(x, "mean")
    (gxm,) = torch.autograd.grad(ym, [x], up)
    deg = torch.zeros(nodes, device="cuda", dtype=torch.float64).index_add_(0, di, torch.ones(nnz, device="cuda", dtype=torch.float64)).clamp(min=1)
    refm = torch.zeros(nodes, F, device="cuda", dtype=torch.float64).index_add_(0, di, xr[si]) / deg[:, None]
    (rgxm,) = torch.autograd.grad(refm, [xr], up.double())
    assert float((ym.double() - refm).abs().max()) <= 2e-5 * float(refm.abs().max())
    assert float((gxm.double() - rgxm).abs().max()) <= 2e-5 * float(rgxm.abs().max())
    # multi-head
    xh = torch.rand(nodes, H, F // 2, device="cuda", requires_grad=True)
    wh = torch.rand(nnz, H, device="cuda", requires_grad=True)
    yh = g.mh_spmm(wh, xh)
    uph = torch.rand_like(yh)
    gxh, gwh = torch.autograd.grad(yh, [xh, wh], uph)
    xhr, whr = xh.detach().double().requires_grad_(), wh.detach().double().requires_grad_()
    refh = dense(si, di, whr, xhr, nodes)
    rgxh, rgwh = torch.autograd.grad(refh, [xhr, whr], uph.double())
    for a, b in ((yh, refh), (gxh, rgxh), (gwh, rgwh)):
        assert a.shape == b.shape
        assert float((a.double() - b).abs().max()) <= 2e-5 * float(b.abs().max())
    # static coefficients: permuted once for the table the calls will read, the same numbers as the edge-order call
    wp = g.plan_order(wh.detach(), xh.detach())
    assert isinstance(wp, geot.PlanOrdered) == (mode == "always")
    assert torch.equal(g.mh_spmm(wp, xh.detach()), yh.detach())
    w1p = g.plan_order(w.detach(), x.detach())
    assert torch.equal(g.gather_weight_scatter(w1p, x.detach()), y.detach())
    # sddmm with its own backward
    m1 = torch.rand(nodes, F, device="cuda", requires_grad=True)
    s = g.sddmm(m1, x)
    ups = torch.rand_like(s)
    g1, g2 = torch.autograd.grad(s, [m1, x], ups)
    m1r = m1.detach().double().requires_grad_()
    sr = (m1r[di] * xr[si]).sum(-1)
    r1, r2 = torch.autograd.grad(sr, [m1r, xr], ups.double())
    for a, b in ((s, sr), (g1, r1), (g2, r2)):
        assert a.shape == b.shape
        assert float((a.double() - b).abs().max()) <= 5e-5 * float(b.abs().max())
    if mode == "always":
        assert g.stats["plans_built"] >= 2 and g.stats["plan_launches"] > 0
    else:
        assert g.stats["plans_built"] == 0 and g.stats["plan_launches"] == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_layer_stays_in_plan_order(geot, dtype):
    """scores = SDDMM(q, k) in plan order -> leaky_relu / exp / per-row normalisation on .values (keys: .dst) -> SpMM, backward through
    all of it: every per-edge tensor lives in the plan's order, nothing passes through the edge permutation except the handle's own
    one-off index maps.  Against the same layer written in edge order with plain torch, float64."""
    nodes, nnz, H, Fh = 2500, 300_000, 4, 64
    si, di = _graph(nodes, nnz, 9)
    g = geot.Graph(si, di, num_src=nodes, num_dst=nodes, slab_mode="always")
    gen = torch.Generator(device="cuda").manual_seed(3)
    q = (torch.randn(nodes, H, Fh, device="cuda", generator=gen) / 8).to(dtype).requires_grad_()
    k = (torch.randn(nodes, H, Fh, device="cuda", generator=gen) / 8).to(dtype).requires_grad_()
    v = torch.rand(nodes, H, Fh, device="cuda", generator=gen).to(dtype).requires_grad_()

    s = g.mh_sddmm(q, k, plan_order=True)
    assert isinstance(s, geot.PlanOrdered) and s.values.shape == (nnz, H)
    e = torch.exp(torch.nn.functional.leaky_relu(s.values.float(), 0.2))
    denom = torch.zeros(nodes, H, device="cuda").index_add_(0, s.dst, e)
    a = s.with_values((e / denom[s.dst]).to(dtype))
    y = g.mh_spmm(a, v)
    up = torch.rand_like(y)
    gq, gk, gv = torch.autograd.grad(y, [q, k, v], up)

    qr, kr, vr = (t.detach().double().requires_grad_() for t in (q, k, v))
    sr = (qr[di] * kr[si]).sum(-1)
    er = torch.exp(torch.nn.functional.leaky_relu(sr, 0.2))
    dr = torch.zeros(nodes, H, device="cuda", dtype=torch.float64).index_add_(0, di, er)
    ar = er / dr[di]
    yr = torch.zeros(nodes, H, Fh, device="cuda", dtype=torch.float64).index_add_(0, di, vr[si] * ar[:, :, None])
    rq, rk, rv = torch.autograd.grad(yr, [qr, kr, vr], up.double())
    tol = 5e-5 if dtype == torch.float32 else 0.03
    for name, a_, b_ in (("y", y, yr), ("dq", gq, rq), ("dk", gk, rk), ("dv", gv, rv)):
        assert a_.shape == b_.shape, name
        assert float((a_.double() - b_).abs().max()) <= tol * float(b_.abs().max()), name
    # the scores leave the plan's order on request, and come back
    back = g.plan_order(s.edge_order().detach(), s)
    assert torch.equal(back.values, s.values.detach())


def test_handle_asks_the_host_layer_nothing(geot):
    """No cache lookup, no fingerprint: the host layer's counters stand still while the handle works; and the handle refuses an
    edge list it cannot own (dst_index with descents)."""
    from geot_amd import ops
    nodes, nnz, F = 2000, 200_000, 64
    si, di = _graph(nodes, nnz, 13, hub=False)
    g = geot.Graph(si, di, num_src=nodes, num_dst=nodes, slab_mode="always")
    x = torch.rand(nodes, F, device="cuda", requires_grad=True)
    w = torch.rand(nnz, device="cuda")
    before = ops.stats()
    for _ in range(3):
        g.gather_weight_scatter(w, x).sum().backward()
    after = ops.stats()
    for key in ("guard_checks", "probes", "transposes", "plans_built", "slab_calls"):
        assert after[key] == before[key], key
    assert g.src_index.data_ptr() != si.data_ptr() and g.dst_index.data_ptr() != di.data_ptr()      # its own clones
    bad = di.clone()
    bad[5], bad[6] = bad[6] + 3, bad[5]
    with pytest.raises(ValueError, match="ascending"):
        geot.Graph(si, bad)


@pytest.mark.parametrize("mode", ["never", "always"])
def test_backward_without_num_src_when_the_last_nodes_have_no_out_edge(geot, mode):
    """README's `geot.Graph(col, row)`: no num_src, so the handle counts max(src_index) + 1 source rows - fewer than x has when the
    trailing nodes have no out-edge.  The gradient must still have x's shape (zero rows for those nodes); round 5's advisor found
    autograd rejecting it ("invalid gradient at index 3")."""
    nodes, nnz, F, H = 3000, 300_000, 128, 4
    si, di = _graph(nodes, nnz, 9)                                   # sources stop at nodes - 4
    g = geot.Graph(si, di, slab_mode=mode)                           # neither num_src nor num_dst
    assert g.src_rows < nodes
    for shape, weight in (((nodes, F), None), ((nodes, F), torch.rand(nnz, device="cuda", requires_grad=True)),
                          ((nodes, H, F // H), torch.rand(nnz, H, device="cuda", requires_grad=True))):
        x = torch.rand(*shape, device="cuda", requires_grad=True)
        if weight is None:
            y = g.gather_scatter(x)
        elif weight.dim() == 1:
            y = g.gather_weight_scatter(weight, x)
        else:
            y = g.mh_spmm(weight, x)
        up = torch.rand_like(y)
        (gx,) = torch.autograd.grad(y, [x], up)
        assert gx.shape == x.shape and float(gx[g.src_rows:].abs().max()) == 0.0
        xr = x.detach().double().requires_grad_()
        msg = xr[si] if weight is None else (xr[si] * weight.detach().double().view(nnz, *([1] * (x.dim() - 1))) if weight.dim() == 1
                                             else xr[si] * weight.detach().double()[:, :, None])
        ref = torch.zeros(y.shape[0], *x.shape[1:], device="cuda", dtype=torch.float64).index_add_(0, di, msg)
        (rgx,) = torch.autograd.grad(ref, [xr], up.double())
        assert float((gx.double() - rgx).abs().max()) <= 2e-5 * float(rgx.abs().max())
    # the SDDMM's operands likewise (both may be longer than the graph's row counts)
    q = torch.rand(nodes + 5, F, device="cuda", requires_grad=True)
    k = torch.rand(nodes + 2, F, device="cuda", requires_grad=True)
    s = g.sddmm(q, k)
    gq, gk = torch.autograd.grad(s, [q, k], torch.rand_like(s))
    assert gq.shape == q.shape and gk.shape == k.shape


def test_plan_order_is_always_a_plan_ordered_and_many_heads_fall_back(geot):
    """`plan_order=True` returns a PlanOrdered whatever the graph: over no plan (a sparse graph, slab_mode 'never', an unsupported
    row width) its values are in edge order and the module's usage example still runs.  More than 16 heads: no plan, the per-edge
    kernel serves the call (the host operator's rule) instead of GEOT_EUNSUPPORTED surfacing from the plan kernel."""
    nodes, nnz, H, F = 2000, 200_000, 4, 64
    si, di = _graph(nodes, nnz, 11)
    q, k, v = (torch.rand(nodes, H, F, device="cuda") / 8 for _ in range(3))
    ref = None
    for mode in ("always", "never"):
        g = geot.Graph(si, di, num_src=nodes, num_dst=nodes, slab_mode=mode)
        s = g.mh_sddmm(q, k, plan_order=True)
        assert isinstance(s, geot.graph.PlanOrdered) and (s.plan is None) == (mode == "never")
        assert s.dst.shape == (nnz,) and s.src.shape == (nnz,)
        a = s.with_values(torch.exp(s.values))                        # the usage example of geot_amd/graph.py
        y = g.mh_spmm(a, v)
        e = s.edge_order()
        if ref is None:
            ref = (y, e)
        else:
            assert torch.allclose(y, ref[0], rtol=2e-5, atol=1e-6) and torch.allclose(e, ref[1], rtol=2e-5, atol=1e-6)
        with pytest.raises(RuntimeError):                              # [nnz, H] values where one value per edge is expected
            g.gather_weight_scatter(s, v[:, 0])
    H2, F2 = 32, 4                                                     # 512-byte fp32 rows, 32 heads
    g = geot.Graph(si, di, num_src=nodes, num_dst=nodes, slab_mode="always")
    x = torch.rand(nodes, H2, F2, device="cuda")
    w = torch.rand(nnz, H2, device="cuda")
    y = g.mh_spmm(w, x)
    want = torch.zeros(nodes, H2, F2, device="cuda", dtype=torch.float64).index_add_(0, di, x[si].double() * w.double()[:, :, None])
    assert float((y.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,cols", [(torch.bfloat16, 8), (torch.float32, 4), (torch.float16, 1), (torch.float32, 0)])
def test_per_edge_values_are_permuted_by_the_library(geot, dtype, cols):
    """graph._rows_at: rows of a per-edge value tensor at an int64 index through geot_gather_rows - what PlanOrdered, edge_order() and
    the handle's backward passes use instead of torch's advanced indexing (which returned garbage for the last 2^26 rows of a
    [115 M, 8] bfloat16 tensor on this stack: tests/test_gpu_round6.py::test_cfg4_full_size_bf16_properties[plan-8-64] found it).
    Against torch's indexing at a size where that is right, [nnz] and [nnz, H]; and the scatter form through the inverse permutation."""
    from geot_amd import graph
    n = 300_000
    gen = torch.Generator(device="cuda").manual_seed(7)
    v = torch.rand((n, cols) if cols else (n,), device="cuda", generator=gen).to(dtype)
    perm = torch.randperm(n, device="cuda", generator=gen)
    got = graph._rows_at(v, perm)
    assert got.shape == v.shape and got.dtype == v.dtype and torch.equal(got, v[perm])
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n, device="cuda")
    want = torch.empty_like(v)
    want[perm] = v                                                      # the scatter ...
    assert torch.equal(graph._rows_at(v, inv), want)                    # ... is the gather through the inverse
