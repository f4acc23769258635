"""FX rewrite (row f2): graphs are rewritten on CPU (no kernels run); execution parity on the GPU."""
import numpy as np
import pytest
import torch

from conftest import powerlaw_index


class GCNLike(torch.nn.Module):
    def __init__(self, fin, fout):
        super().__init__()
        self.lin = torch.nn.Linear(fin, fout, bias=False)

    def forward(self, x, edge_index, edge_weight):
        row, col = edge_index[0], edge_index[1]
        h = self.lin(x)
        msg = h.index_select(0, col) * edge_weight.unsqueeze(-1)
        agg = h.new_zeros(h.shape).index_add(0, row, msg)                 # -> gather_weight_scatter
        plain = h.new_zeros(h.shape).index_add(0, row, h.index_select(0, col))   # -> gather_scatter
        return torch.relu(agg) + plain


class GATLike(torch.nn.Module):
    def forward(self, x3, edge_index, alpha):                              # x3 [N,H,F], alpha [nnz,H]
        row, col = edge_index[0], edge_index[1]
        msg = alpha.unsqueeze(-1) * x3.index_select(0, col)
        return torch.zeros_like(x3).index_add(0, row, msg)                 # -> mh_spmm


class NotZeros(torch.nn.Module):
    def forward(self, x, edge_index):
        row, col = edge_index[0], edge_index[1]
        return x.index_add(0, row, x.index_select(0, col))                 # accumulates into x: must NOT be rewritten


def _inputs(n=50, nnz=400, f=8, h=None, seed=0, device="cpu", sorted_rows=True):
    rng = np.random.default_rng(seed)
    row = powerlaw_index(nnz, n, seed) if sorted_rows else rng.integers(0, n, nnz)
    ei = torch.from_numpy(np.stack([row, rng.integers(0, n, nnz)]).astype(np.int64)).to(device)
    if h is None:
        return torch.rand(n, f, device=device), ei, torch.rand(nnz, device=device)
    return torch.rand(n, h, f, device=device), ei, torch.rand(nnz, h, device=device)


def _targets(ep):
    return [n.target for n in ep.graph_module.graph.nodes if n.op == "call_function"]


def test_rewrite_on_cpu_graphs():
    from geot_amd.match_replace import pattern_transform
    ep = pattern_transform(GCNLike(8, 8), _inputs())
    t = _targets(ep)
    assert ep.geot_fused_nodes == 2
    assert torch.ops.geot.gather_weight_scatter_rows.default in t and torch.ops.geot.gather_scatter_rows.default in t
    assert torch.ops.aten.index_add.default not in t
    ep = pattern_transform(GATLike(), _inputs(h=4))
    assert ep.geot_fused_nodes == 1 and torch.ops.geot.mh_spmm_rows.default in _targets(ep)
    x, ei, _ = _inputs()
    ep = pattern_transform(NotZeros(), (x, ei))
    assert ep.geot_fused_nodes == 0 and torch.ops.aten.index_add.default in _targets(ep)
    ep = pattern_transform(GCNLike(8, 8), _inputs(), sort_edges=True)
    assert torch.ops.aten.sort.stable in _targets(ep)


@pytest.mark.gpu
@pytest.mark.parametrize("sort_edges", [False, True])
def test_rewritten_models_match_eager_on_gpu(sort_edges):
    from geot_amd.match_replace import pattern_transform
    torch.manual_seed(0)
    for model, kw in ((GCNLike(16, 32), dict(n=3000, nnz=60_000, f=16)), (GATLike(), dict(n=2000, nnz=50_000, f=16, h=4))):
        model = model.cuda()
        args = _inputs(device="cuda", seed=3, sorted_rows=not sort_edges, **kw)
        ep = pattern_transform(model, args, sort_edges=sort_edges)
        assert ep.geot_fused_nodes >= 1
        got = ep.module()(*args)
        ref = model(*args)
        assert got.shape == ref.shape                                       # row count preserved (dst.shape[0])
        assert torch.allclose(got, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
def test_ops_run_under_torch_compile():
    """The dispatcher ops carry fake/meta implementations, so torch.compile can trace through call sites
    (the reference's test/compile scripts do the same with its ops)."""
    import geot_amd as geot
    x, ei, w = _inputs(n=4000, nnz=80_000, f=32, device="cuda", seed=5)
    row, col = ei[0].contiguous(), ei[1].contiguous()

    def f(x, w):
        h = geot.gather_weight_scatter(col, row, w, x)
        return torch.relu(h) * 2.0

    ref = f(x, w)
    out = torch.compile(f)(x, w)
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5)

    from geot_amd.match_replace import pattern_transform
    ep = pattern_transform(GCNLike(32, 32).cuda(), (x, ei, w))
    mod = ep.module()
    out2 = torch.compile(mod)(x, ei, w)
    assert torch.allclose(out2, mod(x, ei, w), rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
def test_rewritten_model_is_trainable():
    """Gradients through the fused ops of a rewritten graph equal the eager model's."""
    from geot_amd.match_replace import pattern_transform
    torch._dynamo.reset()          # (this test exports right after the torch.compile test: start from a clean tracer state)
    torch.manual_seed(1)
    model = GCNLike(16, 16).cuda()
    x, ei, w = _inputs(n=1500, nnz=30_000, f=16, device="cuda", seed=7)
    x1, w1 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    x2, w2 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ep = pattern_transform(model, (x, ei, w))
    ep.module()(x1, ei, w1).square().sum().backward()
    model(x2, ei, w2).square().sum().backward()
    # (the gradients are sums of terms of both signs with magnitudes in the thousands: an element that cancels to ~0 carries the
    #  absolute rounding error of those terms - ~2e-3 here, and the eager model's atomics add in a different order every run - so the
    #  absolute tolerance scales with the tensor's magnitude; a fixed 1e-3 failed about one run in thirty)
    for name, a, b in (("d/dx", x1.grad, x2.grad), ("d/dw", w1.grad, w2.grad)):
        err = (a - b).abs()
        at = int(err.argmax())
        atol = 1e-5 * float(b.abs().max())
        assert torch.allclose(a, b, rtol=1e-3, atol=atol), (name, float(err.max()), float(b.flatten()[at]), at, int((err > atol + 1e-3 * b.abs()).sum()))


class GATLayer(torch.nn.Module):
    """A GAT layer written with aten ops, scores and softmax included: what geot/match_replace/fused_mh_spmm.py:4-50 of the reference
    rewrites onto mh_spmm - where the layer then stops training, because the reference's mh_spmm has no backward
    (geot/mh_spmm.py:4-12)."""

    def __init__(self, fin, heads, fout):
        super().__init__()
        self.heads, self.fout = heads, fout
        self.lin = torch.nn.Linear(fin, heads * fout, bias=False)
        self.att_src = torch.nn.Parameter(torch.randn(heads, fout) * 0.3)
        self.att_dst = torch.nn.Parameter(torch.randn(heads, fout) * 0.3)

    def forward(self, x, edge_index):
        row, col = edge_index[0], edge_index[1]
        h = self.lin(x).view(-1, self.heads, self.fout)
        a_src, a_dst = (h * self.att_src).sum(-1), (h * self.att_dst).sum(-1)                 # [N, H]
        e = torch.exp(torch.nn.functional.leaky_relu(a_src.index_select(0, col) + a_dst.index_select(0, row), 0.2))
        denom = torch.zeros_like(a_dst).index_add(0, row, e)
        alpha = e / denom.index_select(0, row)                                                  # [nnz, H]
        msg = alpha.unsqueeze(-1) * h.index_select(0, col)
        return torch.zeros_like(h).index_add(0, row, msg)                                        # -> mh_spmm


@pytest.mark.gpu
def test_rewritten_gat_layer_takes_an_optimiser_step():
    """`mh_spmm` differentiates (d/dsrc over the transposed list, d/dweight by the multi-head SDDMM), so a GAT
    layer rewritten by `pattern_transform` trains: one SGD step of the rewritten program moves the parameters exactly as the eager
    model's step does."""
    from geot_amd.match_replace import pattern_transform
    torch._dynamo.reset()
    torch.manual_seed(3)
    n, nnz, fin, heads, fout = 1200, 40_000, 24, 4, 16
    x, ei, _ = _inputs(n=n, nnz=nnz, f=fin, device="cuda", seed=11)
    eager = GATLayer(fin, heads, fout).cuda()
    twin = GATLayer(fin, heads, fout).cuda()
    twin.load_state_dict(eager.state_dict())
    ep = pattern_transform(twin, (x, ei))
    assert torch.ops.geot.mh_spmm_rows.default in _targets(ep)
    fused = ep.module()
    target = torch.rand(n, heads, fout, device="cuda")
    steps = {}
    for name, mod in (("eager", eager), ("fused", fused)):
        opt = torch.optim.SGD(mod.parameters(), lr=0.05)
        before = [p.detach().clone() for p in mod.parameters()]
        loss = (mod(x, ei) - target).square().mean()
        opt.zero_grad()
        loss.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in mod.parameters()), name
        opt.step()
        steps[name] = (float(loss), sorted(((n_, (p.detach() - b)) for (n_, p), b in zip(mod.named_parameters(), before)), key=lambda t: t[0]))
    assert abs(steps["eager"][0] - steps["fused"][0]) <= 1e-5 * abs(steps["eager"][0])
    for (na, da), (nb, db) in zip(steps["eager"][1], steps["fused"][1]):
        assert da.shape == db.shape and float(da.abs().max()) > 0, na                      # the step moved the parameter ...
        assert torch.allclose(da, db, rtol=2e-3, atol=2e-5 * float(da.abs().max()) + 1e-9), (na, nb, float((da - db).abs().max()))   # ... the same way
