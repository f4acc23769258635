"""The reference's test/compile/ scripts (test_gcn / test_gin / test_graphsage / test_gat / test_appnp / test_sgc .py:
export a PyG layer, `pattern_transform` it, `torch.compile` the result, compare with the eager layer) on layers written
in plain torch - torch_geometric is not in this image.  Each module computes what the PyG layer of that name computes
for a Tensor `edge_index` (messages `x.index_select(0, col)`, optionally scaled, aggregated by `index_add` into zeros -
the decomposition the reference's passes key on, geot/match_replace/fused_gs.py:13-25, fused_gws.py:13-35,
fused_mh_spmm.py:9-30).  CPU: the rewrite itself (node counts).  `-m gpu`: rewritten == eager, also under torch.compile.
"""
import numpy as np
import pytest
import torch

from conftest import powerlaw_index


def _deg_norm(row, col, n, dtype):
    deg = torch.zeros(n, dtype=dtype, device=row.device).index_add(0, row, torch.ones(row.numel(), dtype=dtype, device=row.device))
    dinv = deg.clamp(min=1).pow(-0.5)
    return dinv.index_select(0, row) * dinv.index_select(0, col)          # gcn_norm


class GCN(torch.nn.Module):                                               # GCNConv
    def __init__(self, fin, fout):
        super().__init__()
        self.lin = torch.nn.Linear(fin, fout)

    def forward(self, x, edge_index):
        row, col = edge_index[0], edge_index[1]
        h = self.lin(x)
        norm = _deg_norm(row, col, x.shape[0], x.dtype)
        return h.new_zeros(h.shape).index_add(0, row, norm.view(-1, 1) * h.index_select(0, col))


class GIN(torch.nn.Module):                                               # GINConv(nn=MLP)
    def __init__(self, fin, fout):
        super().__init__()
        self.eps = 0.1
        self.mlp = torch.nn.Sequential(torch.nn.Linear(fin, fout), torch.nn.ReLU(), torch.nn.Linear(fout, fout))

    def forward(self, x, edge_index):
        row, col = edge_index[0], edge_index[1]
        agg = x.new_zeros(x.shape).index_add(0, row, x.index_select(0, col))
        return self.mlp((1 + self.eps) * x + agg)


class SAGE(torch.nn.Module):                                              # SAGEConv(aggr='mean')
    def __init__(self, fin, fout):
        super().__init__()
        self.lin_l, self.lin_r = torch.nn.Linear(fin, fout), torch.nn.Linear(fin, fout, bias=False)

    def forward(self, x, edge_index):
        row, col = edge_index[0], edge_index[1]
        s = x.new_zeros(x.shape).index_add(0, row, x.index_select(0, col))
        cnt = x.new_zeros(x.shape[0]).index_add(0, row, torch.ones(row.numel(), dtype=x.dtype, device=x.device))
        return self.lin_l(s / cnt.clamp(min=1).unsqueeze(-1)) + self.lin_r(x)


class GAT(torch.nn.Module):                                               # GATConv(heads=H, concat=True)
    def __init__(self, fin, fout, heads):
        super().__init__()
        self.h, self.f = heads, fout
        self.lin = torch.nn.Linear(fin, heads * fout, bias=False)
        self.att_src = torch.nn.Parameter(torch.rand(1, heads, fout))
        self.att_dst = torch.nn.Parameter(torch.rand(1, heads, fout))

    def forward(self, x, edge_index):
        row, col = edge_index[0], edge_index[1]
        n = x.shape[0]
        x3 = self.lin(x).view(n, self.h, self.f)
        a = torch.nn.functional.leaky_relu((x3 * self.att_src).sum(-1).index_select(0, col) + (x3 * self.att_dst).sum(-1).index_select(0, row), 0.2)
        amax = torch.full((n, self.h), -1e30, dtype=x.dtype, device=x.device).scatter_reduce(0, row.unsqueeze(-1).expand(-1, self.h), a, "amax")
        e = (a - amax.index_select(0, row)).exp()
        alpha = e / x.new_zeros(n, self.h).index_add(0, row, e).index_select(0, row)       # softmax over incoming edges
        out = torch.zeros_like(x3).index_add(0, row, alpha.unsqueeze(-1) * x3.index_select(0, col))
        return out.view(n, self.h * self.f)


class APPNP(torch.nn.Module):                                             # APPNP(K, alpha) after a linear
    def __init__(self, fin, fout, K=3, alpha=0.1):
        super().__init__()
        self.lin, self.K, self.alpha = torch.nn.Linear(fin, fout), K, alpha

    def forward(self, x, edge_index):
        row, col = edge_index[0], edge_index[1]
        h = self.lin(x)
        norm = _deg_norm(row, col, x.shape[0], x.dtype)
        z = h
        for _ in range(self.K):
            z = (1 - self.alpha) * z.new_zeros(z.shape).index_add(0, row, norm.unsqueeze(-1) * z.index_select(0, col)) + self.alpha * h
        return z


class SGC(torch.nn.Module):                                               # SGConv(K)
    def __init__(self, fin, fout, K=2):
        super().__init__()
        self.lin, self.K = torch.nn.Linear(fin, fout), K

    def forward(self, x, edge_index):
        row, col = edge_index[0], edge_index[1]
        norm = _deg_norm(row, col, x.shape[0], x.dtype)
        for _ in range(self.K):
            x = x.new_zeros(x.shape).index_add(0, row, norm.view(-1, 1) * x.index_select(0, col))
        return self.lin(x)


# model, fused nodes expected, op the propagate must turn into
ZOO = {
    "gcn": (lambda: GCN(16, 32), 1, "gather_weight_scatter_rows"),
    "gin": (lambda: GIN(16, 16), 1, "gather_scatter_rows"),
    "graphsage": (lambda: SAGE(16, 32), 1, "gather_scatter_rows"),
    "gat": (lambda: GAT(16, 8, 4), 1, "mh_spmm_rows"),
    "appnp": (lambda: APPNP(16, 16, K=3), 3, "gather_weight_scatter_rows"),
    "sgc": (lambda: SGC(16, 32, K=2), 2, "gather_weight_scatter_rows"),
}


def _graph(n, nnz, f, seed, device):
    rng = np.random.default_rng(seed)
    row = powerlaw_index(nnz, n, seed)
    ei = torch.from_numpy(np.stack([row, rng.integers(0, n, nnz)]).astype(np.int64)).to(device)
    return torch.rand(n, f, device=device), ei


@pytest.mark.parametrize("name", sorted(ZOO))
def test_every_propagate_is_fused(name):
    from geot_amd.match_replace import pattern_transform
    make, fused, op = ZOO[name]
    torch.manual_seed(0)
    ep = pattern_transform(make(), _graph(60, 500, 16, 1, "cpu"))
    targets = [n.target for n in ep.graph_module.graph.nodes if n.op == "call_function"]
    assert ep.geot_fused_nodes == fused, (name, ep.geot_fused_nodes)
    assert targets.count(getattr(torch.ops.geot, op).default) == fused


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(ZOO))
def test_rewritten_and_compiled_models_match_eager(name):
    from geot_amd.match_replace import pattern_transform
    torch._dynamo.reset()
    make, fused, _ = ZOO[name]
    torch.manual_seed(0)
    model = make().cuda()
    args = _graph(5000, 120_000, 16, 2, "cuda")
    with torch.no_grad():
        ref = model(*args)
        ep = pattern_transform(model, args)
        assert ep.geot_fused_nodes == fused
        mod = ep.module()
        got = mod(*args)
        assert got.shape == ref.shape and torch.allclose(got, ref, rtol=1e-4, atol=1e-4), name
        comp = torch.compile(mod)(*args)
        assert torch.allclose(comp, ref, rtol=1e-4, atol=1e-4), name


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(ZOO))
def test_static_graph_rewrite_matches_eager_and_touches_no_cache(name):
    """`pattern_transform(model, args, static_graph=True)`: the fused nodes run over ONE geot_amd.Graph built at
    export time - `geot::graph_spmm(handle, w, x)` - so a call neither fingerprints the index tensors nor looks anything up: the host
    layer's counters do not move across steps.  Results equal the eager layer; also under torch.compile."""
    from geot_amd import ops
    from geot_amd.graph import release_graph
    from geot_amd.match_replace import pattern_transform
    torch._dynamo.reset()
    make, fused, _ = ZOO[name]
    torch.manual_seed(0)
    model = make().cuda()
    args = _graph(5000, 120_000, 16, 2, "cuda")
    with torch.no_grad():
        ref = model(*args).double()
        ep = pattern_transform(model, args, static_graph=True)
        assert ep.geot_fused_nodes == fused and ep.geot_static_nodes == fused and len(ep.geot_graphs) == 1       # one edge list -> one handle
        targets = [n.target for n in ep.graph_module.graph.nodes if n.op == "call_function"]
        assert targets.count(torch.ops.geot.graph_spmm.default) == fused
        assert not any(getattr(t, "__name__", "").endswith("_rows.default") or "_rows" in str(t) for t in targets if "geot" in str(t))
        mod = ep.module()
        ops.clear_caches()
        got = mod(*args)
        st0 = ops.stats()
        for _ in range(3):
            got = mod(*args)
        assert ops.stats() == st0, "a static-graph program must not touch the host layer's caches / guards"
        scale = float(ref.abs().max())
        assert got.shape == ref.shape and float((got.double() - ref).abs().max()) <= 1e-5 * scale, name
        comp = torch.compile(mod)(*args)
        assert float((comp.double() - ref).abs().max()) <= 1e-5 * scale, name
    for h in ep.geot_graphs:
        release_graph(h)
    with pytest.raises(RuntimeError, match="no geot_amd.Graph is registered"):
        mod(*args)


@pytest.mark.gpu
def test_static_graph_rewrite_trains():
    """Gradients flow through `geot::graph_spmm` (register_autograd -> geot::graph_spmm_backward: d/dx over the transposed list, d/dweight
    by the SDDMM): a rewritten GAT and GCN take optimiser steps and follow the eager model's parameters."""
    from geot_amd.match_replace import pattern_transform
    for make in (lambda: GAT(16, 8, 4), lambda: GCN(16, 32)):
        torch.manual_seed(0)
        eager = make().cuda()
        args = _graph(3000, 90_000, 16, 5, "cuda")
        twin = make().cuda()
        twin.load_state_dict(eager.state_dict())
        ep = pattern_transform(twin, args, static_graph=True)
        mod = ep.module()
        opt_e = torch.optim.SGD(eager.parameters(), lr=0.05)
        opt_m = torch.optim.SGD(mod.parameters(), lr=0.05)
        target = torch.rand_like(eager(*args))
        for _ in range(3):
            for m, opt in ((eager, opt_e), (mod, opt_m)):
                opt.zero_grad()
                loss = ((m(*args) - target) ** 2).mean()
                loss.backward()
                opt.step()
        pe = dict(eager.named_parameters())
        for k, v in mod.named_parameters():
            ref = pe[k]
            assert float((v - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-6, k


@pytest.mark.gpu
def test_static_graph_rewrite_refuses_an_unsorted_edge_list():
    from geot_amd.match_replace import pattern_transform
    torch.manual_seed(0)
    x, ei = _graph(500, 4000, 16, 3, "cuda")
    ei = ei[:, torch.randperm(ei.shape[1], device="cuda")]
    with pytest.raises(ValueError, match="must ascend"):
        pattern_transform(GIN(16, 16).cuda(), (x, ei), static_graph=True)
