"""geot_amd.reorder: one-time node renumbering for gathers on graphs that ship with unordered node ids (no counterpart in the
reference).  CPU: the ordering logic on a shuffled block model.  GPU (`-m gpu`): the renumbered operators
equal the direct ones (forward and gradients), and a structureless graph is declined."""
import numpy as np
import pytest
import torch


def block_model(nodes, nnz, intra, seed, device="cpu"):
    """Communities of 500-3000 nodes, `intra` of a node's edges inside its community, ids shuffled; dst-sorted int64 COO."""
    g = torch.Generator().manual_seed(seed)
    sizes, left = [], nodes
    while left > 0:
        s = min(int(torch.randint(500, 3001, (1,), generator=g).item()), left)
        sizes.append(s)
        left -= s
    sizes_t = torch.tensor(sizes)
    starts = torch.cumsum(sizes_t, 0) - sizes_t
    comm = torch.repeat_interleave(torch.arange(len(sizes)), sizes_t)
    dst = torch.randint(0, nodes, (nnz,), generator=g)
    c = comm[dst]
    src_in = starts[c] + (torch.rand(nnz, generator=g) * sizes_t[c]).long()
    src = torch.where(torch.rand(nnz, generator=g) < intra, src_in, torch.randint(0, nodes, (nnz,), generator=g))
    shuffle = torch.randperm(nodes, generator=g)
    dst, src = shuffle[dst], shuffle[src]
    o = torch.argsort(dst, stable=True)
    truth = torch.empty(nodes, dtype=torch.int64)
    truth[shuffle] = torch.arange(nodes)
    di, si = dst[o].contiguous(), src[o].contiguous()
    di[-1] = nodes - 1
    return si.to(device), di.to(device), truth.to(device), len(sizes)


def test_label_propagation_finds_the_communities_of_a_shuffled_block_model():
    from geot_amd import reorder
    nodes, nnz = 40_000, 1_600_000
    si, di, truth, ncomm = block_model(nodes, nnz, 0.9, 1)
    before = reorder.edge_locality(si, di, window=3000)
    ceiling = reorder.edge_locality(truth[si], truth[di], window=3000)
    labels = reorder.label_propagation(si, di, nodes, sweeps=10)
    rank = reorder.rank_from_labels(labels)
    assert torch.equal(torch.sort(rank).values, torch.arange(nodes))          # a permutation
    after = reorder.edge_locality(rank[si], rank[di], window=3000)
    assert before < 0.2 and ceiling > 0.85 and after >= 0.95 * ceiling, (before, after, ceiling)
    assert int(torch.unique(labels).numel()) <= 2 * ncomm


def test_rank_orders_by_label_then_id_and_directed_mode_runs():
    from geot_amd import reorder
    labels = torch.tensor([5, 2, 5, 2, 9])
    rank = reorder.rank_from_labels(labels)
    assert rank.tolist() == [2, 0, 3, 1, 4]
    si, di, _, _ = block_model(5_000, 100_000, 0.9, 2)
    lab = reorder.label_propagation(si, di, 5_000, sweeps=3, symmetric=False)
    assert lab.shape == (5_000,) and int(lab.min()) >= 0 and int(lab.max()) < 5_000
    assert reorder.edge_locality(si[:0], di[:0]) == 1.0


def test_renumber_accepts_structure_and_declines_its_absence():
    """The one-time part runs on any device (torch ops): a shuffled block model is renumbered (most edges end up within an
    L2-sized window of their destination, the new list is dst-sorted, the permutations are inverse to each other); a graph with
    uniform-random sources has nothing to find and is declined - the two permutations per call would be pure overhead."""
    from geot_amd import reorder
    nodes, nnz = 40_000, 1_200_000
    si, di, truth, _ = block_model(nodes, nnz, 0.8, 7)
    g = reorder.renumber(si, di, nodes)
    assert g is not None and g.rows == nodes and g.locality_after > 0.9 and g.locality_after - g.locality_before >= 0.25
    assert bool((g.dst_index[1:] >= g.dst_index[:-1]).all())
    assert torch.equal(g.order[g.rank], torch.arange(nodes)) and torch.equal(g.rank[g.order], torch.arange(nodes))
    assert torch.equal(g.dst_index, g.rank[di][g.edge_perm]) and torch.equal(g.src_index, g.rank[si][g.edge_perm])
    w = torch.rand(nnz)
    assert torch.equal(g.edge_values(w), w[g.edge_perm])
    gen = torch.Generator().manual_seed(1)
    di_r = torch.randint(0, nodes, (nnz,), generator=gen).sort().values
    di_r[-1] = nodes - 1
    si_r = torch.randint(0, nodes, (nnz,), generator=gen)
    assert reorder.renumber(si_r, di_r, nodes) is None
    with pytest.raises(ValueError, match="int64 permutation"):
        reorder.RenumberedGraph(si, di, nodes, torch.arange(nodes - 1))


@pytest.mark.gpu
def test_renumbered_operators_equal_the_direct_ones_forward_and_backward():
    import geot_amd as geot
    from geot_amd import reorder
    nodes, nnz, F, H = 60_000, 2_400_000, 64, 4
    si, di, truth, _ = block_model(nodes, nnz, 0.9, 3, device="cuda")
    g = reorder.renumber(si, di, nodes)
    assert g is not None and g.locality_after > 0.9 and g.locality_after > g.locality_before + 0.4 and g.rows == nodes   # (60 k nodes: the 16 k-row window already covers half of a shuffled graph)
    gen = torch.Generator(device="cuda").manual_seed(4)
    x = torch.rand(nodes, F, device="cuda", generator=gen)
    w = torch.rand(nnz, device="cuda", generator=gen)
    # forward: gather_scatter / gather_weight_scatter / mh_spmm
    scale = float(geot.gather_scatter(si, di, x).abs().max())
    assert float((g.gather_scatter(x) - geot.gather_scatter(si, di, x)).abs().max()) <= 1e-5 * scale
    want = geot.gather_weight_scatter(si, di, w, x)
    assert float((g.gather_weight_scatter(w, x) - want).abs().max()) <= 1e-5 * float(want.abs().max())
    w_new = g.edge_values(w)                                                  # a static weight, permuted once
    assert float((g.gather_weight_scatter(w_new, x, in_new_order=True) - want).abs().max()) <= 1e-5 * float(want.abs().max())
    xh = torch.rand(nodes, H, F // H, device="cuda", generator=gen)
    wh = torch.rand(nnz, H, device="cuda", generator=gen)
    want_h = geot.mh_spmm(si, di, wh, xh)
    assert float((g.mh_spmm(wh, xh) - want_h).abs().max()) <= 1e-5 * float(want_h.abs().max())
    # two layers that STAY in the new order: one permutation in, one out
    h_new = g.gather_weight_scatter_new_order(w_new, g.rows_in(x))
    y2 = g.rows_out(g.gather_weight_scatter_new_order(w_new, h_new))
    want2 = geot.gather_weight_scatter(si, di, w, want)
    assert float((y2 - want2).abs().max()) <= 1e-5 * float(want2.abs().max())
    # gradients: d/dx and d/dweight through the permutations and geot's own autograd formulas
    xa, wa = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    xb, wb = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    cot = torch.rand(nodes, F, device="cuda", generator=gen)
    (g.gather_weight_scatter(wa, xa) * cot).sum().backward()
    (geot.gather_weight_scatter(si, di, wb, xb) * cot).sum().backward()
    assert float((xa.grad - xb.grad).abs().max()) <= 1e-5 * float(xb.grad.abs().max())
    assert float((wa.grad - wb.grad).abs().max()) <= 1e-5 * float(wb.grad.abs().max())


@pytest.mark.gpu
def test_row_rule_and_structureless_graphs():
    import geot_amd as geot
    from geot_amd import reorder
    nodes, nnz, F = 30_000, 600_000, 32
    gen = torch.Generator(device="cuda").manual_seed(5)
    di = torch.randint(0, nodes - 100, (nnz,), device="cuda", generator=gen).sort().values    # the last 100 nodes receive nothing
    si = torch.randint(0, nodes, (nnz,), device="cuda", generator=gen)
    assert reorder.renumber(si, di, nodes) is None                            # uniform-random sources: nothing to find
    g = reorder.RenumberedGraph(si, di, nodes, torch.randperm(nodes, device="cuda", generator=gen))   # any permutation is correct
    x = torch.rand(nodes, F, device="cuda", generator=gen)
    want = geot.gather_scatter(si, di, x)
    got = g.gather_scatter(x)
    assert got.shape == want.shape == (int(di[-1]) + 1, F)                    # rows = dst_index[-1] + 1, as the reference
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    with pytest.raises(ValueError, match="one per node"):
        g.gather_scatter(x[:-1])
