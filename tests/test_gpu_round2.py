"""Round-2 parity cases (`-m gpu`): the holes VERDICT.md (round 1) named.

* BASELINE.json configs[3]'s SHAPE (mh_spmm, H=4 x F=64 = 1-KiB rows) at >= 1 M edges against the oracle;
* gather mode with a src table beyond 2^31 elements (byte offsets past 8 GB);
* full-size properties of configs[2] (gws, 124 M edges, F=128) and of one GPU's shard of configs[4]
  (gather_scatter, 202 M edges, 57 GB src): checksum of checksums (linearity), sampled segments against
  float64 torch reductions, determinism, exact zeros - the style of test_cfg2_full_size_properties;
* gather_weight_scatter against rocSPARSE's CSR SpMM result on >= 10 M edges (the north star's comparator);
* a wrong `sorted=True` promise / an unsorted dst_index never returns wrong rows; the unsorted path is
  bit-reproducible; the transposed-edge cache cannot alias a dead edge list.
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, powerlaw_index

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.fixture(scope="module")
def geot():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import geot_amd
    return geot_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def device_powerlaw(nnz, keys, seed):
    sys.path.insert(0, ROOT)
    from bench import powerlaw_index as gen
    return gen(nnz, keys, seed, torch.device("cuda"))


def close_to_oracle(got, hi, mag, what):
    got = got.detach().cpu().numpy()
    assert got.shape == hi.shape, (what, got.shape, hi.shape)
    bound = RTOL * mag + 1e-30
    err = np.abs(got.astype(np.float64) - hi.astype(np.float64))
    assert np.all(err <= bound), f"{what}: max err/bound = {np.max(err / bound):.3g}"
    assert np.all(got[mag == 0] == 0), what


def test_mh_spmm_cfg4_shape_against_oracle(geot, oracle):
    """H=4 x F=64 (1-KiB rows, 64 lanes per row, head = f0 / 64), 1.2 M power-law edges, both weight layouts."""
    rng = np.random.default_rng(404)
    nodes, nnz, H, F = 20_000, 1_200_000, 4, 64
    di = powerlaw_index(nnz, nodes, 41)
    si = rng.integers(0, nodes, nnz).astype(np.int64)
    w = rng.random((nnz, H), dtype=np.float32)
    x = rng.random((nodes, H, F), dtype=np.float32)
    hi = oracle.mh_spmm(si, di, w, x, acc64=True)
    a = geot.mh_spmm(dev(si), dev(di), dev(w), dev(x))
    b = geot.mh_spmm(dev(si), dev(di), dev(np.ascontiguousarray(w.T)), dev(x))
    close_to_oracle(a, hi, hi, "mh [nnz,H]")
    close_to_oracle(b, hi, hi, "mh [H,nnz]")
    assert torch.equal(a, geot.mh_spmm(dev(si), dev(di), dev(w), dev(x)))          # deterministic
    # heads must not mix: zero one head's weights -> exactly that head's slice is zero
    w0 = w.copy()
    w0[:, 2] = 0
    c = geot.mh_spmm(dev(si), dev(di), dev(w0), dev(x))
    assert c[:, 2].abs().sum().item() == 0 and torch.equal(c[:, 1], a[:, 1]) and torch.equal(c[:, 3], a[:, 3])


def test_gather_src_table_beyond_2_31_elements(geot):
    """5 M x 512 fp32 = 2.56e9 elements (10.2 GB): gathered-row byte offsets exceed 2^32 and element offsets 2^31."""
    nodes, F, nnz, rows = 5_000_000, 512, 3_000_000, 200_000
    torch.manual_seed(5)
    x = torch.rand(nodes, F, device="cuda")
    di = device_powerlaw(nnz, rows, 51)
    si = torch.randint(0, nodes, (nnz,), device="cuda")
    si[: nnz // 2] = torch.randint(nodes - 400_000, nodes, (nnz // 2,), device="cuda")   # rows above 2^31 elements
    w = torch.rand(nnz, device="cuda")
    out = geot.gather_weight_scatter(si, di, w, x)
    assert out.shape == (rows, F)
    counts = torch.bincount(di, minlength=rows)
    offs = torch.cumsum(counts, 0) - counts
    pick = [int(counts.argmax()), 0, rows - 1] + torch.randint(0, rows, (60,)).tolist()
    high = 0
    for k in pick:
        e = slice(int(offs[k]), int(offs[k] + counts[k]))
        seg = (x[si[e]].double() * w[e].double()[:, None]).sum(0)
        assert torch.allclose(out[k].double(), seg, rtol=1e-5, atol=1e-6), k
        high += int((si[e] * F > 2**31).sum())
    assert high > 0
    assert out[counts == 0].abs().sum().item() == 0
    # checksum of checksums: sum over dst rows == sum over nodes of (total weight into the node) * its row
    cw = torch.zeros(nodes, dtype=torch.float64, device="cuda").index_add_(0, si, w.double())
    want = torch.zeros(F, dtype=torch.float64, device="cuda")
    for lo in range(0, nodes, 1_000_000):
        want += cw[lo:lo + 1_000_000] @ x[lo:lo + 1_000_000].double()
    assert torch.allclose(out.double().sum(0), want, rtol=1e-7)
    gs = geot.gather_scatter(si, di, x)
    k = pick[0]
    assert torch.allclose(gs[k].double(), x[si[int(offs[k]): int(offs[k] + counts[k])]].double().sum(0), rtol=1e-5)


def _gather_properties(geot, si, di, w, x, rows, samples=150):
    out = geot.gather_scatter(si, di, x) if w is None else geot.gather_weight_scatter(si, di, w, x)
    assert out.shape == (rows, x.shape[1])
    again = geot.gather_scatter(si, di, x) if w is None else geot.gather_weight_scatter(si, di, w, x)
    assert torch.equal(out, again)                                             # deterministic, no atomics
    del again
    counts = torch.bincount(di, minlength=rows)
    assert out[counts == 0].abs().sum().item() == 0                            # rows without edges: exactly zero
    offs = torch.cumsum(counts, 0) - counts
    pick = [int(counts.argmax()), 0, rows - 1, rows // 2] + torch.randint(0, rows, (samples,)).tolist()
    for k in pick:
        e = slice(int(offs[k]), int(offs[k] + counts[k]))
        msg = x[si[e]].double()
        if w is not None:
            msg = msg * w[e].double()[:, None]
        assert torch.allclose(out[k].double(), msg.sum(0), rtol=1e-5, atol=1e-6), k
    # linearity: column sums of the result == (per-node total weight) @ x
    nodes = x.shape[0]
    ones = torch.ones(1, dtype=torch.float64, device="cuda").expand(si.numel()) if w is None else w.double()
    cw = torch.zeros(nodes, dtype=torch.float64, device="cuda").index_add_(0, si, ones)
    want = torch.zeros(x.shape[1], dtype=torch.float64, device="cuda")
    step = 4_000_000
    for lo in range(0, nodes, step):
        want += cw[lo:lo + step] @ x[lo:lo + step].double()
    got = torch.zeros_like(want)
    for lo in range(0, rows, step):
        got += out[lo:lo + step].double().sum(0)
    assert torch.allclose(got, want, rtol=1e-8), (got - want).abs().max()
    return out


def test_cfg3_full_size_properties(geot):
    """BASELINE.json configs[2] at full size: gws, 2.45 M nodes, 123.7 M edges, F=128."""
    nodes, nnz, F = 2_449_029, 123_718_280, 128
    di = device_powerlaw(nnz, nodes, 7)
    g = torch.Generator(device="cuda")
    g.manual_seed(8)
    si = torch.randint(0, nodes, (nnz,), device="cuda", generator=g)
    w = torch.rand(nnz, device="cuda", generator=g)
    x = torch.rand(nodes, F, device="cuda", generator=g)
    _gather_properties(geot, si, di, w, x, nodes)


def test_cfg5_one_gpu_shard_full_size_properties(geot):
    """One GPU's shard of BASELINE.json configs[4]: 202 M edges -> 13.9 M rows, all 111 M source rows (56.9 GB)."""
    nodes_all, F = 111_059_956, 128
    nnz, rows = 1_615_685_872 // 8, nodes_all // 8
    di = device_powerlaw(nnz, rows, 13)
    g = torch.Generator(device="cuda")
    g.manual_seed(14)
    si = torch.randint(0, nodes_all, (nnz,), device="cuda", generator=g)
    x = torch.rand(nodes_all, F, device="cuda", generator=g)
    _gather_properties(geot, si, di, None, x, rows, samples=60)
    del x
    torch.cuda.empty_cache()


def test_gws_matches_rocsparse_csr_spmm(geot):
    """>= 10 M edges: the result of rocSPARSE's CSR SpMM (its nnz-split algorithm) on the same matrix."""
    sys.path.insert(0, ROOT)
    from tools import rocsparse
    nodes, nnz, F = 400_000, 12_000_000, 128
    di = device_powerlaw(nnz, nodes, 71)
    g = torch.Generator(device="cuda")
    g.manual_seed(72)
    si = torch.randint(0, nodes, (nnz,), device="cuda", generator=g)
    w = torch.rand(nnz, device="cuda", generator=g)
    x = torch.rand(nodes, F, device="cuda", generator=g)
    out = geot.gather_weight_scatter(si, di, w, x)
    rowptr, col = rocsparse.csr_from_sorted_coo(di, si, nodes)
    y = torch.empty(nodes, F, device="cuda")
    op = rocsparse.CsrSpMM(nodes, nodes, rowptr, col, w, x, y)
    assert op.prepare("csr_nnz_split")
    op.run()
    torch.cuda.synchronize()
    assert out.shape == y.shape
    scale = out.abs().max()
    assert ((out - y).abs().max() / scale).item() < 2e-5
    assert torch.allclose(out, y, rtol=1e-4, atol=1e-4 * float(scale))


def test_wrong_sorted_promise_still_sums_correctly(geot, oracle):
    """The reference's sorted kernels flush with atomicAdd, so sorted=True on a not-quite-sorted index still adds
    up (csrc/cuda/index_scatter_kernel.cuh:180,197).  Here the index is probed once per content: descents are
    routed to the sorted-gather path (or the atomic flush) - never to the atomic-free kernels."""
    from geot_amd import ops
    rng = np.random.default_rng(88)
    nnz, K, F = 300_000, 9000, 32
    index = np.sort(rng.integers(0, K, nnz)).astype(np.int64)
    index[-1] = K - 1
    swap = rng.integers(0, nnz - 600, 50)
    index[swap], index[swap + 500] = index[swap + 500].copy(), index[swap].copy()  # a few local descents
    index[-1] = K - 1
    assert (index[:-1] > index[1:]).sum() > 0
    src = rng.random((nnz, F), dtype=np.float32)
    hi = oracle.index_scatter(np.sort(index, kind="stable"), src[np.argsort(index, kind="stable")], rows=K, acc64=True)
    t_index, t_src = dev(index), dev(src)
    a = geot.index_scatter(0, t_src, t_index, "sum", sorted=True)                     # wrong promise
    close_to_oracle(a, hi, hi, "sorted=True on an index with descents")
    for _ in range(3):                                                                # remembered facts; bit-reproducible
        assert torch.equal(geot.index_scatter(0, t_src, t_index, "sum", sorted=True), a)
    assert torch.equal(geot.index_scatter(0, t_src, t_index, "sum", sorted=False), a)
    t_index.data[-1] = K + 10                         # behind the version counter: the row rule still follows index[-1]
    grown = geot.index_scatter(0, t_src, t_index, "sum", sorted=True)
    assert grown.shape[0] == K + 11 and grown[K - 1:K + 10].abs().sum().item() >= 0
    t_index.data[-1] = K - 1
    assert torch.equal(geot.index_scatter(0, t_src, t_index, "sum", sorted=True), a)
    mx = geot.index_scatter(0, t_src, t_index, "max", sorted=True)
    want = torch.zeros(K, F, device="cuda").scatter_reduce(0, t_index[:, None].expand(-1, F), t_src, "amax", include_self=False)
    assert torch.equal(mx, want)
    # gather ops with an unsorted dst_index
    nodes = K
    si = rng.integers(0, nodes, nnz).astype(np.int64)
    w = rng.random(nnz, dtype=np.float32)
    x = rng.random((nodes, F), dtype=np.float32)
    order = np.argsort(index, kind="stable")
    hi = oracle.gather_weight_scatter(si[order], index[order], w[order], x, rows=K, acc64=True)
    close_to_oracle(geot.gather_weight_scatter(dev(si), t_index, dev(w), dev(x)), hi, hi, "gws unsorted dst")
    hi = oracle.gather_scatter(si[order], index[order], x, rows=K, acc64=True)
    close_to_oracle(geot.gather_scatter(dev(si), t_index, dev(x)), hi, hi, "gs unsorted dst")
    H = 4
    wh = rng.random((nnz, H), dtype=np.float32)
    x3 = x.reshape(nodes, H, F // H)
    hi = oracle.mh_spmm(si[order], index[order], wh[order], x3, rows=K, acc64=True)
    close_to_oracle(geot.mh_spmm(dev(si), t_index, dev(wh), dev(x3)), hi, hi, "mh unsorted dst")
    close_to_oracle(geot.mh_spmm(dev(si), t_index, dev(np.ascontiguousarray(wh.T)), dev(x3)), hi, hi, "mh^T unsorted dst")
    assert ops.stats()["facts"] <= 16


@pytest.mark.parametrize("F", [32, 64, 128])
def test_unsorted_path_is_bit_reproducible(geot, oracle, F):
    """sorted=False on a shuffled index: (stable sort once, gather-mode kernels) -> run-to-run bit-equal; the atomic
    flush (GEOT_UNSORTED=atomic) is only close."""
    from geot_amd import hip
    rng = np.random.default_rng(90 + F)
    nnz, K = 1_000_000, 100_000
    index = rng.integers(0, K, nnz).astype(np.int64)
    index[-1] = K - 1
    src = rng.standard_normal((nnz, F)).astype(np.float32)
    t_index, t_src = dev(index), dev(src)
    a = geot.index_scatter(0, t_src, t_index, "sum", sorted=False)
    for _ in range(3):
        assert torch.equal(geot.index_scatter(0, t_src, t_index, "sum", sorted=False), a)
    order = np.argsort(index, kind="stable")
    hi = oracle.index_scatter(index[order], src[order], rows=K, acc64=True)
    mag = oracle.index_scatter(index[order], np.abs(src[order]), rows=K, acc64=True)
    close_to_oracle(a, hi, mag, "unsorted, sorted-gather path")
    c = hip.index_scatter_out(t_index, t_src, torch.empty_like(a), sorted=False)      # float atomics
    close_to_oracle(c, hi, mag, "unsorted, atomic flush")


def test_transposed_edge_cache_cannot_alias_a_dead_edge_list(geot):
    """an edge list that is freed and re-created with the same size lands on the same address
    with _version 0.  The cache entry keeps its key tensors alive, so that cannot happen while it lives."""
    torch.manual_seed(11)
    n, nnz, F = 500, 20_000, 16
    g = torch.rand(n, F, device="cuda")
    x0 = torch.rand(n, F, device="cuda")
    w0 = torch.rand(nnz, device="cuda")
    seen = set()
    for it in range(6):
        di = torch.sort(torch.randint(0, n, (nnz,), device="cuda")).values
        di[-1] = n - 1
        si = torch.randint(0, n, (nnz,), device="cuda")                # fresh tensors every step (dynamic graph)
        seen.add((si.data_ptr(), di.data_ptr()))
        x1, w1 = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        x2, w2 = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        geot.gather_weight_scatter(si, di, w1, x1).backward(g)
        torch.zeros(n, F, device="cuda").index_add(0, di, x2[si] * w2[:, None]).backward(g)
        assert torch.allclose(x1.grad, x2.grad, rtol=1e-4, atol=1e-4), it
        assert torch.allclose(w1.grad, w2.grad, rtol=1e-4, atol=1e-4), it
        del si, di
    # same for the facts of an index: free + re-create at the same address with different content
    for it in range(4):
        idx = torch.sort(torch.randint(0, 50 + 10 * it, (5000,), device="cuda")).values
        if it % 2:
            idx = idx.flip(0).contiguous()
            idx[-1] = idx.max()
        src = torch.rand(5000, 8, device="cuda")
        out = geot.index_scatter(0, src, idx, "sum", sorted=True)
        ref = torch.zeros(int(idx[-1]) + 1, 8, device="cuda").index_add_(0, idx, src)
        assert out.shape == ref.shape and torch.allclose(out, ref, rtol=1e-5, atol=1e-5), it
        del idx


def test_cached_artefacts_are_safe_on_another_stream(geot):
    """What the host layer caches per index content (the stable sort of an index with descents, the transposed edge
    list, a widened int32 index, the per-edge row ids of a CSR) is enqueued on the stream of the call that made it.  A
    second call on ANOTHER stream, issued while that work may still be running, must wait for it (an event per cache
    entry) - results are checked against float64 references with no synchronisation in between."""
    torch.manual_seed(5)
    nnz, K, F = 6_000_000, 200_000, 32
    index = torch.randint(0, K, (nnz,), device="cuda")
    index[-1] = K - 1
    src = torch.rand(nnz, F, device="cuda")
    ref = torch.zeros(K, F, dtype=torch.float64, device="cuda").index_add_(0, index, src.double())
    torch.cuda.synchronize()
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    a.wait_stream(torch.cuda.current_stream())
    b.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(a):
        out_a = geot.index_scatter(0, src, index, "sum", sorted=False)       # probe, stable sort, reduce - on stream a
    with torch.cuda.stream(b):
        out_b = geot.index_scatter(0, src, index, "sum", sorted=False)       # cached sort, consumed on stream b
    torch.cuda.synchronize()
    for name, o in (("producer stream", out_a), ("consumer stream", out_b)):
        assert torch.allclose(o.double(), ref, rtol=1e-5, atol=1e-4), name
    assert torch.equal(out_a, out_b)

    # transposed edge list + widened int32 indices (the backward pass) made on stream a, used on stream b
    n, e = 100_000, 4_000_000
    di = torch.sort(torch.randint(0, n, (e,), device="cuda")).values
    di[-1] = n - 1
    si = torch.randint(0, n, (e,), device="cuda")
    x = torch.rand(n, F, device="cuda")
    g = torch.rand(n, F, device="cuda")
    w = torch.rand(e, device="cuda")
    gx_ref = torch.zeros(n, F, dtype=torch.float64, device="cuda").index_add_(0, si, g.double()[di] * w.double()[:, None])
    gw_ref = (g.double()[di] * x.double()[si]).sum(1)
    si32, di32 = si.int(), di.int()
    torch.cuda.synchronize()
    grads = []
    for st in (a, b):
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            x1, w1 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            geot.gather_weight_scatter(si, di, w1, x1).backward(g)
            dots = torch.ops.geot.sddmm_coo_impl(si32, di32, g, x)             # the reference's wrapper casts to int32
            grads.append((x1.grad, w1.grad, dots))
    torch.cuda.synchronize()
    for gx, gw, dots in grads:
        assert torch.allclose(gx.double(), gx_ref, rtol=1e-5, atol=1e-4)
        assert torch.allclose(gw.double(), gw_ref, rtol=1e-5, atol=1e-4)
        assert torch.allclose(dots.double(), gw_ref, rtol=1e-5, atol=1e-4)

    # the source-blocked plan (Phase A: scans, a sort, host loop) and the weight in plan order: built on a, used on b
    from geot_amd import ops
    F2 = 64
    x2 = torch.rand(n, F2, device="cuda")
    y_ref = torch.zeros(n, F2, dtype=torch.float64, device="cuda").index_add_(0, di, x2.double()[si] * w.double()[:, None])
    old = ops.set_option("slab_mode", "always")
    try:
        torch.cuda.synchronize()
        before = ops.stats()["slab_calls"]
        ys = []
        for st in (a, b, a, b):
            with torch.cuda.stream(st):
                ys.append(geot.gather_weight_scatter(si, di, w, x2))
        torch.cuda.synchronize()
        assert ops.stats()["slab_calls"] - before == 4
        for y in ys:
            assert torch.allclose(y.double(), y_ref, rtol=1e-5, atol=1e-4)
            assert torch.equal(y, ys[0])
    finally:
        ops.set_option("slab_mode", old)


def test_dispatched_ops_are_hipgraph_capturable_after_one_eager_call(geot):
    """torch.cuda.graph around a whole forward pass (what removes the launch-bound host cost of small graphs): after ONE
    eager call - which probes the index facts - the dispatched operators launch with the remembered row count, do not
    read index[-1] back, cache nothing they produce during the capture, and replay with new feature values."""
    from geot_amd import ops
    torch.manual_seed(21)
    n, nnz, F, H = 3000, 40_000, 32, 4
    di = torch.sort(torch.randint(0, n, (nnz,), device="cuda")).values
    di[-1] = n - 1
    si = torch.randint(0, n, (nnz,), device="cuda")
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    rowptr[1:] = torch.bincount(di, minlength=n).cumsum(0)
    unsorted = torch.randint(0, n, (nnz,), device="cuda")
    unsorted[-1] = n - 1
    w = torch.rand(nnz, device="cuda")
    wh = torch.rand(nnz, H, device="cuda")
    x = torch.rand(n, F, device="cuda")
    xh = torch.rand(n, H, F // H, device="cuda")
    e = torch.rand(nnz, F, device="cuda")

    def forward():
        h = geot.gather_weight_scatter(si, di, w, x)                  # GCN-style aggregation
        h = geot.gather_scatter(si, di, torch.relu(h))
        pooled = geot.index_scatter(0, e, di, "sum", sorted=True)     # edge messages -> nodes
        anyorder = geot.index_scatter(0, e, unsorted, "max", sorted=False)   # index with descents: sort path
        c = geot.csr_gws(rowptr, si, w, x)
        m = geot.mh_spmm(si, di, wh, xh)
        return h, pooled, anyorder, c, m

    def reference():
        h = torch.zeros(n, F, device="cuda", dtype=torch.float64).index_add_(0, di, x.double()[si] * w.double()[:, None])
        h = torch.zeros(n, F, device="cuda", dtype=torch.float64).index_add_(0, di, torch.relu(h)[si])
        pooled = torch.zeros(n, F, device="cuda", dtype=torch.float64).index_add_(0, di, e.double())
        anyorder = torch.zeros(n, F, device="cuda").scatter_reduce_(0, unsorted[:, None].expand(-1, F), e, "amax", include_self=False)
        m = torch.zeros(n, H, F // H, device="cuda", dtype=torch.float64).index_add_(0, di, xh.double()[si] * wh.double()[:, :, None])
        return h, pooled, anyorder, m

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        forward()                                                     # the eager call: facts, workspace
        s.synchronize()
        before = ops.stats()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            outs = forward()
        after = ops.stats()
    assert after["probes"] == before["probes"] and after["facts"] == before["facts"]       # nothing probed, nothing cached
    for rep in range(3):
        x.uniform_()
        e.uniform_()
        w.uniform_()
        g.replay()
        torch.cuda.synchronize()
        h, pooled, anyorder, c, m = outs
        rh, rp, ra, rm = reference()
        assert torch.allclose(h.double(), rh, rtol=1e-5, atol=1e-4), rep
        assert torch.allclose(pooled.double(), rp, rtol=1e-5, atol=1e-4), rep
        assert torch.equal(anyorder, ra), rep
        assert c.shape == (n + 1, F) and torch.allclose(c[:n].double(), torch.zeros(n, F, device="cuda", dtype=torch.float64).index_add_(
            0, di, x.double()[si] * w.double()[:, None]), rtol=1e-5, atol=1e-4), rep
        assert torch.allclose(m.double(), rm, rtol=1e-5, atol=1e-4), rep
    # eager calls after the capture still see consistent caches
    with torch.no_grad():
        h2 = forward()[0]
    assert torch.allclose(h2, outs[0], rtol=1e-6, atol=1e-6)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torchrun: the parent (no GPU call) spawns the ranks as children and relays
    rank 0's line.  The ranks share this box's one GPU over a gloo rendezvous (GEOT_DIST_BACKEND=gloo)."""
    import json
    import subprocess
    env = dict(os.environ, GEOT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    for gpus, extra in ((2, []), (2, ["--workload", "cfg5", "--scale", "0.01"]), (2, ["--strong", "--cuts", "aligned"]),
                        (8, ["--strong"])):                              # the world size of BASELINE.json configs[4]
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "4", "--warmup", "2"] + extra,
                           env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-3000:] + p.stdout[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, lines
        r = json.loads(lines[0])
        assert r["n_gpus"] == gpus and r["value"] > 1e8 and r["unit"] == "edges/s"
        if gpus == 8:
            assert r["scaling"] == "strong" and r["config"]["nnz_per_gpu"] == 1_250_000 and r["boundary_exchange_ms"] > 0
        elif "--strong" in extra:                                           # aligned cuts: no data-path collective at all
            assert r["scaling"] == "strong" and r["boundary_exchange_ms"] is None and r["config"]["nnz_per_gpu"] == 5_000_000
        elif "cfg5" in extra:
            # one global list cut at nnz / 2: the cut may fall on a row start, and then there is nothing to exchange (world 8, seven
            # cuts, always has a shared key: tests/test_gpu_world8.py)
            assert r["scaling"] == "weak" and r["collective"] == "all_gather"
            both = r["boundary_exchange_ms_by_collective"]
            assert (both["all_gather"] is None) == (both["reduce_scatter"] is None) == (r["boundary_exchange_ms"] is None)
        else:
            assert r["scaling"] == "weak" and r["boundary_exchange_ms"] is not None and r["boundary_exchange_ms"] > 0
            both = r["boundary_exchange_ms_by_collective"]                 # the first SCALE run needs no code change: both forms timed
            assert both["all_gather"] > 0 and both["reduce_scatter"] > 0 and r["collective"] == "all_gather"
        assert r["ranks_seen"] == gpus and r["key_exchange_ms"] > 0
        assert 0 < r["roofline"]["frac"] < 1


def test_bench_secondary_object_small_scale(tmp_path):
    """The `secondary` object (gws vs rocSPARSE, mh_spmm) at 2 % scale: keys and sanity of the numbers."""
    import json
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--scale", "0.02",
                        "--no-cpu-baseline", "--secondary", "all", "--full", "--detail-out", str(tmp_path / "detail.json")],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])     # --full: the whole record on stdout (tools/profile_round.sh)
    assert r == json.load(open(tmp_path / "detail.json"))
    sec = r["secondary"]
    assert "error" not in sec, sec
    g = sec["gws_cfg3"]
    assert g["kernel_ms"] > 0 and 0 < g["roofline"]["frac"] < 1 and g["rocsparse_best_ms"] > 0, g
    assert g["max_rel_diff_vs_rocsparse"] < 2e-5 and g["rocsparse_best_algorithm"] in ("csr_nnz_split", "csr_merge_path", "csr_row_split", "default")
    m = sec["mh_spmm_cfg4"]
    assert m["kernel_ms"] > 0 and 0 < m["roofline"]["frac"] < 1
    loc = sec["gws_cfg3_local"]                                       # the stand-in with source locality, rocSPARSE beside it
    assert loc["kernel_ms"] > 0 and loc["rocsparse_best_ms"] > 0 and loc["max_rel_diff_vs_rocsparse"] < 2e-5 and "+-2000" in loc["workload"]
    for name in ("gws_cfg3_bf16", "mh_spmm_cfg4_bf16"):               # additional lines: 16-bit storage, never the headline
        assert sec[name]["kernel_ms"] > 0 and "bfloat16" in sec[name]["workload"], name
    t = sec["gws_train_step_cfg4_graph"]                              # forward + backward through autograd, both ways
    assert "error" not in t and t["as_dispatched"]["forward_backward_ms"] > t["as_dispatched"]["forward_ms"] > 0 and t["per_edge_kernels"]["forward_backward_ms"] > 0, t
    assert r["dtype"] == "f32" and "traffic_source" in r["roofline"]
