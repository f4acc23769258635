"""The source-blocked kernels (csrc/seg_slab.hip) against the oracle and against the per-edge gather kernels
(`-m gpu`): all weight modes, the three row widths, several rounds, split hubs, empty rows, out-of-range sources,
determinism, and the host layer's routing (second call with the same edge list, GEOT_SLAB)."""
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


def close(got, hi, what):
    got = got.detach().cpu().numpy().reshape(hi.shape)
    bound = 1e-5 * np.abs(hi) + 1e-30
    err = np.abs(got.astype(np.float64) - hi.astype(np.float64))
    assert np.all(err <= bound), f"{what}: max err/bound = {np.max(err / bound):.3g}"
    assert np.all(got[hi == 0] == 0), what


@pytest.mark.parametrize("nodes,nnz,H,Fh", [(40_000, 3_000_000, 4, 64), (70_000, 2_500_000, 1, 128), (150_000, 3_000_000, 1, 64),
                                           (3000, 400_000, 2, 32), (500, 1_000_000, 8, 32)])
def test_slab_kernel_against_oracle(geot, oracle, nodes, nnz, H, Fh):
    from geot_amd import hip, slab
    rng = np.random.default_rng(nodes + H)
    di = powerlaw_index(nnz, nodes, nodes)
    di[di == 7] = 8                                                     # an empty key
    si = rng.integers(0, nodes, nnz).astype(np.int64)
    F = H * Fh
    x = rng.random((nodes, H, Fh), dtype=np.float32)
    out = torch.empty(nodes, H, Fh, device="cuda")
    d_si, d_di, d_x = dev(si), dev(di), dev(x)
    if H == 1:
        w = rng.random(nnz, dtype=np.float32)
        plan = slab.build_plan(d_si, d_di, nodes, nodes, F * 4, 1, 1)
        slab.slab_spmm_out(plan, dev(w), 1, d_x, out, 1, Fh)
        hi = oracle.gather_weight_scatter(si, di, w, x.reshape(nodes, F), rows=nodes, acc64=True)
        close(out, hi, "gws")
        again = torch.empty_like(out)
        slab.slab_spmm_out(plan, dev(w), 1, d_x, again, 1, Fh)
        assert torch.equal(out, again)                                  # deterministic
        plan0 = slab.build_plan(d_si, d_di, nodes, nodes, F * 4, 0, 1)
        slab.slab_spmm_out(plan0, None, 0, d_x, out, 1, Fh)
        close(out, oracle.gather_scatter(si, di, x.reshape(nodes, F), rows=nodes, acc64=True), "gs")
        ref = hip.gather_scatter_out(d_si, d_di, d_x.view(nodes, F), torch.empty(nodes, F, device="cuda"))
        assert torch.allclose(out.view(nodes, F), ref, rtol=1e-5, atol=1e-5)
    else:
        w = rng.random((nnz, H), dtype=np.float32)
        hi = oracle.mh_spmm(si, di, w, x, rows=nodes, acc64=True)
        plan = slab.build_plan(d_si, d_di, nodes, nodes, F * 4, 2, H)
        assert plan.meta["rounds"] >= 1
        slab.slab_spmm_out(plan, dev(w), 2, d_x, out, H, Fh)
        close(out, hi, "mh edge-major")
        plan3 = slab.build_plan(d_si, d_di, nodes, nodes, F * 4, 3, H)
        slab.slab_spmm_out(plan3, dev(np.ascontiguousarray(w.T)), 3, d_x, out, H, Fh)
        close(out, hi, "mh head-major")


def test_slab_hub_rows_many_rounds_and_bad_sources(geot, oracle):
    """One row with a third of all edges (split into carry pieces), > 2 rounds of groups, sources out of range
    (ignored like in every other kernel: memory-safe, row 0 is NOT added)."""
    from geot_amd import slab
    rng = np.random.default_rng(3)
    nodes, nnz, F = 120_000, 4_000_000, 256                              # 1-KiB rows: 2048 units x 15 rows per round
    di = powerlaw_index(nnz, nodes, 5)
    di[: nnz // 3] = di[nnz // 3]
    di = np.sort(di)
    si = rng.integers(0, nodes, nnz).astype(np.int64)
    w = rng.random(nnz, dtype=np.float32)
    x = rng.random((nodes, F), dtype=np.float32)
    plan = slab.build_plan(dev(si), dev(di), nodes, nodes, F * 4, 1, 1)
    assert plan.meta["rounds"] >= 3 and plan.meta["split_rows"] >= 1
    out = torch.empty(nodes, F, device="cuda")
    slab.slab_spmm_out(plan, dev(w), 1, dev(x), out, 1, F)
    close(out, oracle.gather_weight_scatter(si, di, w, x, rows=nodes, acc64=True), "hub")
    # fewer output rows than keys (keys >= out_rows are ignored), more source rows referenced than exist
    small = nodes - 1000
    plan2 = slab.build_plan(dev(si), dev(di), nodes, nodes, F * 4, 1, 1)
    out2 = torch.empty(small, F, device="cuda")
    slab.slab_spmm_out(plan2, dev(w), 1, dev(x), out2, 1, F)
    assert torch.equal(out2, out[:small])


def test_host_layer_routes_dense_graphs_on_the_second_call(geot, oracle, monkeypatch):
    from geot_amd import ops
    rng = np.random.default_rng(9)
    nodes, nnz, H, Fh = 30_000, 9_000_000, 4, 64                          # dense enough for the rule (>= 8 M edges)
    di = powerlaw_index(nnz, nodes, 2)
    si = rng.integers(0, nodes, nnz).astype(np.int64)
    w = rng.random((nnz, H), dtype=np.float32)
    x = rng.random((nodes, H, Fh), dtype=np.float32)
    from geot_amd import slab
    assert slab.worthwhile(nnz, nodes, nodes, H * Fh * 4)
    d_si, d_di, d_w, d_x = dev(si), dev(di), dev(w), dev(x)
    built0, calls0, trials0, rejected0 = (ops.stats()[k] for k in ("plans_built", "slab_calls", "plan_trials", "plans_rejected"))
    a = geot.mh_spmm(d_si, d_di, d_w, d_x)                                # first sighting: per-edge gather kernel
    assert ops.stats()["plans_built"] == built0
    b = geot.mh_spmm(d_si, d_di, d_w, d_x)                                # second: plan built and TRIED against the per-edge kernels
    c = geot.mh_spmm(d_si, d_di, d_w, d_x)                                # third: whichever was faster on this box
    st = ops.stats()
    assert st["plans_built"] == built0 + 1 and st["plan_trials"] == trials0 + 1
    kept = st["plans_rejected"] == rejected0
    assert st["slab_calls"] == calls0 + (2 if kept else 1)
    assert torch.allclose(b, c, rtol=1e-5, atol=1e-5) and torch.allclose(a, b, rtol=1e-5, atol=1e-5)
    assert torch.equal(c, geot.mh_spmm(d_si, d_di, d_w, d_x))             # the decision stands: same kernels, same bits
    hi = oracle.mh_spmm(si, di, w, x, rows=nodes, acc64=True)
    close(b, hi, "routed mh_spmm")
    d_si[0] = (d_si[0] + 1) % nodes                                       # in-place edit: the plan must not be reused
    si[0] = (si[0] + 1) % nodes
    d = geot.mh_spmm(d_si, d_di, d_w, d_x)
    close(d, oracle.mh_spmm(si, di, w, x, rows=nodes, acc64=True), "after an in-place edit")
    mode0 = ops.set_option("slab_mode", "never")
    e = geot.mh_spmm(d_si, d_di, d_w, d_x)
    assert torch.allclose(d, e, rtol=1e-5, atol=1e-5)
    # small graphs never take the path in auto mode; GEOT_SLAB=1 forces it (gws + gs + backward through autograd)
    ops.set_option("slab_mode", "always")
    monkeypatch.setattr(ops, "_restore_slab_mode", mode0, raising=False)
    n2, z2, F = 2000, 60_000, 64
    di2 = dev(powerlaw_index(z2, n2, 3))
    si2 = dev(rng.integers(0, n2, z2).astype(np.int64))
    w2 = torch.rand(z2, device="cuda", requires_grad=True)
    x2 = torch.rand(n2, F, device="cuda", requires_grad=True)
    calls = ops.stats()["slab_calls"]
    try:
        y = geot.gather_weight_scatter(si2, di2, w2, x2)
        y.sum().backward()
        gs = geot.gather_scatter(si2, di2, x2.detach())
    finally:
        ops.set_option("slab_mode", mode0)
    assert ops.stats()["slab_calls"] >= calls + 3                         # forward, d/dsrc and gather_scatter on the slab kernel
    ref = torch.zeros(n2, F, device="cuda").index_add(0, di2, x2.detach()[si2] * w2.detach()[:, None])
    assert torch.allclose(y, ref, rtol=1e-4, atol=1e-4)
    xg = torch.zeros(n2, F, device="cuda").index_add(0, si2, w2.detach()[:, None].expand(-1, F).contiguous())
    assert torch.allclose(x2.grad, xg, rtol=1e-4, atol=1e-4)
    assert torch.allclose(gs, torch.zeros(n2, F, device="cuda").index_add(0, di2, x2.detach()[si2]), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("reduce", ["mean", "max", "min", "sum", "add"])
def test_slab_reductions_match_the_per_edge_kernels(geot, reduce):
    """GraphSAGE-style mean / max aggregation on a dense graph: the source-blocked kernel (forced here) against
    torch.scatter_reduce over the materialised messages and against the per-edge kernels; a hub that is split into
    carry pieces and an empty row included."""
    from geot_amd import ops
    rng = np.random.default_rng(21)
    nodes, nnz, F = 30_000, 2_000_000, 128
    di_h = powerlaw_index(nnz, nodes, 6)
    di_h[: nnz // 4] = di_h[nnz // 4]
    di_h = np.sort(di_h)
    di_h[di_h == 9] = 10
    si, di = dev(rng.integers(0, nodes, nnz).astype(np.int64)), dev(di_h)
    x = dev(rng.standard_normal((nodes, F)).astype(np.float32))
    w = dev(rng.random(nnz, dtype=np.float32) + 0.5)
    kind = {"add": "sum", "max": "amax", "min": "amin"}.get(reduce, reduce)
    for weight in (None, w):
        msg = x[si] if weight is None else x[si] * weight[:, None]
        if reduce in ("max", "min"):
            ref = torch.zeros(nodes, F, device="cuda").scatter_reduce(0, di[:, None].expand(-1, F), msg, kind, include_self=False)
        else:                                  # float64 reference + the magnitude the fp32 error bound scales with
            # (di is ascending: a segmented reduction - index_add_ would hammer the hub's one row with atomics for seconds)
            lengths = torch.bincount(di, minlength=nodes)
            ref = torch.segment_reduce(msg.double(), "sum", lengths=lengths, axis=0, unsafe=True)
            mag = torch.segment_reduce(msg.double().abs(), "sum", lengths=lengths, axis=0, unsafe=True)
            if reduce == "mean":
                cnt = torch.bincount(di, minlength=nodes).clamp(min=1).double()[:, None]
                ref, mag = ref / cnt, mag / cnt
        del msg
        run = (lambda: geot.gather_scatter(si, di, x, reduce)) if weight is None else (lambda: geot.gather_weight_scatter(si, di, weight, x, reduce))
        old = ops.set_option("slab_mode", "always")
        calls = ops.stats()["slab_calls"]
        try:
            out = run()
            again = run()
        finally:
            ops.set_option("slab_mode", "never")
        assert ops.stats()["slab_calls"] == calls + 2
        try:
            tile = run()                                                  # the per-edge kernels on the same inputs
        finally:
            ops.set_option("slab_mode", old)
        assert torch.equal(out, again)
        if reduce in ("max", "min"):
            assert torch.equal(out, ref) and torch.equal(out, tile)
        else:
            bound = 1e-5 * mag + 1e-30
            assert bool(((out.double() - ref).abs() <= bound).all()) and bool(((tile.double() - ref).abs() <= bound).all())
        assert out[9].abs().sum().item() == 0                             # the empty row


@pytest.mark.parametrize("nodes,nnz,F", [(30_000, 2_500_000, 128), (60_000, 2_000_000, 256), (100_000, 2_000_000, 64), (300, 200_000, 128)])
def test_slab_sddmm_against_oracle_and_the_per_edge_kernel(geot, oracle, nodes, nnz, F):
    """d/dweight of gather_weight_scatter over the plan: every edge written once, in ORIGINAL edge order; hubs (split
    rows share their m1 row), out-of-range sources (-> 0), more m2 rows than m1 rows."""
    from geot_amd import hip, slab
    rng = np.random.default_rng(nodes)
    di = powerlaw_index(nnz, nodes, nodes + 1)
    di[: nnz // 5] = di[nnz // 5]
    di = np.sort(di)
    src_rows = nodes + 37
    si = rng.integers(0, src_rows, nnz).astype(np.int64)
    m1 = rng.standard_normal((nodes, F)).astype(np.float32)
    m2 = rng.standard_normal((src_rows, F)).astype(np.float32)
    plan = slab.build_plan(dev(si), dev(di), nodes, src_rows, F * 4, 1, 1)
    out = torch.full((nnz,), float("nan"), device="cuda")
    slab.slab_sddmm_out(plan, dev(m1), dev(m2), out)
    assert not torch.isnan(out).any()                                     # every edge written
    ref = oracle.sddmm_coo(si, di, m1, m2, acc64=True)
    mag = oracle.sddmm_coo(si, di, np.abs(m1), np.abs(m2), acc64=True)
    got = out.cpu().numpy()
    assert np.all(np.abs(got - ref) <= 1e-5 * mag + 1e-30)
    tile = hip.sddmm_coo_out(dev(si), dev(di), dev(m1), dev(m2), torch.empty(nnz, device="cuda"))
    assert torch.allclose(out, tile, rtol=1e-4, atol=1e-4)
    again = torch.empty(nnz, device="cuda")
    slab.slab_sddmm_out(plan, dev(m1), dev(m2), again)
    assert torch.equal(out, again)


def test_gws_training_step_on_the_source_blocked_path(geot):
    """forward, d/dsrc (the same op on the transposed edge list) and d/dweight (SDDMM over the forward's plan) all take
    the source-blocked kernels when the path is on; gradients against dense autograd."""
    from geot_amd import ops
    rng = np.random.default_rng(5)
    n, nnz, F = 5000, 400_000, 128
    di = dev(powerlaw_index(nnz, n, 9))
    si = dev(rng.integers(0, n, nnz).astype(np.int64))
    g = torch.rand(n, F, device="cuda")
    x0, w0 = torch.rand(n, F, device="cuda"), torch.rand(nnz, device="cuda")
    old = ops.set_option("slab_mode", "always")
    try:
        calls = ops.stats()["slab_calls"]
        x1, w1 = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        geot.gather_weight_scatter(si, di, w1, x1).backward(g)
        assert ops.stats()["slab_calls"] == calls + 3
    finally:
        ops.set_option("slab_mode", old)
    x2, w2 = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
    torch.zeros(n, F, device="cuda").index_add(0, di, x2[si] * w2[:, None]).backward(g)
    assert torch.allclose(x1.grad, x2.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(w1.grad, w2.grad, rtol=1e-4, atol=1e-3)


def test_static_weights_are_permuted_once_and_edits_are_seen(geot, oracle):
    """gather_weight_scatter on the source-blocked path keeps a plan-ordered copy of a weight tensor it sees twice
    (a normalised adjacency); an in-place edit of the weight (version counter) must not be served from the copy."""
    from geot_amd import ops
    rng = np.random.default_rng(31)
    n, nnz, F = 4000, 300_000, 64
    di_h, si_h = powerlaw_index(nnz, n, 4), rng.integers(0, n, nnz).astype(np.int64)
    w_h, x_h = rng.random(nnz, dtype=np.float32), rng.random((n, F), dtype=np.float32)
    si, di, w, x = dev(si_h), dev(di_h), dev(w_h), dev(x_h)
    old = ops.set_option("slab_mode", "always")
    try:
        outs = [geot.gather_weight_scatter(si, di, w, x) for _ in range(4)]     # 3rd call on: the plan-ordered copy
        hi = oracle.gather_weight_scatter(si_h, di_h, w_h, x_h, acc64=True)
        for o in outs:
            close(o, hi, "static weights")
            assert torch.equal(o, outs[0])
        w.mul_(2.0)                                                              # in place: version counter moves
        close(geot.gather_weight_scatter(si, di, w, x), 2 * hi, "after an in-place edit of the weights")
        w2 = torch.rand(nnz, device="cuda")                                      # a different tensor every call (attention)
        ref = torch.zeros(n, F, device="cuda").index_add_(0, di, x[si] * w2[:, None])
        assert torch.allclose(geot.gather_weight_scatter(si, di, w2, x), ref, rtol=1e-4, atol=1e-4)
    finally:
        ops.set_option("slab_mode", old)


def test_slab_calls_are_hipgraph_capturable(geot):
    """geot_slab_spmm / geot_slab_sddmm: two memsets and kernels on the caller's stream, no host sync, no allocation -
    capture once, replay with new operand values at the same addresses."""
    from geot_amd import slab
    rng = np.random.default_rng(13)
    n, nnz, F = 20_000, 1_500_000, 128
    di = dev(powerlaw_index(nnz, n, 8))
    si = dev(rng.integers(0, n, nnz).astype(np.int64))
    w, x, g_ = torch.rand(nnz, device="cuda"), torch.rand(n, F, device="cuda"), torch.rand(n, F, device="cuda")
    out, dw = torch.empty(n, F, device="cuda"), torch.empty(nnz, device="cuda")
    plan = slab.build_plan(si, di, n, n, F * 4, 1, 1)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        slab.slab_spmm_out(plan, w, 1, x, out, 1, F)                    # warm-up on the capture stream (workspace)
        slab.slab_sddmm_out(plan, g_, x, dw)
        s.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            slab.slab_spmm_out(plan, w, 1, x, out, 1, F)
            slab.slab_sddmm_out(plan, g_, x, dw)
    for rep in range(3):
        x.uniform_()
        w.uniform_()
        out.fill_(float("nan"))
        dw.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        ref = torch.zeros_like(out).index_add_(0, di, x[si] * w[:, None])
        assert torch.allclose(out, ref, rtol=1e-5, atol=1e-4)
        assert torch.allclose(dw, (g_[di] * x[si]).sum(1), rtol=1e-4, atol=1e-4)


def test_dispatched_slab_path_is_capturable(geot):
    """torch.cuda.graph around the dispatched operator on a graph that has a source-blocked plan: under capture the cached
    plan (and the weight kept in plan order) is used, nothing is built or cached, replays follow new feature values."""
    from geot_amd import ops
    rng = np.random.default_rng(17)
    n, nnz, F = 20_000, 1_200_000, 128
    di = dev(powerlaw_index(nnz, n, 9))
    si = dev(rng.integers(0, n, nnz).astype(np.int64))
    w, x = torch.rand(nnz, device="cuda"), torch.rand(n, F, device="cuda")
    old = ops.set_option("slab_mode", "always")
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(3):                                        # plan built, static weight permuted into plan order
                geot.gather_weight_scatter(si, di, w, x)
            s.synchronize()
            st0 = ops.stats()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                y = geot.gather_weight_scatter(si, di, w, x)
                ymax = torch.ops.geot.gather_reduce(si, di, None, x, "max")
            st1 = ops.stats()
        assert st1["slab_calls"] - st0["slab_calls"] == 2 and st1["plans_built"] == st0["plans_built"]
        for rep in range(3):
            x.uniform_()
            g.replay()
            torch.cuda.synchronize()
            ref = torch.zeros(n, F, device="cuda", dtype=torch.float64).index_add_(0, di, x.double()[si] * w.double()[:, None])
            assert torch.allclose(y.double(), ref, rtol=1e-5, atol=1e-4), rep
            refmax = torch.full((n, F), float("-inf"), device="cuda").scatter_reduce_(0, di[:, None].expand(-1, F), x[si], "amax")
            refmax[torch.bincount(di, minlength=n) == 0] = 0
            assert torch.equal(ymax, refmax), rep
    finally:
        ops.set_option("slab_mode", old)


def test_backward_computes_only_what_autograd_asks_for(geot):
    """A GCN's normalised adjacency does not require grad: no SDDMM; its transposed copy is kept while its content is
    unchanged.  Gradients against dense autograd in every combination."""
    from geot_amd import ops
    rng = np.random.default_rng(41)
    n, nnz, F = 3000, 120_000, 32
    di = dev(powerlaw_index(nnz, n, 3))
    si = dev(rng.integers(0, n, nnz).astype(np.int64))
    g = torch.rand(n, F, device="cuda")
    x0, w0 = torch.rand(n, F, device="cuda"), torch.rand(nnz, device="cuda")
    for wreq, xreq in ((False, True), (True, False), (True, True)):
        for it in range(3):                                              # repeated calls: the kept transposed weights
            x1, w1 = x0.clone().requires_grad_(xreq), w0 if not wreq else w0.clone().requires_grad_(True)
            x2, w2 = x0.clone().requires_grad_(xreq), w0.clone().requires_grad_(wreq)
            geot.gather_weight_scatter(si, di, w1, x1).backward(g)
            torch.zeros(n, F, device="cuda").index_add(0, di, x2[si] * w2[:, None]).backward(g)
            if xreq:
                assert torch.allclose(x1.grad, x2.grad, rtol=1e-4, atol=1e-4)
            else:
                assert x1.grad is None
            if wreq:
                assert torch.allclose(w1.grad, w2.grad, rtol=1e-4, atol=1e-4)
    w0.mul_(0.5)                                                         # in-place edit of the static weight: seen
    x1, x2 = x0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
    geot.gather_weight_scatter(si, di, w0, x1).backward(g)
    torch.zeros(n, F, device="cuda").index_add(0, di, x2[si] * w0[:, None]).backward(g)
    assert torch.allclose(x1.grad, x2.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("nodes,nnz,rowbytes,wmode,heads,R,units,slab_bytes", [
    (40_000, 3_000_000, 1024, 2, 4, 0, 0, 0),            # the library's own R / units / 2-MiB slabs
    (70_000, 2_500_000, 512, 1, 1, 0, 0, 0),
    (300, 200_000, 256, 0, 1, 4, 8, 256 * 37),           # few units: many rounds; tiny slabs
    (50, 300_000, 256, 1, 1, 3, 16, 0),                  # almost everything is a split hub
    (2_000, 400_000, 512, 1, 1, 5, 32, 512 * 11),
    (232_965, 20_000_000, 1024, 2, 4, 0, 0, 0),          # configs[3]'s node count
])
def test_device_plan_builder_is_bit_identical_to_the_aten_formulation(geot, nodes, nnz, rowbytes, wmode, heads, R, units, slab_bytes):
    """Phase A as device code (csrc/seg_plan.hip: row pointers by binary search, one scan of (virtual rows, split, carry
    slots), greedy grouping by pointer doubling, two radix sorts) against the ATen formulation it replaced (host_plan.cpp
    slab_build_aten, whose arrays the numpy emulation of tests/test_slab_plan.py validates): every array and scalar equal."""
    from geot_amd import ops, slab
    rng = np.random.default_rng(nodes + R)
    di = powerlaw_index(nnz, nodes, nodes + 1)
    di[: nnz // 4] = di[nnz // 4]                        # a hub that is split into virtual rows
    di = np.sort(di)
    di[di == 5] = 6                                      # an empty key
    si = rng.integers(-3, nodes + 3, nnz).astype(np.int64)      # a few out-of-range sources (clamped to the end slabs)
    d_si, d_di = dev(si), dev(di)
    plans = {}
    for name, builder in (("device", 0), ("aten", 1)):
        old = ops.set_option("slab_builder", builder)
        try:
            plans[name] = slab.build_plan(d_si, d_di, nodes, nodes, rowbytes, wmode, heads, slab_bytes=slab_bytes, rows_per_group=R or None,
                                          units=units or None)
        finally:
            ops.set_option("slab_builder", old)
    a, b = plans["device"], plans["aten"]
    assert a.meta == b.meta
    for f in ("n_groups", "n_vrows", "n_carry", "n_split", "nnz", "units", "rows_per_group", "slab_shift", "n_slabs"):
        assert getattr(a.struct, f) == getattr(b.struct, f), f
    assert a.struct.n_split >= 1
    sizes = {"e_src": nnz, "e_dl": nnz, "e_perm": nnz, "g_begin": a.struct.n_groups + 1, "g_vrow0": a.struct.n_groups, "g_nv": a.struct.n_groups,
             "v_out": a.struct.n_vrows, "v_row": a.struct.n_vrows, "v_total": a.struct.n_vrows, "c_row": a.struct.n_split,
             "c_first": a.struct.n_split, "c_count": a.struct.n_split, "c_total": a.struct.n_split}
    for name, n in sizes.items():
        ta, tb = a.tensors[name], b.tensors[name]
        assert ta.dtype == tb.dtype and ta.numel() == tb.numel() == n, (name, ta.dtype, tb.dtype, ta.numel(), tb.numel(), n)
        assert torch.equal(ta, tb), name


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("nodes,nnz,H,Fh", [(40_000, 3_000_000, 4, 64), (70_000, 2_500_000, 1, 128), (20_000, 2_000_000, 1, 512),
                                           (3000, 400_000, 2, 64), (30_000, 2_000_000, 1, 64)])
def test_slab_16bit_storage_fp32_accumulate(geot, oracle, dtype, nodes, nnz, H, Fh):
    """half / bfloat16 storage through the source-blocked kernel: 8 elements per 16-byte lane, fp32 accumulators in LDS,
    weights in the storage type, ONE rounding when a row is written - the semantics of the per-edge kernels and of the
    reference's CPU path (csrc/cpu/index_scatter_cpu.cpp:78-86,114-116).  Rows of 128 B (the last case) .. 1024 B."""
    from geot_amd import slab
    rng = np.random.default_rng(nodes + H)
    di = powerlaw_index(nnz, nodes, nodes)
    di[: nnz // 20] = di[nnz // 20]                                     # a hub that is split (fp32 carry slots)
    di = np.sort(di)
    di[di == 7] = 8                                                     # an empty key
    si = rng.integers(0, nodes, nnz).astype(np.int64)
    F = H * Fh
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    scale = 1.0 / 16 if dtype == torch.float16 else 1.0                 # (the hub's sum must stay below float16's 65504)
    x = torch.from_numpy(rng.random((nodes, H, Fh), dtype=np.float32) * scale).to(dtype)
    d_si, d_di, d_x = dev(si), dev(di), x.cuda()
    R = slab.rows_per_group(2 if H > 1 else 1, H, dtype)
    assert R == slab.rows_per_group(2 if H > 1 else 1, H, torch.float32) // 2 or R >= 1
    out = torch.empty(nodes, H, Fh, device="cuda", dtype=dtype)

    def check(got, hi, what):
        got = got.float().cpu().numpy().reshape(hi.shape)
        assert np.all(np.abs(got - hi) <= ulp * np.abs(hi) + 1e-6), (what, float(np.max(np.abs(got - hi) / (np.abs(hi) + 1e-6))))
        assert np.all(got[hi == 0] == 0), what

    if H == 1:
        w = torch.from_numpy(rng.random(nnz, dtype=np.float32)).to(dtype)
        plan = slab.build_plan(d_si, d_di, nodes, nodes, F * 2, 1, 1, rows_per_group=R)
        slab.slab_spmm_out(plan, w.cuda(), 1, d_x, out, 1, Fh)
        check(out, oracle.gather_weight_scatter(si, di, w.float().numpy(), x.float().numpy().reshape(nodes, F), rows=nodes, acc64=True), "gws")
        again = torch.empty_like(out)
        slab.slab_spmm_out(plan, w.cuda(), 1, d_x, again, 1, Fh)
        assert torch.equal(out, again)                                  # deterministic
        wp = w.cuda()[plan.tensors["e_perm"].long()].contiguous()       # a static weight, permuted into plan order (mode 4)
        slab.slab_spmm_out(plan, wp, 4, d_x, again, 1, Fh)
        assert torch.equal(out, again)
        slab.slab_spmm_out(plan, None, 0, d_x, out, 1, Fh)
        hi = oracle.gather_scatter(si, di, x.float().numpy().reshape(nodes, F), rows=nodes, acc64=True)
        check(out, hi, "gs")
        for red, tred in (("max", "amax"), ("mean", "mean")):
            slab.slab_spmm_out(plan, None, 0, d_x, out, 1, Fh, reduce=red)
            want = torch.zeros(nodes, F, device="cuda").scatter_reduce(0, d_di[:, None].expand(-1, F), d_x.view(nodes, F)[d_si].float(), tred,
                                                                       include_self=False)
            check(out, want.cpu().numpy().astype(np.float64), red)
    else:
        w = torch.from_numpy(rng.random((nnz, H), dtype=np.float32)).to(dtype)
        hi = oracle.mh_spmm(si, di, w.float().numpy(), x.float().numpy(), rows=nodes, acc64=True)
        plan = slab.build_plan(d_si, d_di, nodes, nodes, F * 2, 2, H, rows_per_group=R)
        slab.slab_spmm_out(plan, w.cuda(), 2, d_x, out, H, Fh)
        check(out, hi, "mh edge-major")
        slab.slab_spmm_out(plan, w.t().contiguous().cuda(), 3, d_x, out, H, Fh)
        check(out, hi, "mh head-major")


def test_host_layer_routes_16bit_dense_graphs_to_the_source_blocked_kernel(geot):
    from geot_amd import ops
    nodes, nnz, H, Fh = 30_000, 3_000_000, 4, 64
    di = dev(powerlaw_index(nnz, nodes, 3))
    g = torch.Generator(device="cuda").manual_seed(1)
    si = torch.randint(0, nodes, (nnz,), device="cuda", generator=g)
    x = torch.rand(nodes, H, Fh, device="cuda", generator=g).bfloat16()
    w = torch.rand(nnz, H, device="cuda", generator=g).bfloat16()
    old = ops.set_option("slab_mode", "always")
    try:
        ops.clear_caches()
        st0 = ops.stats()
        got = geot.mh_spmm(si, di, w, x)
        assert ops.stats()["slab_calls"] == st0["slab_calls"] + 1 and got.dtype == torch.bfloat16
        ops.set_option("slab_mode", "never")
        ref = geot.mh_spmm(si, di, w, x)                                 # the per-edge kernels, same storage type
        assert torch.allclose(got.float(), ref.float(), rtol=2.0 ** -6, atol=1e-6)
        ops.set_option("slab_mode", "always")
        x2 = x.view(nodes, H * Fh)
        w1 = w[:, 0].contiguous()
        got = geot.gather_weight_scatter(si, di, w1, x2)
        ops.set_option("slab_mode", "never")
        assert torch.allclose(got.float(), geot.gather_weight_scatter(si, di, w1, x2).float(), rtol=2.0 ** -6, atol=1e-6)
    finally:
        ops.set_option("slab_mode", old)
        ops.clear_caches()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("nodes,nnz,F", [(30_000, 2_000_000, 128), (20_000, 2_000_000, 512), (300, 200_000, 256)])
def test_slab_sddmm_16bit_storage(geot, oracle, dtype, nodes, nnz, F):
    """d/dweight over the plan with half / bfloat16 operands: fp32 dot products, the result rounded once to the storage type."""
    from geot_amd import hip, slab
    rng = np.random.default_rng(nodes + F)
    di = powerlaw_index(nnz, nodes, nodes + 2)
    di[: nnz // 6] = di[nnz // 6]
    di = np.sort(di)
    src_rows = nodes + 11
    si = rng.integers(0, src_rows, nnz).astype(np.int64)
    m1 = torch.from_numpy(rng.standard_normal((nodes, F)).astype(np.float32)).to(dtype)
    m2 = torch.from_numpy(rng.standard_normal((src_rows, F)).astype(np.float32)).to(dtype)
    plan = slab.build_plan(dev(si), dev(di), nodes, src_rows, F * 2, 1, 1, rows_per_group=slab.rows_per_group(1, 1, dtype))
    out = torch.full((nnz,), float("nan"), device="cuda", dtype=dtype)
    slab.slab_sddmm_out(plan, m1.cuda(), m2.cuda(), out)
    assert not torch.isnan(out.float()).any()
    ref = oracle.sddmm_coo(si, di, m1.float().numpy(), m2.float().numpy(), acc64=True)
    mag = oracle.sddmm_coo(si, di, m1.float().abs().numpy(), m2.float().abs().numpy(), acc64=True)
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    got = out.float().cpu().numpy()
    assert np.all(np.abs(got - ref) <= ulp * np.abs(ref) + 2e-5 * mag + 1e-6)
    tile = hip.sddmm_coo_out(dev(si), dev(di), m1.cuda(), m2.cuda(), torch.empty(nnz, device="cuda", dtype=dtype))
    assert torch.allclose(out.float(), tile.float(), rtol=4 * ulp, atol=4 * ulp)
    again = torch.empty_like(out)
    slab.slab_sddmm_out(plan, m1.cuda(), m2.cuda(), again)
    assert torch.equal(out, again)


def test_a_plan_that_loses_to_the_per_edge_kernels_is_dropped(geot, oracle, request):
    """The density rule that routes a graph to the source-blocked kernels was calibrated on uniform-random sources.  A dense graph
    whose sources sit next to their destinations is served much faster by the per-edge kernels (its gathers hit in L2; the plan's
    chip-wide slab walk only makes waves wait): the first call over a plan runs both and keeps the faster.  Results never change."""
    from geot_amd import ops, slab
    rng = np.random.default_rng(21)
    nodes, nnz, F = 40_000, 12_000_000, 128
    di = powerlaw_index(nnz, nodes, 4)
    si = np.clip(di + rng.integers(-300, 301, nnz), 0, nodes - 1).astype(np.int64)     # sources within +-300 rows
    w = rng.random(nnz, dtype=np.float32)
    x = rng.random((nodes, F), dtype=np.float32)
    assert slab.worthwhile(nnz, nodes, nodes, F * 4)
    d_si, d_di, d_w, d_x = dev(si), dev(di), dev(w), dev(x)
    hi = oracle.gather_weight_scatter(si, di, w, x, rows=nodes, acc64=True)
    # (1) the routing itself notices the locality - the groups of this graph touch a few per cent of the source slabs - and never
    #     builds a plan: no Phase A, no trial, the per-edge kernels from the first call on
    ops.clear_caches()
    st0 = ops.stats()
    outs = [geot.gather_weight_scatter(d_si, d_di, d_w, d_x) for _ in range(3)]
    st = ops.stats()
    assert st["plans_declined"] == st0["plans_declined"] + 1 and st["plans_built"] == st0["plans_built"] and st["plan_trials"] == st0["plan_trials"]
    assert st["last_coverage_permille"] < 200 and st["slab_calls"] == st0["slab_calls"]
    for o in outs:
        close(o, hi, "local dense graph, declined by the coverage probe")
    # (2) with the probe off the plan is built, TRIED on its first use, loses and is dropped
    old_cov = ops.set_option("slab_min_coverage_pct", 0)
    request.addfinalizer(lambda: ops.set_option("slab_min_coverage_pct", old_cov))
    ops.clear_caches()
    st0 = ops.stats()
    outs = [geot.gather_weight_scatter(d_si, d_di, d_w, d_x) for _ in range(4)]
    st = ops.stats()
    assert st["plans_built"] == st0["plans_built"] + 1 and st["plan_trials"] == st0["plan_trials"] + 1
    assert st["plans_rejected"] == st0["plans_rejected"] + 1, "the per-edge kernels are several times faster on this graph"
    assert st["slab_calls"] == st0["slab_calls"] + 1                       # the trial's one run over the plan
    assert st["cache_bytes"] < 64 << 20                                    # the rejected plan's arrays (9 B per edge) are gone
    for o in outs:
        close(o, hi, "local dense graph")
    assert torch.equal(outs[2], outs[3])
    # the decision covers the SDDMM over the same edge list, and a captured graph never tries an undecided plan
    g = torch.rand(nodes, F, device="cuda")
    calls = ops.stats()["slab_calls"]
    torch.ops.geot.sddmm_coo_impl(d_si, d_di, g, d_x)
    assert ops.stats()["slab_calls"] == calls and ops.stats()["plan_trials"] == st["plan_trials"]
