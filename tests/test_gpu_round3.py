"""Round-3 parity cases (`-m gpu`).

* the DESCENT GUARD: an index written behind its version counter (`.data`) keeps its remembered "ascending" fact, yet the
  call that meets it still returns the right sums - the sorted kernels ballot "key below its predecessor" while they stage
  the keys and the fix-up launch repairs the call with atomics, the reference's own formulation
  (csrc/cuda/index_scatter_kernel.cuh:180,197); reductions / dtypes without float atomics are NaN-filled and the next call
  raises; the C ABI called with sorted=1 on an unsorted index is correct as well;
* BASELINE.json configs[3] at FULL size (mh_spmm, 232 965 nodes, 114.6 M edges, H=4 x F=64), auto-dispatched: the second call
  takes the source-blocked plan; determinism, exact zero rows, sampled rows against float64, column checksums, and agreement
  with the per-edge kernels (comparator of test/test_mh_spmm.py:4-10: index_select * w -> index_add).
"""
import sys
import warnings

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.fixture(scope="module")
def geot():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import geot_amd
    return geot_amd


@pytest.fixture()
def fresh(geot):
    """No remembered facts before or after (an alarm drops them anyway), warnings about repairs kept out of the log."""
    from geot_amd import ops
    ops.clear_caches()
    torch.cuda.synchronize()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        yield ops
        torch.cuda.synchronize()
        try:
            ops.stats()
            ops.clear_caches()
        except RuntimeError:
            pass


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def device_powerlaw(nnz, keys, seed):
    sys.path.insert(0, ROOT)
    from bench import powerlaw_index as gen
    return gen(nnz, keys, seed, torch.device("cuda"))


def close(got, hi, mag, what):
    got = got.detach().cpu().numpy()
    assert got.shape == hi.shape, (what, got.shape, hi.shape)
    bound = RTOL * mag + 1e-30
    err = np.abs(got.astype(np.float64) - hi.astype(np.float64))
    assert np.all(err <= bound), f"{what}: max err/bound = {np.max(err / bound):.3g}"


def ascending_index(rng, nnz, K):
    index = np.sort(rng.integers(0, K, nnz)).astype(np.int64)
    index[-1] = K - 1
    return index


def scramble(rng, index, swaps=40, dist=700):
    """A copy of `index` with a few local descents; first and last key unchanged (the row rule still gives K rows)."""
    bad = index.copy()
    at = rng.integers(1, index.size - dist - 2, swaps)
    bad[at], bad[at + dist] = index[at + dist].copy(), index[at].copy()
    assert (bad[:-1] > bad[1:]).sum() > 0 and bad[-1] == index[-1]
    return bad


def write_behind_the_version_counter(t_index, bad):
    v = t_index._version
    t_index.data.copy_(dev(bad))
    assert t_index._version == v                   # autograd's counter has not moved: the remembered facts look valid


@pytest.mark.parametrize("F,dtype", [(64, torch.float32), (4, torch.float32), (17, torch.float32), (32, torch.float64)])
def test_descents_behind_the_version_counter_index_scatter(geot, oracle, fresh, F, dtype):
    ops = fresh
    rng = np.random.default_rng(300 + F)
    nnz, K = 400_000, 11_000
    index = ascending_index(rng, nnz, K)
    src = rng.random((nnz, F)).astype(np.float32 if dtype == torch.float32 else np.float64)
    t_index, t_src = dev(index), dev(src)
    want0 = oracle.index_scatter(index, src.astype(np.float32), rows=K, acc64=True)
    alarms0 = ops.stats()["alarms"]
    out0 = geot.index_scatter(0, t_src, t_index, "sum", True)                 # probed: ascending, remembered
    close(out0, want0, want0, "ascending")
    torch.cuda.synchronize()
    geot.index_scatter(0, t_src, t_index, "sum", True)
    assert ops.stats()["alarms"] == alarms0                                   # no false alarm on an ascending index
    bad = scramble(rng, index)
    write_behind_the_version_counter(t_index, bad)
    order = np.argsort(bad, kind="stable")
    want = oracle.index_scatter(bad[order], src[order].astype(np.float32), rows=K, acc64=True)
    probes = ops.stats()["probes"]
    out = geot.index_scatter(0, t_src, t_index, "sum", True)                  # stale fact -> atomic-free kernels -> repaired
    close(out, want, want, "descents written through .data")
    assert out.shape == (K, F)
    torch.cuda.synchronize()
    out2 = geot.index_scatter(0, t_src, t_index, "sum", True)                 # the host has seen the alarm: probe + sort path
    st = ops.stats()
    assert st["alarms"] == alarms0 + 1 and st["probes"] == probes + 1 and st["sorts"] >= 1
    close(out2, want, want, "after the alarm")
    assert torch.equal(out2, geot.index_scatter(0, t_src, t_index, "sum", True))        # deterministic again
    # the workspace's control words were left zero: an ascending index on the same stream is served as before
    close(geot.index_scatter(0, t_src, dev(index), "sum", True), want0, want0, "ascending again")
    torch.cuda.synchronize()
    geot.index_scatter(0, t_src, dev(index), "sum", True)
    assert ops.stats()["alarms"] == alarms0 + 1


def test_descents_behind_the_version_counter_gather_ops(geot, oracle, fresh):
    ops = fresh
    rng = np.random.default_rng(77)
    nnz, K, F, H = 300_000, 9_000, 32, 4
    index = ascending_index(rng, nnz, K)
    si = rng.integers(0, K, nnz).astype(np.int64)
    w = rng.random(nnz, dtype=np.float32)
    wh = rng.random((nnz, H), dtype=np.float32)
    x = rng.random((K, F), dtype=np.float32)
    x3 = rng.random((K, H, 8), dtype=np.float32)
    t_si, t_w, t_wh, t_x, t_x3 = dev(si), dev(w), dev(wh), dev(x), dev(x3)
    bad = scramble(rng, index)
    order = np.argsort(bad, kind="stable")
    cases = {
        "gather_scatter": (lambda di: geot.gather_scatter(t_si, di, t_x),
                           oracle.gather_scatter(si[order], bad[order], x, rows=K, acc64=True)),
        "gather_weight_scatter": (lambda di: geot.gather_weight_scatter(t_si, di, t_w, t_x),
                                  oracle.gather_weight_scatter(si[order], bad[order], w[order], x, rows=K, acc64=True)),
        "mh_spmm": (lambda di: geot.mh_spmm(t_si, di, t_wh, t_x3),
                    oracle.mh_spmm(si[order], bad[order], wh[order], x3, rows=K, acc64=True)),
        "mh_spmm head-major": (lambda di: geot.mh_spmm(t_si, di, t_wh.t().contiguous(), t_x3),
                               oracle.mh_spmm(si[order], bad[order], wh[order], x3, rows=K, acc64=True)),
    }
    alarms = ops.stats()["alarms"]
    for name, (call, want) in cases.items():
        t_di = dev(index)
        call(t_di)                                                            # facts: ascending
        write_behind_the_version_counter(t_di, bad)
        got = call(t_di)
        close(got.reshape(K, -1), want.reshape(K, -1), want.reshape(K, -1), name)
        torch.cuda.synchronize()
        close(call(t_di).reshape(K, -1), want.reshape(K, -1), want.reshape(K, -1), name + " after the alarm")
        alarms += 1
        assert ops.stats()["alarms"] == alarms, name


def test_descents_without_float_atomics_are_loud_not_wrong(geot, fresh):
    """max / mean / 16-bit storage have no atomic to fall back on: the output of the call that meets the stale fact is NaN
    from end to end and the next call on the thread raises, naming the cause; the call after that is right."""
    ops = fresh
    rng = np.random.default_rng(5)
    nnz, K, F = 200_000, 5_000, 64
    index = ascending_index(rng, nnz, K)
    bad = scramble(rng, index)
    src = torch.rand(nnz, F, device="cuda")
    for reduce, dtype in (("max", torch.float32), ("mean", torch.float32), ("sum", torch.bfloat16)):
        t_index = dev(index)
        s = src.to(dtype)
        geot.index_scatter(0, s, t_index, reduce, True)
        write_behind_the_version_counter(t_index, bad)
        out = geot.index_scatter(0, s, t_index, reduce, True)
        assert bool(torch.isnan(out.float()).all()), (reduce, dtype)
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="DESCENTS"):
            geot.index_scatter(0, s, t_index, reduce, True)
        good = geot.index_scatter(0, s, t_index, reduce, True)                # facts dropped: probed, sort path
        assert not bool(torch.isnan(good.float()).any())
        want = torch.zeros(K, F, device="cuda").scatter_reduce(0, dev(bad)[:, None].expand(-1, F), s.float(),
                                                                {"max": "amax", "mean": "mean", "sum": "sum"}[reduce], include_self=False)
        assert torch.allclose(good.float(), want, rtol=2e-2 if dtype == torch.bfloat16 else 1e-5, atol=1e-5), (reduce, dtype)


def test_c_abi_sorted_promise_on_an_unsorted_index(geot, oracle):
    """The C ABI takes `sorted` on faith (include/geot_hip.h) - as the reference's index_scatter_cuda does.  A wrong promise
    still adds up; a grid larger than the chip (the repair's bounded wait) and the narrow-row kernels included."""
    from geot_amd import hip
    rng = np.random.default_rng(9)
    for nnz, K, F, opt in ((1_000_000, 50_000, 64, None), (600_000, 20_000, 2, None), (600_000, 20_000, 3, ("narrow", 2)),
                           (40_000_000, 1_000_000, 64, None)):
        index = ascending_index(rng, nnz, K)
        bad = scramble(rng, index, swaps=200, dist=900)
        t_bad = dev(bad)
        src = torch.rand(nnz, F, device="cuda")
        out = torch.full((K, F), 7.0, device="cuda")
        if opt:
            hip.set_option(*opt)
        try:
            hip.index_scatter_out(t_bad, src, out, sorted=True)
            again = torch.full((K, F), -3.0, device="cuda")
            hip.index_scatter_out(t_bad, src, again, sorted=True)                  # control words were left zero
        finally:
            if opt:
                hip.set_option(opt[0], 1)
        want = torch.zeros(K, F, device="cuda", dtype=torch.float64).index_add_(0, t_bad, src.double())
        assert torch.allclose(out.double(), want, rtol=1e-5, atol=1e-5), (nnz, F)
        assert torch.allclose(again.double(), want, rtol=1e-5, atol=1e-5), (nnz, F)
        t_good = dev(index)
        hip.index_scatter_out(t_good, src, out, sorted=True)                       # and the atomic-free path still is what runs
        hip.index_scatter_out(t_good, src, again, sorted=True)
        assert torch.equal(out, again)
        del src, out, again, want
    torch.cuda.empty_cache()


def test_descent_guard_inside_a_captured_graph(geot, oracle, fresh):
    """A hipGraph replays kernels, not host decisions: the guard is part of the kernels, so a replay on an index that has
    meanwhile been scrambled in place is still right."""
    rng = np.random.default_rng(21)
    nnz, K, F = 200_000, 4_000, 64
    index = ascending_index(rng, nnz, K)
    t_index, src = dev(index), torch.rand(nnz, F, device="cuda")
    geot.index_scatter(0, src, t_index, "sum", True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = geot.index_scatter(0, src, t_index, "sum", True)
    g.replay()
    torch.cuda.synchronize()
    assert torch.allclose(out, torch.zeros(K, F, device="cuda").index_add_(0, t_index, src), rtol=1e-5, atol=1e-5)
    bad = scramble(rng, index)
    write_behind_the_version_counter(t_index, bad)
    g.replay()
    torch.cuda.synchronize()
    want = torch.zeros(K, F, device="cuda", dtype=torch.float64).index_add_(0, t_index, src.double())
    assert torch.allclose(out.double(), want, rtol=1e-5, atol=1e-5)


def test_cfg4_full_size_properties(geot):
    """BASELINE.json configs[3] at full size, as dispatched: mh_spmm, 232 965 nodes, 114 615 892 edges, H=4 x F=64."""
    from geot_amd import ops
    nodes, nnz, H, F = 232_965, 114_615_892, 4, 64
    ops.clear_caches()
    di = device_powerlaw(nnz, nodes, 11)
    g = torch.Generator(device="cuda")
    g.manual_seed(12)
    si = torch.randint(0, nodes, (nnz,), device="cuda", generator=g)
    w = torch.rand(nnz, H, device="cuda", generator=g)
    x = torch.rand(nodes, H, F, device="cuda", generator=g)
    st0 = ops.stats()
    first = geot.mh_spmm(si, di, w, x)                       # first sighting of the edge list: per-edge kernels
    st1 = ops.stats()
    out = geot.mh_spmm(si, di, w, x)                         # second: Phase A, then the source-blocked kernel
    st2 = ops.stats()
    assert st1["slab_calls"] == st0["slab_calls"] and st2["slab_calls"] == st1["slab_calls"] + 1 and st2["plans_built"] == st1["plans_built"] + 1
    assert out.shape == (nodes, H, F)
    again = geot.mh_spmm(si, di, w, x)
    assert ops.stats()["slab_calls"] == st2["slab_calls"] + 1 and torch.equal(out, again)        # deterministic
    del again
    # the two kernel families agree (<= 1e-5 of the sum of |contributions|; data are non-negative: that sum is the result)
    assert bool(((out - first).abs() <= 1e-5 * first.abs() + 1e-30).all())
    # ... and both weight layouts (head-major is transposed once per call in front of the source-blocked kernel)
    hm = geot.mh_spmm(si, di, w.t().contiguous(), x)
    assert torch.equal(hm, out)
    del hm, first
    counts = torch.bincount(di, minlength=nodes)
    assert out[counts == 0].abs().sum().item() == 0          # rows without edges: exactly zero
    offs = torch.cumsum(counts, 0) - counts
    gen = torch.Generator().manual_seed(3)
    pick = [int(counts.argmax()), 0, nodes - 1, nodes // 2] + torch.randint(0, nodes, (150,), generator=gen).tolist()
    for k in pick:                                           # comparator of test/test_mh_spmm.py:4-10, in float64
        e = slice(int(offs[k]), int(offs[k] + counts[k]))
        msg = x[si[e]].double() * w[e].double()[:, :, None]
        assert torch.allclose(out[k].double(), msg.sum(0), rtol=1e-5, atol=1e-6), k
    # checksum of checksums: column sums of the result == (per-node, per-head total weight) contracted with x
    cw = torch.zeros(nodes, H, dtype=torch.float64, device="cuda").index_add_(0, si, w.double())
    want = torch.einsum("nh,nhf->hf", cw, x.double())
    got = out.double().sum(0)
    assert torch.allclose(got, want, rtol=1e-8), (got - want).abs().max()
    ops.clear_caches()
    torch.cuda.empty_cache()


def test_cached_artefacts_follow_their_sources_and_a_byte_budget(geot):
    """The host layer's caches hold derived artefacts only: the tensors they were derived from are referenced weakly (an
    entry dies with its source instead of pinning the user's edge list) and all artefacts together stay under a byte budget."""
    from geot_amd import ops
    ops.clear_caches()
    nnz, nodes = 2_000_000, 50_000
    g = torch.Generator(device="cuda").manual_seed(0)
    si = torch.randint(0, nodes, (nnz,), device="cuda", generator=g)
    di = torch.sort(torch.randint(0, nodes, (nnz,), device="cuda", generator=g)).values
    perm, s_sorted, d_perm = torch.ops.geot.transpose_edges(si, di)
    st = ops.stats()
    assert st["transposed"] == 1 and st["cache_bytes"] >= 3 * 8 * nnz
    assert torch.equal(torch.ops.geot.transpose_edges(si, di)[0], perm) and ops.stats()["transposes"] == st["transposes"]    # served from the cache
    before = torch.cuda.memory_allocated()
    del si
    st2 = ops.stats()                                      # (stats sweep entries whose source has died)
    assert st2["transposed"] == 0 and st2["cache_bytes"] < st["cache_bytes"]
    del perm, s_sorted, d_perm
    assert torch.cuda.memory_allocated() <= before - 4 * 8 * nnz + (1 << 20)          # the edge list AND the artefacts are gone
    # budget: artefacts of 48 MB against a 16 MiB budget are not kept (the call still returns them)
    old = ops.set_option("cache_mb", 16)
    try:
        si = torch.randint(0, nodes, (nnz,), device="cuda", generator=g)
        a = torch.ops.geot.transpose_edges(si, di)
        assert ops.stats()["cache_bytes"] <= 16 << 20
        b = torch.ops.geot.transpose_edges(si, di)
        assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
        x = torch.rand(nodes, 32, device="cuda", requires_grad=True)               # ... and training through it still works
        y = geot.gather_scatter(si, di, x)
        y.sum().backward()
        want = torch.zeros(nodes, device="cuda").index_add_(0, si, torch.ones(nnz, device="cuda"))
        assert torch.allclose(x.grad[:, 0], want)
    finally:
        ops.set_option("cache_mb", old)
        ops.clear_caches()


# ---- in-kernel hand-off of the tile carries (seg_tile_kernel "hand-off"): the run that straddles tiles is finished by the
# ---- tile in which it ends; the second launch only tidies up
def _segment_sums(index, values64, rows):
    """float64 sums per key of an ASCENDING index (a segmented reduction: index_add_ into a hub's one row takes seconds)."""
    return torch.segment_reduce(values64, "sum", lengths=torch.bincount(index, minlength=rows), axis=0, unsafe=True)


def _handoff_cases():
    rng = np.random.default_rng(33)
    cases = []
    for name, nnz, K, F, dtype in (("powerlaw F=64", 3_000_000, 300_000, 64, torch.float32),
                                   ("every run spans several tiles", 4_000_000, 2_000, 64, torch.float32),
                                   ("mixed: hubs over dozens of tiles + short runs", 2_000_000, 40_000, 64, torch.float32),
                                   ("F=16", 1_500_000, 100_000, 16, torch.float32),
                                   ("F=256 (a wave per row, 16 loads in flight)", 1_000_000, 50_000, 256, torch.float32),
                                   ("F=48 (ragged lanes)", 1_000_000, 30_000, 48, torch.float32),
                                   ("bf16 F=64 (8 elements per lane)", 2_000_000, 20_000, 64, torch.bfloat16)):
        if name.startswith("mixed"):
            from conftest import powerlaw_index
            index = powerlaw_index(nnz, K, 5)
            index[: nnz // 3] = index[nnz // 3]                      # one hub over a third of the edges (hundreds of tiles)
            index = np.sort(index)
        elif name.startswith("every"):
            index = np.sort(rng.integers(0, K, nnz)).astype(np.int64)
        else:
            from conftest import powerlaw_index
            index = powerlaw_index(nnz, K, F)
        index[-1] = K - 1
        cases.append((name, index, K, F, dtype))
    return cases


def test_handoff_results_with_data_that_changes_every_call(geot):
    """The carry rows travel between workgroups through write-through stores and flag words while the kernel runs.  A stale
    read - a cached line of the PREVIOUS call's carry row - can only show when the data differ from call to call: the same
    index, new values every call, every result against float64; repeated calls bit-equal; the classic second pass agrees."""
    from geot_amd import hip
    for name, index, K, F, dtype in _handoff_cases():
        t_index = dev(index)
        nnz = index.size
        base = torch.rand(nnz, F, device="cuda")
        ref0 = _segment_sums(t_index, base.to(dtype).double(), K)
        tol = 2.0 ** -7 if dtype == torch.bfloat16 else 1e-5
        for it in range(12):
            scale = float(it + 1)
            src = (base * scale).to(dtype)
            want = ref0 * scale if dtype == torch.float32 else _segment_sums(t_index, src.double(), K)
            out = geot.index_scatter(0, src, t_index, "sum", True)
            assert torch.allclose(out.double(), want, rtol=tol, atol=1e-6 * scale), (name, it)
            if it in (3, 7):
                assert torch.equal(out, geot.index_scatter(0, src, t_index, "sum", True)), (name, "not reproducible")
                hip.set_option("handoff", 0)
                try:
                    classic = geot.index_scatter(0, src, t_index, "sum", True)
                finally:
                    hip.set_option("handoff", 1)
                assert torch.allclose(out.double(), classic.double(), rtol=tol, atol=1e-6 * scale), (name, "classic second pass disagrees")
        mx = geot.index_scatter(0, src, t_index, "max", True)
        want = torch.zeros(K, F, device="cuda").scatter_reduce(0, t_index[:, None].expand(-1, F), src.float(), "amax", include_self=False)
        assert torch.equal(mx.float(), want), (name, "max")
        # mean: the partial runs' edge counts travel with their rows (fp32 storage; 16-bit storage keeps the classic second pass)
        counts = torch.bincount(t_index, minlength=K).clamp_min(1).double()[:, None]
        for it in range(4):
            src = (base * float(it + 1)).to(dtype)
            want = _segment_sums(t_index, src.double(), K) / counts
            mean = geot.index_scatter(0, src, t_index, "mean", True)
            assert torch.allclose(mean.double(), want, rtol=tol, atol=1e-6 * (it + 1)), (name, "mean", it)
        assert torch.equal(mean, geot.index_scatter(0, src, t_index, "mean", True)), (name, "mean not reproducible")
        hip.set_option("handoff", 0)
        try:
            classic = geot.index_scatter(0, src, t_index, "mean", True)
        finally:
            hip.set_option("handoff", 1)
        assert torch.allclose(mean.double(), classic.double(), rtol=tol, atol=1e-6 * (it + 1)), (name, "mean: classic second pass disagrees")
        del base, ref0, src, want, out, mx, mean, classic


def test_handoff_gives_up_gracefully_and_the_second_launch_finishes_the_call(geot):
    """`handoff_tries` = 0: a tile samples its predecessor's flag ONCE and, if it is not up yet, leaves the run to the second
    launch, which then redoes every straddling run from the carry rows (and lowers the flags): same results, whatever mix of
    in-kernel and deferred runs the timing of a call produces; afterwards the normal mode works on the same workspace."""
    from geot_amd import hip
    name, index, K, F, dtype = _handoff_cases()[2]
    t_index = dev(index)
    src = torch.rand(index.size, F, device="cuda")
    want = _segment_sums(t_index, src.double(), K)
    want_mean = want / torch.bincount(t_index, minlength=K).clamp_min(1).double()[:, None]
    hip.set_option("handoff_tries", 0)
    try:
        for _ in range(6):
            out = geot.index_scatter(0, src, t_index, "sum", True)
            assert torch.allclose(out.double(), want, rtol=1e-5, atol=1e-6)
            mean = geot.index_scatter(0, src, t_index, "mean", True)         # (deferred runs: the second launch divides by the counts)
            assert torch.allclose(mean.double(), want_mean, rtol=1e-5, atol=1e-6)
    finally:
        hip.set_option("handoff_tries", 400000)
    out = geot.index_scatter(0, src, t_index, "sum", True)
    assert torch.allclose(out.double(), want, rtol=1e-5, atol=1e-6)
    assert torch.equal(out, geot.index_scatter(0, src, t_index, "sum", True))


def test_handoff_inside_a_replayed_graph(geot):
    """A captured call carries its flag tag with it: every replay re-uses it, so the second launch lowers the flags again.
    Replays with new data, interleaved with eager calls of another shape on the same stream (same workspace)."""
    name, index, K, F, dtype = _handoff_cases()[0]
    t_index = dev(index)
    src = torch.rand(index.size, F, device="cuda")
    other_i = dev(np.sort(np.random.default_rng(1).integers(0, 5000, 700_000)).astype(np.int64))
    other_s = torch.rand(700_000, 32, device="cuda")
    geot.index_scatter(0, src, t_index, "sum", True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = geot.index_scatter(0, src, t_index, "sum", True)
    for it in range(6):
        src.copy_(torch.rand_like(src) * (it + 1))
        g.replay()
        want = _segment_sums(t_index, src.double(), K)
        assert torch.allclose(out.double(), want, rtol=1e-5, atol=1e-5), it
        o2 = geot.index_scatter(0, other_s, other_i, "sum", True)
        w2 = torch.zeros(int(other_i[-1]) + 1, 32, device="cuda", dtype=torch.float64).index_add_(0, other_i, other_s.double())
        assert torch.allclose(o2.double(), w2, rtol=1e-5, atol=1e-5), it


def test_handoff_under_uneven_load(geot):
    """The micro-architecture guide's advice for every inter-workgroup hand-off: test it under UNEVEN load, with data that
    changes, checking every word.  A second stream keeps the chip busy with streaming and compute kernels of varying length
    while the reductions run (their workgroups then start late, in bursts, and out of step with their neighbours)."""
    name, index, K, F, dtype = _handoff_cases()[2]                  # hubs over hundreds of tiles + short runs
    t_index = dev(index)
    nnz = index.size
    base = torch.rand(nnz, F, device="cuda")
    ref0 = _segment_sums(t_index, base.double(), K)
    noise_stream = torch.cuda.Stream()
    a = torch.rand(4096, 4096, device="cuda")
    big = torch.rand(64 << 20, device="cuda")
    main = torch.cuda.current_stream()
    for it in range(24):
        with torch.cuda.stream(noise_stream):
            for rep in range(1 + it % 4):                           # bursts of different length on the other stream
                if (it + rep) % 2:
                    a = torch.mm(a, a).clamp_(-1, 1)
                else:
                    big.mul_(1.0001)
        src = base * float(it + 1)
        out = geot.index_scatter(0, src, t_index, "sum", True)
        main.synchronize()
        assert torch.allclose(out.double(), ref0 * float(it + 1), rtol=1e-5, atol=1e-5 * (it + 1)), it
        mx = geot.index_scatter(0, src, t_index, "max", True)
        if it % 6 == 0:
            want = torch.zeros(K, F, device="cuda").scatter_reduce(0, t_index[:, None].expand(-1, F), src, "amax", include_self=False)
            assert torch.equal(mx, want), it
    torch.cuda.synchronize()
