"""Seeded fuzzing of the HIP path against the oracle: random sizes, key distributions (runs, gaps, hubs,
leading empty rows), feature widths, reductions and ops.  Deterministic (fixed seeds), a few seconds."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def random_index(rng, nnz):
    kind = rng.integers(0, 6)
    if kind == 0:                                   # uniform keys
        keys = int(rng.integers(1, max(2, nnz * 2)))
        idx = np.sort(rng.integers(0, keys, nnz))
    elif kind == 1:                                 # long runs
        keys = int(rng.integers(1, max(2, nnz // 50 + 2)))
        idx = np.sort(rng.integers(0, keys, nnz))
    elif kind == 2:                                 # sparse keys with gaps of random size
        idx = np.cumsum(rng.integers(0, int(rng.choice([2, 5, 40, 300])), nnz))
    elif kind == 3:                                 # one hub + noise
        hub = int(rng.integers(0, 20))
        idx = np.sort(np.concatenate([np.full(nnz - nnz // 4, hub), rng.integers(0, 40, nnz // 4)]))
    elif kind == 4:                                 # unit segments with an offset
        idx = np.arange(nnz) + int(rng.integers(0, 100))
    else:                                           # runs whose lengths are multiples of 64 / 16 (step boundaries)
        reps = int(rng.choice([16, 64, 128]))
        idx = np.repeat(np.arange(nnz // reps + 1), reps)[:nnz]
    return idx.astype(np.int64)


def close(got, hi, mag, tol=1e-5):
    got = got.detach().cpu().numpy().astype(np.float64)
    return got.shape == hi.shape and np.all(np.abs(got - hi) <= tol * mag + 1e-30) and np.all(got[mag == 0] == 0)


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_index_scatter(oracle, seed):
    import geot_amd as geot
    rng = np.random.default_rng(1000 + seed)
    for _ in range(40):
        nnz = int(rng.choice([1, 2, 3, 63, 64, 65, 500, 1023, 1024, 1025, 4096, 20_000, int(rng.integers(1, 60_000))]))
        F = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 24, 32, 33, 64, 96, 128, 130, 256, int(rng.integers(1, 300))]))
        index = random_index(rng, nnz)
        src = rng.standard_normal((nnz, F)).astype(np.float32)
        red = str(rng.choice(["sum", "sum", "sum", "mean", "min", "max"]))
        t_idx, t_src = torch.from_numpy(index).cuda(), torch.from_numpy(src).cuda()
        out = geot.index_scatter(0, t_src, t_idx, red, True)
        what = f"seed={seed} nnz={nnz} F={F} red={red} keys={index[-1] + 1}"
        if red == "sum":
            hi = oracle.index_scatter(index, src, acc64=True)
            mag = oracle.index_scatter(index, np.abs(src), acc64=True)
            assert close(out, hi, mag), what
            assert close(geot.index_scatter(0, t_src, t_idx, "sum", False), hi, mag), what + " (sorted=False)"
        else:
            ref = oracle.index_scatter_3pass(index, src, reduce=red)
            got = out.cpu().numpy()
            assert got.shape == ref.shape, what
            if red == "mean":
                assert np.allclose(got, ref, rtol=1e-4, atol=1e-5), what
            else:
                assert np.array_equal(got, ref), what


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_gather_ops(oracle, seed):
    import geot_amd as geot
    rng = np.random.default_rng(2000 + seed)
    for _ in range(25):
        nnz = int(rng.choice([1, 64, 65, 1000, 1024, 5000, int(rng.integers(1, 40_000))]))
        F = int(rng.choice([1, 3, 4, 8, 16, 32, 48, 64, 100, 128, int(rng.integers(1, 200))]))
        di = random_index(rng, nnz)
        nodes = int(di[-1]) + 1 + int(rng.integers(0, 50))
        si = rng.integers(0, nodes, nnz).astype(np.int64)
        w = rng.random(nnz, dtype=np.float32)
        x = rng.standard_normal((nodes, F)).astype(np.float32)
        t = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
        hi = oracle.gather_weight_scatter(si, di, w, x, acc64=True)
        mag = oracle.gather_weight_scatter(si, di, w, np.abs(x), acc64=True)
        assert close(geot.gather_weight_scatter(t(si), t(di), t(w), t(x)), hi, mag), (seed, nnz, F)
        hi = oracle.gather_scatter(si, di, x, acc64=True)
        mag = oracle.gather_scatter(si, di, np.abs(x), acc64=True)
        assert close(geot.gather_scatter(t(si), t(di), t(x)), hi, mag), (seed, nnz, F)
        H = int(rng.choice([1, 2, 3, 4, 8]))
        Fh = int(rng.choice([1, 2, 4, 6, 16, 32]))
        x3 = rng.standard_normal((nodes, H, Fh)).astype(np.float32)
        wh = rng.random((nnz, H), dtype=np.float32)
        hi = oracle.mh_spmm(si, di, wh, x3, acc64=True)
        mag = oracle.mh_spmm(si, di, wh, np.abs(x3), acc64=True)
        assert close(geot.mh_spmm(t(si), t(di), t(wh), t(x3)), hi, mag), (seed, nnz, H, Fh)
        if nnz != H:   # weight.size(0)==nnz picks the [nnz,H] layout first, as in the reference
            assert close(geot.mh_spmm(t(si), t(di), t(np.ascontiguousarray(wh.T)), t(x3)), hi, mag), (seed, nnz, H, Fh, "T")
        sd = geot.sddmm_coo_impl(t(si), t(di), t(x[: int(di[-1]) + 1 if False else nodes]), t(x))
        ref = oracle.sddmm_coo(si, di, x, x, acc64=True)
        assert np.allclose(sd.cpu().numpy(), ref, rtol=1e-4, atol=1e-4), (seed, nnz, F, "sddmm")


@pytest.mark.parametrize("seed", range(3))
def test_fuzz_source_blocked_kernel(oracle, seed):
    """The same random graphs through the source-blocked path (forced: these graphs are far too small for the density
    rule): every weight mode, the three row widths, hubs that must be split, gaps, tiny inputs."""
    import geot_amd as geot
    from geot_amd import ops
    rng = np.random.default_rng(3000 + seed)
    t = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    old = ops.set_option("slab_mode", "always")
    try:
        for _ in range(20):
            nnz = int(rng.choice([1, 64, 65, 1000, 5000, 30_000, int(rng.integers(1, 60_000))]))
            di = random_index(rng, nnz)
            nodes = int(di[-1]) + 1 + int(rng.integers(0, 50))
            si = rng.integers(0, nodes, nnz).astype(np.int64)
            H, Fh = [(1, 64), (1, 128), (1, 256), (2, 32), (4, 64), (8, 32), (4, 16), (16, 16)][int(rng.integers(0, 8))]
            x3 = rng.standard_normal((nodes, H, Fh)).astype(np.float32)
            calls = ops.stats()["slab_calls"]
            if H == 1:
                w = rng.random(nnz, dtype=np.float32)
                x = x3.reshape(nodes, Fh)
                hi = oracle.gather_weight_scatter(si, di, w, x, acc64=True)
                mag = oracle.gather_weight_scatter(si, di, w, np.abs(x), acc64=True)
                assert close(geot.gather_weight_scatter(t(si), t(di), t(w), t(x)), hi, mag), (seed, nnz, Fh, "gws")
                hi = oracle.gather_scatter(si, di, x, acc64=True)
                mag = oracle.gather_scatter(si, di, np.abs(x), acc64=True)
                assert close(geot.gather_scatter(t(si), t(di), t(x)), hi, mag), (seed, nnz, Fh, "gs")
                red = str(rng.choice(["mean", "max", "min"]))
                kind = {"max": "amax", "min": "amin"}.get(red, red)
                msg = t(x)[t(si)] * t(w)[:, None]
                ref = torch.zeros(hi.shape[0], Fh, device="cuda").scatter_reduce(0, t(di)[:, None].expand(-1, Fh), msg, kind, include_self=False)
                got = geot.gather_weight_scatter(t(si), t(di), t(w), t(x), red)
                assert got.shape == ref.shape and (torch.equal(got, ref) if red != "mean" else torch.allclose(got, ref, rtol=1e-4, atol=1e-5)), (seed, nnz, Fh, red)
                sd = geot.sddmm_coo_impl(t(si), t(di), t(x), t(x))               # SDDMM over the same plan
                ref = oracle.sddmm_coo(si, di, x, x, acc64=True)
                assert np.allclose(sd.cpu().numpy(), ref, rtol=1e-4, atol=1e-4), (seed, nnz, Fh, "sddmm")
                assert ops.stats()["slab_calls"] == calls + 4
            else:
                wh = rng.random((nnz, H), dtype=np.float32)
                hi = oracle.mh_spmm(si, di, wh, x3, acc64=True)
                mag = oracle.mh_spmm(si, di, wh, np.abs(x3), acc64=True)
                assert close(geot.mh_spmm(t(si), t(di), t(wh), t(x3)), hi, mag), (seed, nnz, H, Fh)
                if nnz != H:
                    assert close(geot.mh_spmm(t(si), t(di), t(np.ascontiguousarray(wh.T)), t(x3)), hi, mag), (seed, nnz, H, Fh, "T")
                assert ops.stats()["slab_calls"] > calls
    finally:
        ops.set_option("slab_mode", old)


def test_host_layer_streams_and_inplace_edits(oracle):
    """The plugin's caches under use a model would make of them: two streams at once (one workspace per stream),
    in-place edits of the index between calls (version counter moves: facts, sorted form and plans must not be reused),
    views of one edge_index tensor (they share their storage's facts)."""
    import geot_amd as geot
    rng = np.random.default_rng(77)
    nnz, K, F = 200_000, 5000, 64
    index = np.sort(rng.integers(0, K, nnz)).astype(np.int64)
    index[-1] = K - 1
    src = rng.standard_normal((nnz, F)).astype(np.float32)
    t_idx, t_src = torch.from_numpy(index).cuda(), torch.from_numpy(src).cuda()
    hi = oracle.index_scatter(index, src, acc64=True)
    mag = oracle.index_scatter(index, np.abs(src), acc64=True)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    torch.cuda.synchronize()
    for rep in range(6):
        for s in (s1, s2):
            with torch.cuda.stream(s):
                outs.append(geot.index_scatter(0, t_src, t_idx, "sum", True))
    torch.cuda.synchronize()
    for o in outs:
        assert close(o, hi, mag)
        assert torch.equal(o, outs[0])
    # shuffle IN PLACE (version counter moves): must be noticed, and noticed again when sorted back
    perm = torch.randperm(nnz - 1, device="cuda")
    t_idx[:-1] = t_idx[:-1][perm]
    t_src2 = t_src.clone()
    t_src2[:-1] = t_src[:-1][perm]
    assert close(geot.index_scatter(0, t_src2, t_idx, "sum", True), hi, mag, tol=2e-5)
    t_idx.copy_(torch.from_numpy(index).cuda())
    assert torch.equal(geot.index_scatter(0, t_src, t_idx, "sum", True), outs[0])
    # a [2, nnz] edge_index: its rows are views of one storage at different offsets
    edge_index = torch.stack([torch.from_numpy(rng.integers(0, K, nnz)).cuda(), t_idx])
    x = torch.rand(K, F, device="cuda")
    a = geot.gather_scatter(edge_index[0], edge_index[1], x)
    b = geot.gather_scatter(edge_index[0].clone(), edge_index[1].clone(), x)
    assert torch.equal(a, b)


def test_host_layer_from_several_threads(oracle):
    """The dispatcher releases the GIL inside the C++ ops: four Python threads, each on its own stream, call the
    operators concurrently (shared caches behind a mutex, per-thread read-back slots and workspaces)."""
    import threading
    import geot_amd as geot
    rng = np.random.default_rng(99)
    nnz, K, F = 150_000, 4000, 64
    index = np.sort(rng.integers(0, K, nnz)).astype(np.int64)
    index[-1] = K - 1
    shuffled = index.copy()
    shuffled[:-1] = rng.permutation(shuffled[:-1])
    src = rng.standard_normal((nnz, F)).astype(np.float32)
    si = rng.integers(0, K, nnz).astype(np.int64)
    w = rng.random(nnz, dtype=np.float32)
    x = rng.standard_normal((K, F)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    t_index, t_shuf, t_src, t_si, t_w, t_x = t(index), t(shuffled), t(src), t(si), t(w), t(x)
    want_is = geot.index_scatter(0, t_src, t_index)
    want_un = geot.index_scatter(0, t_src, t_shuf, "sum", False)
    want_gws = geot.gather_weight_scatter(t_si, t_index, t_w, t_x)
    torch.cuda.synchronize()
    errors = []

    def worker(seed):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for it in range(40):
                    kind = (seed + it) % 3
                    if kind == 0:
                        ok = torch.equal(geot.index_scatter(0, t_src, t_index), want_is)
                    elif kind == 1:
                        ok = torch.equal(geot.index_scatter(0, t_src, t_shuf, "sum", False), want_un)
                    else:
                        ok = torch.equal(geot.gather_weight_scatter(t_si, t_index, t_w, t_x), want_gws)
                    if not ok:
                        errors.append((seed, it, kind))
                s.synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append((seed, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]
