"""Content guard of the remembered products (`-m gpu`; csrc/seg_guard.hip, "content guard" in csrc/host_cache.cpp).

The reference keeps nothing between calls: every call reads the caller's tensors (csrc/gather_scatter.cpp:25-34,
geot/gather_scatter.py:30-33 re-sorts on every backward call), so its result always follows the bytes it is handed.  The host
side here remembers what it derived from index tensors - the slab plan, the static weight in plan order, the edge list sorted
by source, the stable sort of an index with descents, widened int32 indices, expanded CSR row ids - under the tensors'
identity and version counter.  These tests write NEW content into such tensors behind the version counter (`.data`), and
require of every operator: the result of the call that follows is the oracle's result for the bytes the tensors hold at that
moment, bit-for-bit what a process that never saw the old content returns.
"""
import ctypes
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.fixture(scope="module")
def geot():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import geot_amd
    return geot_amd


@pytest.fixture()
def ops(geot):
    from geot_amd import ops
    ops.clear_caches()
    saved = {k: ops.get_option(k) for k in ("slab_mode", "content_guard", "trust_version")}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        yield ops
    for k, v in saved.items():
        ops.set_option(k, v)
    ops.clear_caches()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def behind_the_version_counter(t, new):
    v = t._version
    t.data.copy_(dev(new) if isinstance(new, np.ndarray) else new)
    assert t._version == v


def close(got, hi, what):
    got = got.detach().cpu().numpy().astype(np.float64)
    assert got.shape == hi.shape, (what, got.shape, hi.shape)
    bound = RTOL * np.abs(hi).max() + 1e-30
    assert np.abs(got - hi).max() <= bound, f"{what}: {np.abs(got - hi).max() / bound:.3g} x bound"


def graph(rng, nnz, K):
    di = np.sort(rng.integers(0, K, nnz)).astype(np.int64)
    di[-1] = K - 1
    si = rng.integers(0, K, nnz).astype(np.int64)
    return si, di


# ---- the kernel through the C ABI --------------------------------------------------------------------------------------------
def fingerprint(_lib, bufs, fp, compare, verdict, seq, scratch):
    n = len(bufs)
    ptrs = (ctypes.c_void_p * n)(*[b.data_ptr() for b in bufs])
    sizes = (ctypes.c_size_t * n)(*[b.numel() * b.element_size() for b in bufs])
    L = _lib.load()
    rc = L.geot_content_fingerprint(ptrs, sizes, n, fp.data_ptr(), compare, verdict.data_ptr() if verdict is not None else None, seq,
                                    scratch.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, L.geot_last_error()


def test_fingerprint_kernel_sees_every_word_and_its_position(geot):
    from geot_amd import _lib
    rng = np.random.default_rng(5)
    scratch = torch.zeros(_lib.load().geot_content_fingerprint_scratch_bytes() // 8, dtype=torch.int64, device="cuda")
    verdict = torch.zeros(2, dtype=torch.int64).pin_memory()
    seq = 0

    def same(bufs, fp):
        nonlocal seq
        seq += 1
        fingerprint(_lib, bufs, fp, 1, verdict, seq, scratch)
        torch.cuda.synchronize()
        assert int(verdict[1]) == seq and int(verdict[0]) in (1, 2)
        assert int(scratch[0]) == 0                                            # the ticket is left zero for the next launch
        return int(verdict[0]) == 1

    for n, dtype in ((1, np.int64), (2, np.int64), (7, np.int64), (4097, np.int64), (3_000_001, np.int64), (1_000_003, np.int32),
                     (999_999, np.float16), (5, np.int32), (1, np.int16)):
        a = dev(rng.integers(1, 1000, n).astype(dtype))
        b = dev(rng.integers(1, 2 ** 15, max(n // 3, 1)).astype(np.int64))
        fp = torch.zeros(2, dtype=torch.int64, device="cuda")
        fingerprint(_lib, [a, b], fp, 0, None, 0, scratch)
        assert same([a, b], fp), (n, dtype)
        assert same([a.clone(), b.clone()], fp), "content, not address"
        for at in sorted({0, n // 2, n - 1}):                                  # one word changed, anywhere
            keep = a[at].clone()
            a[at] = keep + 1
            assert not same([a, b], fp), (n, dtype, at)
            a[at] = keep
        assert same([a, b], fp)
        if n >= 2:                                                             # two words exchanged: same multiset, other content
            i, j = 0, n - 1
            if a[i] != a[j]:
                a[[i, j]] = a[[j, i]]
                assert not same([a, b], fp), ("swap", n, dtype)
                a[[i, j]] = a[[j, i]]
        assert not same([a, b[:-1]] if b.numel() > 1 else [a], fp), "another length"
        if n > 4:                                                              # an unaligned view (element-aligned only) of the same values
            shifted = torch.empty(n + 1, dtype=a.dtype, device="cuda")
            shifted[1:] = a
            fp2 = torch.zeros(2, dtype=torch.int64, device="cuda")
            fingerprint(_lib, [shifted[1:], b], fp2, 0, None, 0, scratch)
            assert same([shifted[1:], b], fp2)
            shifted[n // 2] += 1
            assert not same([shifted[1:], b], fp2)


# ---- every remembered product, one by one -----------------------------------------------------------------------------------
def test_a_new_edge_list_in_the_same_tensors_forward_and_plan(geot, oracle, ops):
    rng = np.random.default_rng(11)
    nnz, K, F = 600_000, 3_000, 64
    si, di = graph(rng, nnz, K)
    si2, di2 = graph(rng, nnz, K)
    w = rng.random(nnz, dtype=np.float32)
    x = rng.random((K, F), dtype=np.float32)
    t_si, t_di, t_w, t_x = dev(si), dev(di), dev(w), dev(x)
    ops.set_option("slab_mode", "always")                                      # (a plan from the first call on)
    want1 = oracle.gather_weight_scatter(si, di, w, x, rows=K, acc64=True)
    for _ in range(3):                                                          # third call: the static weight sits in plan order too
        close(geot.gather_weight_scatter(t_si, t_di, t_w, t_x), want1, "first content")
    st = ops.stats()
    assert st["plans"] == 1 and st["guard_checks"] >= 2
    behind_the_version_counter(t_si, si2)
    behind_the_version_counter(t_di, di2)
    stale0 = st["stale_products"]
    want2 = oracle.gather_weight_scatter(si2, di2, w, x, rows=K, acc64=True)
    got = geot.gather_weight_scatter(t_si, t_di, t_w, t_x)
    close(got, want2, "new edge list behind the version counter")
    st = ops.stats()
    assert st["stale_products"] == stale0 + 1
    for _ in range(3):                                                          # ... and a plan of the new content serves from here on
        close(geot.gather_weight_scatter(t_si, t_di, t_w, t_x), want2, "new content again")
    assert ops.stats()["stale_products"] == stale0 + 1 and ops.stats()["plans"] == 1
    # only the source side changes (the destination index, its facts and the row count stay as they were)
    si3 = rng.integers(0, K, nnz).astype(np.int64)
    behind_the_version_counter(t_si, si3)
    close(geot.gather_scatter(t_si, t_di, t_x), oracle.gather_scatter(si3, di2, x, rows=K, acc64=True), "sources only")
    # the static weight in plan order: new values behind the version counter
    for _ in range(3):
        geot.gather_weight_scatter(t_si, t_di, t_w, t_x)
    w2 = rng.random(nnz, dtype=np.float32)
    behind_the_version_counter(t_w, w2)
    close(geot.gather_weight_scatter(t_si, t_di, t_w, t_x), oracle.gather_weight_scatter(si3, di2, w2, x, rows=K, acc64=True), "static weight")
    # multi-head over the same plan machinery
    H = 4
    wh = rng.random((nnz, H), dtype=np.float32)
    x3 = rng.random((K, H, 16), dtype=np.float32)
    t_wh, t_x3 = dev(wh), dev(x3)
    for _ in range(2):
        geot.mh_spmm(t_si, t_di, t_wh, t_x3)
    si4 = rng.integers(0, K, nnz).astype(np.int64)
    behind_the_version_counter(t_si, si4)
    close(geot.mh_spmm(t_si, t_di, t_wh, t_x3).reshape(K, -1), oracle.mh_spmm(si4, di2, wh, x3, rows=K, acc64=True).reshape(K, -1), "mh_spmm")
    # the static MULTI-HEAD weight in plan order (round 5: weight mode 5 - kept like the single weight is): the third call reads the
    # plan-order copy; new values behind the version counter must still win
    for _ in range(3):
        got = geot.mh_spmm(t_si, t_di, t_wh, t_x3)
    close(got.reshape(K, -1), oracle.mh_spmm(si4, di2, wh, x3, rows=K, acc64=True).reshape(K, -1), "mh_spmm, static weight in plan order")
    wh2 = rng.random((nnz, H), dtype=np.float32)
    stale1 = ops.stats()["stale_products"]
    behind_the_version_counter(t_wh, wh2)
    close(geot.mh_spmm(t_si, t_di, t_wh, t_x3).reshape(K, -1), oracle.mh_spmm(si4, di2, wh2, x3, rows=K, acc64=True).reshape(K, -1),
          "mh_spmm, new weight values behind the version counter")
    assert ops.stats()["stale_products"] == stale1 + 1


def test_a_new_edge_list_in_the_same_tensors_backward(geot, oracle, ops):
    rng = np.random.default_rng(12)
    nnz, K, F = 200_000, 5_000, 32
    si, di = graph(rng, nnz, K)
    si[:K] = np.arange(K)                                                       # every node has an out-edge (the reference's grad row rule)
    si2 = rng.integers(0, K, nnz).astype(np.int64)
    si2[:K] = np.arange(K)[::-1]
    w = rng.random(nnz, dtype=np.float32)
    x = rng.random((K, F), dtype=np.float32)
    g = rng.random((K, F), dtype=np.float32)
    t_si, t_di, t_w, t_g = dev(si), dev(di), dev(w), dev(g)

    def grads():
        t_x = dev(x).requires_grad_(True)
        geot.gather_weight_scatter(t_si, t_di, t_w, t_x).backward(t_g)
        return t_x.grad

    def want(si_now):                                                           # d/dsrc = the transposed product
        order = np.argsort(si_now, kind="stable")
        return oracle.gather_weight_scatter(di[order], si_now[order], w[order], g, rows=K, acc64=True)

    for _ in range(2):
        close(grads(), want(si), "first content")
    assert ops.stats()["transposed"] == 1
    stale0 = ops.stats()["stale_products"]
    behind_the_version_counter(t_si, si2)
    close(grads(), want(si2), "new sources behind the version counter")
    assert ops.stats()["stale_products"] >= stale0 + 1
    close(grads(), want(si2), "and again")
    w2 = rng.random(nnz, dtype=np.float32)                                      # the static weight in transposed order
    behind_the_version_counter(t_w, w2)
    w = w2
    close(grads(), want(si2), "static weight")


def test_int32_indices_and_row_pointers_behind_the_version_counter(geot, oracle, ops):
    rng = np.random.default_rng(13)
    nnz, K, F = 150_000, 2_000, 32
    si, di = graph(rng, nnz, K)
    si2, di2 = graph(rng, nnz, K)
    a = rng.random((K, F), dtype=np.float32)
    b = rng.random((K, F), dtype=np.float32)
    t_si32, t_di32, t_a, t_b = dev(si.astype(np.int32)), dev(di.astype(np.int32)), dev(a), dev(b)
    sddmm = torch.ops.geot.sddmm_coo_impl
    for _ in range(2):
        close(sddmm(t_si32, t_di32, t_a, t_b), oracle.sddmm_coo(si, di, a, b, acc64=True), "sddmm, int32 indices (the reference's wrapper casts)")
    behind_the_version_counter(t_si32, si2.astype(np.int32))
    behind_the_version_counter(t_di32, di2.astype(np.int32))
    close(sddmm(t_si32, t_di32, t_a, t_b), oracle.sddmm_coo(si2, di2, a, b, acc64=True), "sddmm after the write")
    # CSR: row pointers expanded once per content (the source-blocked path of csr_gws)
    w = rng.random(nnz, dtype=np.float32)

    def rowptr(d):
        return np.concatenate([[0], np.cumsum(np.bincount(d, minlength=K))]).astype(np.int32)

    t_ptr, t_ind, t_w = dev(rowptr(di)), dev(si.astype(np.int32)), dev(w)
    ops.set_option("slab_mode", "always")
    for _ in range(2):
        close(geot.csr_gws(t_ptr, t_ind, t_w, t_a), oracle.csr_gws(rowptr(di).astype(np.int64), si, w, a, out_rows=K + 1, acc64=True), "csr_gws")
    behind_the_version_counter(t_ptr, rowptr(di2))
    behind_the_version_counter(t_ind, si2.astype(np.int32))
    close(geot.csr_gws(t_ptr, t_ind, t_w, t_a), oracle.csr_gws(rowptr(di2).astype(np.int64), si2, w, a, out_rows=K + 1, acc64=True), "csr_gws after the write")


def test_the_remembered_sort_of_an_unsorted_index(geot, oracle, ops):
    rng = np.random.default_rng(14)
    nnz, K, F = 300_000, 4_000, 16
    index = rng.integers(0, K, nnz).astype(np.int64)
    index[-1] = K - 1
    index2 = rng.integers(0, K, nnz).astype(np.int64)
    index2[-1] = K - 1
    src = rng.random((nnz, F), dtype=np.float32)
    t_index, t_src = dev(index), dev(src)

    def want(ix):
        order = np.argsort(ix, kind="stable")
        return oracle.index_scatter(ix[order], src[order], rows=K, acc64=True)

    for _ in range(2):
        close(geot.index_scatter(0, t_src, t_index, "sum", False), want(index), "unsorted index")
    assert ops.stats()["sorts"] >= 1
    sorts = ops.stats()["sorts"]
    geot.index_scatter(0, t_src, t_index, "sum", False)
    assert ops.stats()["sorts"] == sorts                                        # (the sort is remembered)
    behind_the_version_counter(t_index, index2)
    close(geot.index_scatter(0, t_src, t_index, "sum", False), want(index2), "another unsorted content in the same tensor")
    close(geot.index_scatter(0, t_src, t_index, "sum", False), want(index2), "and again")


def test_guard_costs_nothing_where_it_is_off_and_little_where_it_is_on(geot, ops):
    rng = np.random.default_rng(15)
    nnz, K, F = 400_000, 3_000, 64
    si, di = graph(rng, nnz, K)
    t_si, t_di, t_x = dev(si), dev(di), dev(rng.random((K, F), dtype=np.float32))
    ops.set_option("slab_mode", "always")
    ref = geot.gather_scatter(t_si, t_di, t_x)
    for name, value in (("content_guard", 0), ("trust_version", 2)):
        old = ops.set_option(name, value)
        checks = ops.stats()["guard_checks"]
        for _ in range(3):
            assert torch.equal(geot.gather_scatter(t_si, t_di, t_x), ref)
        assert ops.stats()["guard_checks"] == checks, name
        ops.set_option(name, old)
    checks, stale = ops.stats()["guard_checks"], ops.stats()["stale_products"]
    for _ in range(3):
        assert torch.equal(geot.gather_scatter(t_si, t_di, t_x), ref)
    assert ops.stats()["guard_checks"] == checks + 3 and ops.stats()["stale_products"] == stale
    # a captured graph replays without fingerprints (static content is the contract of a capture)
    g = torch.cuda.CUDAGraph()
    out = torch.empty_like(ref)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        geot.gather_scatter(t_si, t_di, t_x)
        torch.cuda.synchronize()
        checks = ops.stats()["guard_checks"]
        with torch.cuda.graph(g, stream=s):
            out.copy_(geot.gather_scatter(t_si, t_di, t_x))
    assert ops.stats()["guard_checks"] == checks
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


def test_two_threads_one_of_them_rewriting_its_edge_list(geot, ops):
    """Operator calls release the GIL: one thread's stale product drops every cache while the other thread is in the middle of
    using its own plan.  Both must keep getting the result of the bytes they hold."""
    import threading
    from geot_amd import hip
    rng = np.random.default_rng(16)
    nnz, K, F = 500_000, 2_500, 64
    ops.set_option("slab_mode", "always")
    errors = []

    def expected(t_si, t_di, t_x):
        out = torch.empty(K, F, device="cuda")
        hip.gather_scatter_out(t_si, t_di, t_x, out)                           # per-edge kernels through the C ABI: no host caches
        return out

    def worker(seed, rewrite):
        try:
            r = np.random.default_rng(seed)
            si, di = graph(r, nnz, K)
            t_si, t_di, t_x = dev(si), dev(di), dev(r.random((K, F), dtype=np.float32))
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                want = expected(t_si, t_di, t_x)
                for it in range(40):
                    if rewrite and it % 3 == 2:
                        behind_the_version_counter(t_si, r.integers(0, K, nnz).astype(np.int64))
                        want = expected(t_si, t_di, t_x)
                    got = geot.gather_scatter(t_si, t_di, t_x)
                    err = float(((got - want).abs().max() / want.abs().max()).item())
                    if not err < 1e-5:
                        errors.append((seed, it, err))
                        return
        except Exception as e:  # noqa: BLE001
            errors.append((seed, repr(e)))

    threads = [threading.Thread(target=worker, args=(100, False)), threading.Thread(target=worker, args=(101, True)),
               threading.Thread(target=worker, args=(102, True))]
    stale0 = ops.stats()["stale_products"]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    # (26 rewrites; one thread's stale product drops every cache, so the other's next rewrite may meet no remembered product at all)
    assert ops.stats()["stale_products"] >= stale0 + 3
