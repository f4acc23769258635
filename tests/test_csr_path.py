"""CSR path: csr_gws and coo_to_csr.  CPU part: the oracle against the
comparators of test/test_csr_gws.py; GPU part: the HIP path against the oracle."""
import numpy as np
import pytest
import torch

from conftest import powerlaw_index


def make_csr(rng, nrow, nnz, F, empty_tail=0):
    dst = np.sort(rng.integers(0, nrow - empty_tail, nnz)).astype(np.int64)
    col = rng.integers(0, nrow, nnz).astype(np.int64)
    w = rng.random(nnz, dtype=np.float32)
    src = rng.random((nrow, F), dtype=np.float32)
    # test/test_csr_gws.py:6-12: rowptr = [0, cumsum(bincount(row, minlength=nrow))]
    rowptr = np.zeros(nrow + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(np.bincount(dst, minlength=nrow))
    return dst, col, w, src, rowptr


def test_oracle_csr_gws_matches_reference_test_comparator(oracle):
    rng = np.random.default_rng(0)
    dst, col, w, src, rowptr = make_csr(rng, 100, 1000, 32)            # the reference test's shape
    out = oracle.csr_gws(rowptr, col, w, src)
    assert out.shape == (101, 32) and np.all(out[100] == 0)             # nrow+1 rows (csrc/csr_gws.cpp:29-31)
    adj = torch.sparse_coo_tensor(torch.stack([torch.from_numpy(dst), torch.from_numpy(col)]),
                                  torch.from_numpy(w), (100, 100)).coalesce()
    ref = torch.sparse.mm(adj, torch.from_numpy(src)).numpy()           # test/test_csr_gws.py:16-25
    np.testing.assert_allclose(out[:100], ref, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(out[:100], oracle.gather_weight_scatter(col, dst, w, src, rows=100), rtol=1e-6)
    np.testing.assert_array_equal(oracle.coo_to_csr(dst, 100), rowptr.astype(np.int32))
    perm = rng.permutation(1000)
    np.testing.assert_array_equal(oracle.coo_to_csr(dst[perm], 100), rowptr.astype(np.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("nrow,nnz,F,empty_tail", [(100, 1000, 32, 0), (5000, 100_000, 128, 0), (3000, 50_000, 7, 700),
                                                   (50, 20_000, 64, 0), (20_000, 3_000, 16, 5000)])
def test_csr_gws_hip(oracle, nrow, nnz, F, empty_tail):
    import geot_amd as geot
    rng = np.random.default_rng(nrow + F)
    dst, col, w, src, rowptr = make_csr(rng, nrow, nnz, F, empty_tail)
    t = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    out = geot.csr_gws(t(rowptr), t(col), t(w), t(src))
    assert out.shape == (nrow + 1, F)
    hi = oracle.csr_gws(rowptr, col, w, src, acc64=True)
    got = out.cpu().numpy()
    assert np.all(np.abs(got - hi) <= 1e-5 * np.abs(hi) + 1e-30) and np.all(got[hi == 0] == 0)
    out32 = geot.csr_gws(t(rowptr).int(), t(col).int(), t(w), t(src))    # int32 CSR like the reference's
    assert torch.equal(out32, out)
    ref = geot.gather_weight_scatter(t(col), t(dst), t(w), t(src))
    assert torch.allclose(out[: ref.shape[0]], ref, rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_csr_gws_dense_graph_takes_the_source_blocked_path(oracle):
    """A CSR call on a graph the density rule accepts (forced here) goes through the COO path: row ids expanded once per
    indptr content, then the source-blocked kernel; same rows (indptr.size(0), the last one zero) and values."""
    import geot_amd as geot
    from geot_amd import ops
    rng = np.random.default_rng(11)
    nrow, nnz, F = 4000, 300_000, 64
    dst, col, w, src, rowptr = make_csr(rng, nrow, nnz, F, empty_tail=300)
    t = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    args = (t(rowptr), t(col), t(w), t(src))
    base = geot.csr_gws(*args)
    old = ops.set_option("slab_mode", "always")
    try:
        calls = ops.stats()["slab_calls"]
        out = geot.csr_gws(*args)
        out32 = geot.csr_gws(args[0].int(), args[1].int(), args[2], args[3])
        assert ops.stats()["slab_calls"] == calls + 2
    finally:
        ops.set_option("slab_mode", old)
    assert out.shape == (nrow + 1, F) and out[nrow - 300:].abs().sum().item() == 0
    hi = oracle.csr_gws(rowptr, col, w, src, acc64=True)
    got = out.cpu().numpy()
    assert np.all(np.abs(got - hi) <= 1e-5 * np.abs(hi) + 1e-30) and np.all(got[hi == 0] == 0)
    assert torch.allclose(out, base, rtol=1e-5, atol=1e-6) and torch.equal(out32, out)


@pytest.mark.gpu
def test_coo_to_csr_hip(oracle):
    import geot_amd as geot
    from geot_amd import hip
    for nnz, nrow, seed in ((1000, 100, 1), (300_000, 20_000, 2), (5000, 40_000, 3)):
        row = powerlaw_index(nnz, nrow, seed)
        expect = oracle.coo_to_csr(row, nrow)
        got = geot.coo_to_csr(torch.from_numpy(row).cuda())
        assert got.dtype == torch.int32 and got.shape == (nrow + 1,)
        np.testing.assert_array_equal(got.cpu().numpy(), expect)
        perm = np.random.default_rng(seed).permutation(nnz)
        np.testing.assert_array_equal(geot.coo_to_csr(torch.from_numpy(row[perm]).cuda()).cpu().numpy(), expect)
        fast = hip.coo_to_csr_out(torch.from_numpy(row).cuda(), torch.empty(nrow + 1, dtype=torch.int32, device="cuda"),
                                  assume_sorted=True)
        np.testing.assert_array_equal(fast.cpu().numpy(), expect)
