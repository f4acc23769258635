"""The reference's `geot.triton.launch_*` surface (geot/triton/__init__.py:1-5) on the HIP engine:
names / argument order on CPU, accumulate-into-output parity against the oracle on the GPU."""
import inspect

import numpy as np
import pytest
import torch

from oracle import api as oracle

# argument names as the reference spells them (geot/triton/seg_reduction.py:76,81; spmm.py:78,83;
# torch_compile.py:23)
SIGNATURES = {
    "launch_parallel_reduction": ["indices", "input", "output", "num_edges", "feature_size", "BLOCK_SIZE"],
    "launch_serial_reduction": ["edges", "input", "output", "num_edges", "feature_size", "group_size"],
    "launch_pr_spmm": ["indices", "input", "output", "num_edges", "feature_size", "BLOCK_SIZE"],
    "launch_sr_spmm": ["edges", "input", "output", "num_edges", "feature_size", "group_size"],
    "launch_torch_compile_spmm": ["in0", "in1", "out", "num_edges", "feature_size", "XBLOCK"],
}


def test_launcher_surface_matches_reference():
    import geot.triton as gt
    from geot_amd import comparators
    assert sorted(gt.__all__) == sorted(SIGNATURES)
    for name, args in SIGNATURES.items():
        fn = getattr(gt, name)
        assert fn is getattr(comparators, name)
        assert list(inspect.signature(fn).parameters) == args


def test_launchers_refuse_cpu_tensors():
    import geot.triton as gt
    idx = torch.zeros(4, dtype=torch.int64)
    with pytest.raises((RuntimeError, ValueError, TypeError)):
        gt.launch_serial_reduction(idx, torch.ones(4, 2), torch.zeros(1, 2), 4, 2, 32)


@pytest.mark.gpu
@pytest.mark.parametrize("F", [1, 8, 32, 64])
def test_reduction_launchers_accumulate(F):
    import geot.triton as gt
    rng = np.random.default_rng(F)
    nnz, K = 5000, 300
    index = np.sort(rng.integers(0, K, nnz)).astype(np.int64)
    index[-1] = K - 1
    src = rng.random((nnz, F), dtype=np.float32)
    base = rng.random((K, F), dtype=np.float32)
    want = base.astype(np.float64) + oracle.index_scatter(index, src, acc64=True)
    dev = torch.device("cuda:0")
    for fn in (gt.launch_parallel_reduction, gt.launch_serial_reduction):
        out = torch.from_numpy(base).to(dev)
        fn(torch.from_numpy(index).to(dev), torch.from_numpy(src).to(dev), out, nnz, F, 32)
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("F", [4, 32, 100])
def test_spmm_launchers_accumulate(F):
    import geot.triton as gt
    rng = np.random.default_rng(100 + F)
    nnz, N = 4000, 250
    dst = np.sort(rng.integers(0, N, nnz)).astype(np.int64)
    dst[-1] = N - 1
    si = rng.integers(0, N, nnz).astype(np.int64)
    x = rng.random((N, F), dtype=np.float32)
    base = rng.random((N, F), dtype=np.float32)
    want = base.astype(np.float64) + oracle.gather_scatter(si, dst, x, acc64=True)
    dev = torch.device("cuda:0")
    edges = torch.from_numpy(np.stack([si, dst])).to(dev)
    for fn in (gt.launch_pr_spmm, gt.launch_sr_spmm, gt.launch_torch_compile_spmm):
        out = torch.from_numpy(base).to(dev)
        fn(edges, torch.from_numpy(x).to(dev), out, nnz, F, 32)
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    # the order-agnostic launcher on shuffled edges
    perm = rng.permutation(nnz)
    out = torch.from_numpy(base).to(dev)
    gt.launch_torch_compile_spmm(torch.from_numpy(np.stack([si[perm], dst[perm]])).to(dev),
                                 torch.from_numpy(x).to(dev), out, nnz, F, 1024)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
