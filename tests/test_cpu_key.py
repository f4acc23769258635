"""The CPU dispatch key of geot::index_scatter (`geot_amd/csrc/torch_ops.cpp` index_scatter_cpu_op).

The reference registers a CPU kernel for index_scatter and for nothing else (csrc/index_scatter.cpp:11-24,53 ->
csrc/cpu/index_scatter_cpu.cpp).  Ours is pinned to it bit-for-bit through the operand identity
ref(index, src) == ours(index, src[index]) - against the committed golden outputs of the compiled reference and, when
oracle/_ref is built, live; and to the oracle's restatement on the intended operand.  It serves CPU tensors only:
GPU tensors never reach it, and the package still does not import without libgeot_hip.so.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, sorted_index

import geot_amd

IS_CASES = load_golden("index_scatter.npz")
SMALL = sorted(c for c in IS_CASES if "index" in IS_CASES[c] and c not in ("f16", "bf16_bits"))


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize("case", SMALL)
def test_cpu_key_matches_captured_reference_output(oracle, case):
    g = IS_CASES[case]
    index, src = g["index"], g["src"]
    out = geot_amd.index_scatter(0, t(src), t(index), "sum", sorted=True).numpy()
    np.testing.assert_array_equal(out, oracle.index_scatter(index, src))           # intended operand, edge order
    if "ref_out" in g:                                                               # the compiled reference, as shipped
        via_identity = geot_amd.index_scatter(0, t(src[index]), t(index), "sum", sorted=True).numpy()
        np.testing.assert_array_equal(via_identity, g["ref_out"])


def test_cpu_key_reductions_and_nan_pinned_to_reference(oracle):
    g = load_golden("reductions.npz")["reductions"]
    index, src = g["index"], g["src"]
    for red in ("sum", "mean", "min", "max", "prod"):
        ours = geot_amd.index_scatter(0, t(src[index]), t(index), red, sorted=True).numpy()
        np.testing.assert_array_equal(ours, g[f"ref_{red}"], err_msg=red)
        assert np.all(ours[17] == 0)                                                # empty key stays 0 for every reduce
    assert torch.equal(geot_amd.index_scatter(0, t(src), t(index), "amax"), geot_amd.index_scatter(0, t(src), t(index), "max"))
    g = load_golden("reductions.npz")["reductions_nan"]
    for red in ("min", "max", "sum"):
        ours = geot_amd.index_scatter(0, t(g["src"][g["index"]]), t(g["index"]), red).numpy()
        np.testing.assert_array_equal(ours, g[f"ref_{red}"], err_msg="nan " + red)


def test_cpu_key_16bit_storage_pinned_to_reference():
    g = IS_CASES["f16"]
    ours = geot_amd.index_scatter(0, t(g["src"][g["index"]]), t(g["index"])).numpy()
    np.testing.assert_array_equal(ours.view(np.uint16), g["ref_out"].view(np.uint16))
    g = IS_CASES["bf16_bits"]
    src = t(g["src"].view(np.int16)).view(torch.bfloat16)
    ours = geot_amd.index_scatter(0, src[t(g["index"])], t(g["index"]))
    np.testing.assert_array_equal(ours.view(torch.int16).numpy().view(np.uint16), g["ref_out"])


@pytest.mark.parametrize("nnz,keys,F", [(1, 1, 1), (257, 3, 5), (4096, 4096, 16), (200_000, 777, 33), (300_000, 3, 8)])
def test_cpu_key_threads_and_hubs(oracle, nnz, keys, F):
    """Rows much longer than a thread's edge range (a hub), one row per edge, odd sizes: every row is reduced by one
    thread in edge order - the result does not depend on the thread count."""
    rng = np.random.default_rng(nnz + F)
    index = sorted_index(rng, nnz, min(keys, nnz))
    src = rng.standard_normal((nnz, F)).astype(np.float32)
    want = oracle.index_scatter(index, src)
    before = torch.get_num_threads()
    try:
        for threads in (1, 4):
            torch.set_num_threads(threads)
            np.testing.assert_array_equal(geot_amd.index_scatter(0, t(src), t(index)).numpy(), want)
        for red in ("mean", "max", "prod"):
            got = geot_amd.index_scatter(0, t(src), t(index), red).numpy()
            np.testing.assert_array_equal(got, oracle.index_scatter_3pass(index, src, reduce=red), err_msg=red)
    finally:
        torch.set_num_threads(before)
    from oracle import ref
    if ref.available():
        np.testing.assert_array_equal(ref.index_scatter_cpu(index, src), geot_amd.index_scatter(0, t(src[index]), t(index)).numpy())


def test_cpu_key_contract():
    src = torch.rand(6, 4)
    idx = torch.tensor([0, 0, 1, 1, 3, 3])
    out = geot_amd.index_scatter(0, src, idx)
    assert out.shape == (4, 4) and out.dtype == src.dtype and out.device.type == "cpu" and torch.all(out[2] == 0)
    assert torch.allclose(out, torch.zeros(4, 4).index_add_(0, idx, src))
    # dim != 0 is honoured; float64
    s3 = torch.rand(3, 6, 2, dtype=torch.float64)
    assert torch.allclose(geot_amd.index_scatter(1, s3, idx), torch.zeros(3, 4, 2, dtype=torch.float64).index_add_(1, idx, s3))
    # an index with descents (the reference: "unsorted index is not supported yet") is reduced over its stable sort;
    # rows stay index[-1] + 1 and keys beyond are ignored
    shuf = torch.tensor([3, 0, 1, 7, 0, 3])
    want = torch.zeros(4, 4).index_add_(0, shuf[shuf < 4], src[shuf < 4])
    assert torch.allclose(geot_amd.index_scatter(0, src, shuf, "sum", sorted=False), want)
    assert torch.allclose(geot_amd.index_scatter(0, src, shuf, "sum", sorted=True), want)      # a wrong promise is survived
    # autograd: d/dsrc[e] = grad[index[e]]
    s = src.clone().requires_grad_(True)
    geot_amd.index_scatter(0, s, idx).backward(torch.arange(16.0).view(4, 4))
    assert torch.equal(s.grad, torch.arange(16.0).view(4, 4)[idx])
    with pytest.raises(RuntimeError, match="reduce argument must be either sum, prod, mean, amax or amin, got nope"):
        geot_amd.index_scatter(0, src, idx, "nope")
    with pytest.raises(RuntimeError, match="not implemented for 'Long'"):
        geot_amd.index_scatter(0, (src * 9).long(), idx)
    with pytest.raises(RuntimeError, match="expected scalar type Long but found Int"):
        geot_amd.index_scatter(0, src, idx.int())
