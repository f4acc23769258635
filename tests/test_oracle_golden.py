"""The oracle (our C restatement) against the golden vectors and the compiled reference.

Pins: (1) int64 bookkeeping and fp32 accumulation order of index_scatter bit-for-bit against the
REFERENCE's own CPU kernel (captured `ref_out` arrays, plus a live comparison when oracle/_ref is
present) through the operand identity ref(index, src) == oracle(index, src[index]);
(2) all five reductions of the reference CPU path; (3) the gather ops against the torch comparators
the reference's own tests use.
"""
import hashlib

import numpy as np
import pytest

from conftest import load_golden, sorted_index

IS_CASES = load_golden("index_scatter.npz")
SMALL = sorted(c for c in IS_CASES if "index" in IS_CASES[c] and c not in ("f16", "bf16_bits"))


def bf16_bits_to_f32(bits):
    return (bits.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_bits(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)


def test_16bit_storage_semantics_pinned_to_reference(oracle):
    """half / bfloat16: fp32 accumulation in edge order, ONE rounding at the end - bit-for-bit the compiled
    reference (csrc/cpu/index_scatter_cpu.cpp:78-86,114-116), through the src[index] identity."""
    g = IS_CASES["f16"]
    up = g["src"].astype(np.float32)
    ours = oracle.index_scatter(g["index"], up[g["index"]]).astype(np.float16)
    np.testing.assert_array_equal(ours.view(np.uint16), g["ref_out"].view(np.uint16))
    g = IS_CASES["bf16_bits"]
    up = bf16_bits_to_f32(g["src"])
    ours = f32_to_bf16_bits(oracle.index_scatter(g["index"], up[g["index"]]))
    np.testing.assert_array_equal(ours, g["ref_out"])


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("case", SMALL)
def test_index_scatter_matches_reference_test_comparators(oracle, case):
    g = IS_CASES[case]
    out = oracle.index_scatter(g["index"], g["src"])
    assert out.shape[0] == int(g["rows"]) == int(g["index"][-1]) + 1          # row rule
    # sequential fp32 order == torch CPU index_add_ (bit-exact), scatter_add_ within fp32 rounding
    np.testing.assert_array_equal(out, g["torch_index_add"])
    if "torch_scatter_add" in g:
        np.testing.assert_allclose(out, g["torch_scatter_add"], rtol=1e-5, atol=1e-4)  # test/test_index_scatter.py:19
    # fp32 result within 1e-5 relative of the float64-accumulated one (positive or mixed-sign data)
    if g["src"].dtype == np.float32:
        hi = oracle.index_scatter(g["index"], g["src"], acc64=True)
        mag = oracle.index_scatter(g["index"], np.abs(g["src"]), acc64=True)
        assert np.all(np.abs(out - hi) <= 1e-5 * mag + 1e-30)


@pytest.mark.parametrize("case", [c for c in SMALL if "ref_out" in IS_CASES[c]])
def test_identity_against_captured_reference_output(oracle, case):
    """ref(index, src) == oracle(index, src[index]) bit-for-bit (the shipped kernel reads src[index[n]])."""
    g = IS_CASES[case]
    np.testing.assert_array_equal(oracle.index_scatter(g["index"], g["src"][g["index"]]), g["ref_out"])
    if g["src"].dtype == np.float32 and g["src"].ndim == 2:
        np.testing.assert_array_equal(oracle.index_scatter(g["index"], g["src"], refquirk=True), g["ref_out"])


def test_cfg1_regenerated_from_seed(oracle):
    """BASELINE.json configs[0]: 100k x 32 -> 10k segments; inputs rebuilt from the recorded seeds."""
    g = IS_CASES["cfg1_100k_x32_10k"]
    index = sorted_index(np.random.default_rng(int(g["seed_index"])), 100_000, 10_000)
    src = np.random.default_rng(int(g["seed_src"])).random((100_000, 32), dtype=np.float32)
    assert sha(index) == str(g["index_sha256"]) and sha(src) == str(g["src_sha256"])
    out = oracle.index_scatter(index, src)
    assert sha(out) == str(g["torch_index_add_sha256"])
    np.testing.assert_array_equal(out[:64], g["torch_index_add_head"])
    quirk = oracle.index_scatter(index, src[index])
    assert sha(quirk) == str(g["ref_out_sha256"])                     # the compiled reference, bit-exact
    np.testing.assert_array_equal(quirk[:64], g["ref_out_head"])
    np.testing.assert_array_equal(oracle.index_scatter_3pass(index, src, threads=2), out)


def test_reductions_match_reference_cpu_path(oracle):
    g = load_golden("reductions.npz")["reductions"]
    index, src = g["index"], g["src"]
    for red in ("sum", "mean", "min", "max", "prod"):
        ours = oracle.index_scatter_3pass(index, src[index], reduce=red)
        np.testing.assert_array_equal(ours, g[f"ref_{red}"], err_msg=red)
        assert np.all(ours[17] == 0)                                   # empty key stays 0 for every reduce
    with pytest.raises(ValueError, match="reduce argument must be either sum, prod, mean, amax or amin, got foo"):
        oracle.index_scatter_3pass(index, src, reduce="foo")
    # NaN propagation of min / max / sum, bit-for-bit (NaN positions included) against the compiled reference
    g = load_golden("reductions.npz")["reductions_nan"]
    for red in ("min", "max", "sum"):
        ours = oracle.index_scatter_3pass(g["index"], g["src"][g["index"]], reduce=red)
        np.testing.assert_array_equal(ours, g[f"ref_{red}"], err_msg="nan " + red)
        assert np.isnan(ours).any()


def test_segment_table_int64_bookkeeping(oracle):
    index = np.array([5, 5, 9, 9, 9, 2**31 + 7, 2**31 + 7, 2**40], dtype=np.int64)
    rows, offs = oracle.segment_table(index)
    np.testing.assert_array_equal(rows, [5, 9, 2**31 + 7, 2**40])
    np.testing.assert_array_equal(offs, [0, 2, 5, 7, 8])
    assert oracle.out_rows(index) == 2**40 + 1
    with pytest.raises(IndexError):
        oracle.out_rows(np.zeros(0, dtype=np.int64))


GATHER = load_golden("gather.npz")


@pytest.mark.parametrize("case", sorted(c for c in GATHER if not c.startswith("mh_")))
def test_gather_ops_match_reference_test_comparators(oracle, case):
    g = GATHER[case]
    gs = oracle.gather_scatter(g["src_index"], g["dst_index"], g["src"])
    gws = oracle.gather_weight_scatter(g["src_index"], g["dst_index"], g["weight"], g["src"])
    # test/test_gather_scatter.py:27 / test_gather_weight_scatter.py:27: allclose(atol=1e-4)
    np.testing.assert_allclose(gs, g["torch_spmm_unweighted"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(gws, g["torch_spmm_weighted"], rtol=1e-5, atol=1e-4)
    hi = oracle.gather_weight_scatter(g["src_index"], g["dst_index"], g["weight"], g["src"], acc64=True)
    assert np.all(np.abs(gws - hi) <= 1e-5 * np.abs(hi) + 1e-30)


@pytest.mark.parametrize("case", sorted(c for c in GATHER if c.startswith("mh_")))
def test_mh_spmm_matches_reference_test_comparator(oracle, case):
    g = GATHER[case]
    a = oracle.mh_spmm(g["src_index"], g["dst_index"], g["weight"], g["src"], transposed=False)
    b = oracle.mh_spmm(g["src_index"], g["dst_index"], np.ascontiguousarray(g["weight"].T), g["src"], transposed=True)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_allclose(a, g["torch_mh"], rtol=1e-5, atol=1e-4)          # test/test_mh_spmm.py:28
    with pytest.raises(ValueError, match="Invalid weight size"):
        oracle.mh_spmm(g["src_index"], g["dst_index"], g["weight"][:-1], g["src"])


def test_sddmm_and_gather_rows(oracle):
    rng = np.random.default_rng(7)
    nodes, nnz, F = 20, 300, 13
    si, di = rng.integers(0, nodes, nnz), rng.integers(0, nodes, nnz)
    m1, m2 = rng.random((nodes, F), dtype=np.float32), rng.random((nodes, F), dtype=np.float32)
    out = oracle.sddmm_coo(si.astype(np.int64), di.astype(np.int64), m1, m2)
    np.testing.assert_allclose(out, (m1[di].astype(np.float64) * m2[si]).sum(-1), rtol=1e-5)
    np.testing.assert_array_equal(oracle.gather_rows(si.astype(np.int64), m1), m1[si])


def test_pyref_autograd_fixture_is_consistent(oracle):
    """The captured outputs of the reference's Python wrappers agree with the oracle's forward ops;
    its d/dsrc formulas agree with dense autograd, its d/dweight (as shipped) does not."""
    g = load_golden("pyref_autograd.npz")["pyref_autograd"]
    fwd = oracle.gather_scatter(g["src_index"], g["dst_index"], g["src"])
    np.testing.assert_allclose(fwd, g["pyref_gs_fwd"], rtol=1e-5, atol=1e-5)
    fwd_w = oracle.gather_weight_scatter(g["src_index"], g["dst_index"], g["weight"], g["src"])
    np.testing.assert_allclose(fwd_w, g["pyref_gws_fwd"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(g["pyref_gws_dsrc"], g["dense_gws_dsrc"], rtol=1e-5, atol=1e-5)
    assert np.abs(g["pyref_gws_dweight_as_shipped"] - g["dense_gws_dweight"]).max() > 0.1   # known defect
    dw = oracle.sddmm_coo(g["src_index"], g["dst_index"], g["grad_out"], g["src"])
    np.testing.assert_allclose(dw, g["dense_gws_dweight"], rtol=1e-5, atol=1e-5)
    assert str(g["schema_gather_scatter"]) == "geot::gather_scatter(Tensor src_index, Tensor dst_index, Tensor src) -> Tensor"


# ---- live comparison with the compiled reference (only where oracle/_ref exists) --------------------
def _ref():
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built (needs /root/reference: `make -C oracle ref`)")
    return ref


@pytest.mark.parametrize("nnz,keys,F", [(1, 1, 1), (257, 3, 5), (4096, 4096, 16), (20000, 777, 33)])
def test_live_identity_against_compiled_reference(oracle, nnz, keys, F):
    ref = _ref()
    rng = np.random.default_rng(nnz + F)
    index = sorted_index(rng, nnz, min(keys, nnz))
    src = rng.standard_normal((nnz, F)).astype(np.float32)
    np.testing.assert_array_equal(ref.index_scatter_cpu(index, src), oracle.index_scatter(index, src[index]))
    if ref.available(omp=True):
        np.testing.assert_array_equal(ref.index_scatter_cpu(index, src, omp=True, threads=4),
                                      oracle.index_scatter(index, src[index]))


def test_live_reference_error_behaviour():
    ref = _ref()
    index = np.array([0, 0, 1], dtype=np.int64)
    src = np.ones((3, 2), dtype=np.float32)
    with pytest.raises(RuntimeError, match="unsorted index is not supported yet"):
        ref.index_scatter_cpu(index, src, sorted=False)
    with pytest.raises(RuntimeError, match="reduce argument must be either sum, prod, mean, amax or amin, got nope"):
        ref.index_scatter_cpu(index, src, reduce="nope")
