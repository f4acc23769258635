"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the golden
vectors, on a real MI355X.  `python -m pytest tests -m gpu`.

Tolerance (BASELINE.json north_star): int64 bookkeeping exact (row count, which rows are written,
empty rows exactly 0); fp32 sums within 1e-5 relative.  "Relative" is taken against the
float64-accumulated oracle and scaled by sum(|contribution|) per output element, which equals the
result itself for the non-negative data the reference's tests use and stays meaningful under
cancellation.  The kernel's summation order (blocked, then merged in edge order) differs from the
oracle's strictly sequential order, so bit-equality with the fp32 oracle is not expected; run-to-run
bit-equality of the sorted path IS (no atomics).
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_golden, powerlaw_index, sorted_index

pytestmark = pytest.mark.gpu

RTOL = 1e-5  # BASELINE.json: "within 1e-5 rel fp32 for sum-reduce"


@pytest.fixture(scope="module")
def geot():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import geot_amd
    return geot_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def assert_close_to_oracle(got, hi, mag, what=""):
    got = got.detach().cpu().numpy()
    assert got.shape == hi.shape, (what, got.shape, hi.shape)
    assert not np.isnan(got).any(), what
    bound = RTOL * mag + 1e-30
    err = np.abs(got.astype(np.float64) - hi.astype(np.float64))
    assert np.all(err <= bound), f"{what}: max err/bound = {np.max(err / bound):.3g}"
    assert np.all(got[mag == 0] == 0), f"{what}: rows without edges must be exactly 0"


def check_index_scatter(geot, oracle, index, src, sorted=True, what=""):
    out = geot.index_scatter(0, dev(src), dev(index), "sum", sorted=sorted)
    assert out.shape[0] == int(index[-1]) + 1
    assert tuple(out.shape[1:]) == tuple(src.shape[1:]) and out.dtype == dev(src).dtype
    hi = oracle.index_scatter(index, src, rows=int(index[-1]) + 1, acc64=src.dtype == np.float32)
    mag = oracle.index_scatter(index, np.abs(src), rows=int(index[-1]) + 1, acc64=src.dtype == np.float32)
    assert_close_to_oracle(out, hi, mag, what)
    return out


IS_CASES = load_golden("index_scatter.npz")


@pytest.mark.parametrize("case", sorted(c for c in IS_CASES if "index" in IS_CASES[c] and c not in ("f16", "bf16_bits")))
@pytest.mark.parametrize("sorted_flag", [True, False])
def test_index_scatter_golden(geot, oracle, case, sorted_flag):
    g = IS_CASES[case]
    out = check_index_scatter(geot, oracle, g["index"], g["src"], sorted=sorted_flag, what=case)
    # the comparators of test/test_index_scatter.py:17-23 with its tolerance
    assert torch.allclose(out.cpu(), torch.from_numpy(g["torch_index_add"]), atol=1e-4)


@pytest.mark.parametrize("case", sorted(c for c in IS_CASES if "index" in IS_CASES[c] and c not in ("f16", "bf16_bits")))
def test_index_scatter_golden_atomic_flush(geot, oracle, case):
    """The atomic (unsorted-index) kernels on the golden inputs, through the pointer-level doorway: the
    operator would route these ascending indices to the atomic-free kernels."""
    from geot_amd import hip
    g = IS_CASES[case]
    index, src = g["index"], g["src"]
    rows = int(index[-1]) + 1
    out = torch.empty((rows,) + tuple(src.shape[1:]), dtype=dev(src).dtype, device="cuda")
    hip.index_scatter_out(dev(index), dev(src), out, sorted=False)
    acc64 = src.dtype == np.float32
    assert_close_to_oracle(out, oracle.index_scatter(index, src, rows=rows, acc64=acc64),
                           oracle.index_scatter(index, np.abs(src), rows=rows, acc64=acc64), case + " atomic")


def test_reference_test_shape_and_flags(geot, oracle):
    """test/test_index_scatter.py:5-23: 1000x32 rand, 10 keys, called with sorted=False on sorted data."""
    torch.manual_seed(0)
    src = torch.rand(1000, 32).cuda()
    index = torch.randint(0, 10, (1000,)).cuda()
    index = index[torch.argsort(index)]
    keys = int(index[-1]) + 1
    for flag in (False, True):
        out = geot.index_scatter(0, src, index, "sum", sorted=flag)
        ref = torch.zeros(keys, 32).cuda().scatter_add_(0, index.unsqueeze(-1).expand_as(src), src)
        assert torch.allclose(out, ref, atol=1e-4)
        ref = torch.zeros(keys, 32).cuda().index_add_(0, index, src)
        assert torch.allclose(out, ref, atol=1e-4)


def test_cfg1_full(geot, oracle):
    g = IS_CASES["cfg1_100k_x32_10k"]
    index = sorted_index(np.random.default_rng(int(g["seed_index"])), 100_000, 10_000)
    src = np.random.default_rng(int(g["seed_src"])).random((100_000, 32), dtype=np.float32)
    out = check_index_scatter(geot, oracle, index, src, what="cfg1")
    np.testing.assert_allclose(out[:64].cpu().numpy(), g["torch_index_add_head"], rtol=RTOL)
    np.testing.assert_allclose(out.double().sum(1).cpu().numpy(), g["torch_index_add_rowsum"], rtol=1e-6)


@pytest.mark.parametrize("F", [1, 2, 3, 4, 5, 7, 8, 12, 16, 31, 32, 33, 48, 64, 65, 100, 128, 200, 256, 260, 512, 1000])
def test_feature_widths(geot, oracle, F):
    rng = np.random.default_rng(F)
    index = sorted_index(rng, 3000, 400)
    check_index_scatter(geot, oracle, index, rng.random((3000, F), dtype=np.float32), what=f"F={F}")


@pytest.mark.parametrize("narrow", [1, 2, 0])
def test_narrow_rows_both_kernels(geot, oracle, narrow):
    """F <= 8 fp32 has three implementations (1: lane-sequential kernel, 2: lane-per-edge scan kernel,
    0: lane groups): all must hold parity on every segment shape, including runs that end exactly on lane-chunk
    (4 / 8 edges), 64-edge step and tile boundaries; the lane-sequential kernel also with max / min / prod and
    with an unaligned src (falls back to the scan kernel)."""
    from geot_amd import hip
    hip.set_option("narrow", narrow)
    try:
        rng = np.random.default_rng(40 + narrow)
        shapes = {
            "powerlaw": powerlaw_index(150_000, 12_000, 3),
            "hub": np.sort(np.concatenate([np.full(9000, 5), rng.integers(0, 40, 3000)])).astype(np.int64),
            "runs_of_64": np.repeat(np.arange(700, dtype=np.int64), 64),       # every run ends on a step boundary
            "runs_of_32": np.repeat(np.arange(1500, dtype=np.int64) * 2, 32),
            "units": np.arange(9000, dtype=np.int64),
            "gaps": np.sort(rng.integers(0, 500, 8000)).astype(np.int64) * 19 + 7,
            "tiny": np.array([0, 0, 3], dtype=np.int64),
        }
        for name, index in shapes.items():
            for F in (1, 2, 3, 4, 5, 6, 7, 8):
                src = rng.standard_normal((len(index), F)).astype(np.float32)
                check_index_scatter(geot, oracle, index, src, what=f"narrow={narrow} {name} F={F}")
                if narrow == 1 and F in (1, 4, 7, 8):
                    for red in ("max", "min", "prod"):
                        s2 = (rng.random((len(index), F), dtype=np.float32) * 0.2 + 0.9) if red == "prod" else src
                        out = geot.index_scatter(0, dev(s2), dev(index), red, True).cpu().numpy()
                        ref = oracle.index_scatter_3pass(index, s2, reduce=red)
                        if red == "prod":
                            ok = np.isclose(out, ref, rtol=1e-2, atol=1e-30) | (np.abs(ref) < 1e-30)
                            assert ok.mean() > 0.999, (name, F, red)
                        else:
                            np.testing.assert_array_equal(out, ref, err_msg=f"{name} F={F} {red}")
            if narrow == 1:   # a src that is only 4-byte aligned: the lane-sequential kernel must not be used
                base = torch.from_numpy(rng.standard_normal(len(index) * 3 + 1).astype(np.float32)).cuda()
                view = base[1:].view(len(index), 3)
                out = geot.index_scatter(0, view, dev(index), "sum", True)
                ref = geot.index_scatter(0, view.clone(), dev(index), "sum", True)
                assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5)
                for F in (2, 4):  # a dst that is only 4-byte aligned (pointer-level doorway): lane groups, scalar stores
                    rows = int(index[-1]) + 1
                    src = torch.from_numpy(rng.standard_normal((len(index), F)).astype(np.float32)).cuda()
                    obase = torch.full((rows * F + 1,), float("nan"), device="cuda")
                    got = hip.index_scatter_out(dev(index), src, obase[1:].view(rows, F), sorted=True)
                    want = hip.index_scatter_out(dev(index), src, torch.empty(rows, F, device="cuda"), sorted=True)
                    # two summation orders of the same row (a hub row adds 9 000 cancelling terms): bound by the row's magnitude
                    mag = torch.zeros(rows, F, device="cuda").index_add_(0, dev(index), src.abs())
                    bad = (got - want).abs() > 2e-6 * mag + 1e-6
                    rows_bad = torch.nonzero(bad.any(1)).flatten()[:8].tolist()
                    assert not rows_bad, (name, F, rows_bad, got[rows_bad].tolist(), want[rows_bad].tolist())
                    assert bool(torch.isnan(obase[0]))
        # NaN propagation through the lane-sequential max / min (ATen semantics, positions exact)
        if narrow == 1:
            index = shapes["powerlaw"]
            nanv = rng.random((len(index), 4), dtype=np.float32)
            nanv[len(index) // 3, 2] = np.nan
            for red in ("max", "min"):
                out = geot.index_scatter(0, dev(nanv), dev(index), red, True).cpu().numpy()
                np.testing.assert_array_equal(np.isnan(out), np.isnan(oracle.index_scatter_3pass(index, nanv, reduce=red)))
    finally:
        hip.set_option("narrow", 1)


@pytest.mark.parametrize("nnz,keys", [(1, 1), (2, 1), (63, 5), (64, 64), (65, 7), (511, 40), (512, 512), (513, 3),
                                      (2047, 100), (2048, 2048), (2049, 9), (100_003, 997)])
def test_ragged_tile_boundaries(geot, oracle, nnz, keys):
    rng = np.random.default_rng(nnz)
    index = sorted_index(rng, nnz, keys)
    check_index_scatter(geot, oracle, index, rng.random((nnz, 64), dtype=np.float32), what=f"nnz={nnz}")
    check_index_scatter(geot, oracle, index, rng.random((nnz, 6), dtype=np.float32), what=f"nnz={nnz} F=6")


def test_segment_shapes(geot, oracle):
    rng = np.random.default_rng(11)
    f = lambda n, F=64: rng.random((n, F), dtype=np.float32)  # noqa: E731
    check_index_scatter(geot, oracle, np.zeros(70_000, dtype=np.int64), f(70_000), what="one hub over many tiles")
    check_index_scatter(geot, oracle, np.full(70_000, 9, dtype=np.int64), f(70_000), what="hub, rows 0-8 empty")
    check_index_scatter(geot, oracle, np.arange(30_000, dtype=np.int64), f(30_000), what="unit segments")
    check_index_scatter(geot, oracle, np.arange(30_000, dtype=np.int64) * 3, f(30_000, 32), what="small gaps")
    check_index_scatter(geot, oracle, np.arange(20_000, dtype=np.int64) * 40, f(20_000, 8), what="large gaps")
    idx = sorted_index(rng, 9000, 300)
    idx = idx + np.where(idx >= 150, 100_000, 5000)
    check_index_scatter(geot, oracle, idx, f(9000), what="leading gap 5000, middle gap 100k")
    hubs = np.sort(np.concatenate([np.full(40_000, 3), np.full(25_000, 4), rng.integers(0, 50, 5000)])).astype(np.int64)
    check_index_scatter(geot, oracle, hubs, f(len(hubs)), what="two adjacent hubs")
    check_index_scatter(geot, oracle, powerlaw_index(400_000, 30_000, 3), f(400_000), what="power law")


def test_mixed_sign_and_fp64(geot, oracle):
    rng = np.random.default_rng(12)
    index = powerlaw_index(50_000, 4000, 5)
    check_index_scatter(geot, oracle, index, rng.standard_normal((50_000, 64)).astype(np.float32), what="signed f32")
    for F in (1, 6, 64, 130):
        src = rng.standard_normal((50_000, F))
        out = geot.index_scatter(0, dev(src), dev(index))
        assert out.dtype == torch.float64
        hi = oracle.index_scatter(index, src)
        mag = oracle.index_scatter(index, np.abs(src))
        assert np.all(np.abs(out.cpu().numpy() - hi) <= 1e-13 * mag + 1e-300)


def test_dim_and_nd_src(geot, oracle):
    rng = np.random.default_rng(13)
    index = sorted_index(rng, 500, 40)
    src = rng.random((500, 3, 5), dtype=np.float32)
    out = check_index_scatter(geot, oracle, index, src, what="3-D src")
    assert tuple(out.shape) == (40, 3, 5)
    src_t = np.ascontiguousarray(np.moveaxis(src, 0, 1))            # [3, 500, 5], reduce along dim 1
    out1 = geot.index_scatter(1, dev(src_t), dev(index))
    assert tuple(out1.shape) == (3, 40, 5)
    assert torch.equal(out1.movedim(1, 0), out)
    # non-contiguous src view
    wide = dev(rng.random((500, 20), dtype=np.float32))
    view = wide[:, ::2]
    ref = geot.index_scatter(0, view.contiguous(), dev(index))
    assert torch.equal(geot.index_scatter(0, view, dev(index)), ref)


def test_sorted_path_is_deterministic_and_atomic_free(geot):
    index = dev(powerlaw_index(2_000_000, 200_000, 9))
    src = torch.rand(2_000_000, 64, device="cuda")
    a = geot.index_scatter(0, src, index)
    for _ in range(3):
        assert torch.equal(geot.index_scatter(0, src, index), a)      # bit-identical run to run
    b = geot.index_scatter(0, src, index, sorted=False)                # probe finds it ascending: same kernels
    assert torch.equal(a, b)
    from geot_amd import hip
    c = hip.index_scatter_out(index, src, torch.empty_like(a), sorted=False)   # float atomics: only close
    assert torch.allclose(a, c, rtol=1e-5, atol=1e-5)


def test_sort_index_matches_a_stable_sort_bit_for_bit(geot):
    """geot_sort_index (32-bit radix passes over the bits in use) against numpy's stable argsort: keys and positions
    equal exactly - key ranges from one bit to 2^32-1, odd lengths, a misaligned index, ties everywhere; and the range
    probe (last, descents, min, max) that sizes it.  The host layer's unsorted path reduces over this sort."""
    from geot_amd import hip
    rng = np.random.default_rng(2024)
    cases = [(1, 1), (2, 5), (1000, 1), (1001, 2), (4097, 37), (70_001, 1 << 20), (300_000, 1 << 31),
             (300_001, (1 << 32) - 1), (2_000_003, 1000), (2_000_000, 1_000_000)]
    for nnz, top in cases:
        index = rng.integers(0, top + 1, nnz, dtype=np.int64)
        index[rng.integers(0, nnz)] = top                                   # the declared maximum is present
        for off in (0, 1):                                                    # off=1: data pointer 8 mod 16
            buf = torch.empty(nnz + off, dtype=torch.int64, device="cuda")
            t = buf[off:]
            t.copy_(torch.from_numpy(index))
            probe = hip.index_probe_range_out(t, torch.empty(4, dtype=torch.int64, device="cuda")).cpu().tolist()
            assert probe == [int(index[-1]), int((index[:-1] > index[1:]).sum()), int(index.min()), int(index.max())], (nnz, top)
            keys, perm = hip.sort_index(t, int(index.max()))
            order = np.argsort(index, kind="stable")
            assert np.array_equal(perm.cpu().numpy(), order), (nnz, top, off)
            assert np.array_equal(keys.cpu().numpy(), index[order]), (nnz, top, off)
    L = hip._lib.load()
    assert L.geot_sort_supported(10, 0, (1 << 32) - 1) == 1
    assert L.geot_sort_supported(10, -1, 5) == 0 and L.geot_sort_supported(10, 0, 1 << 32) == 0


def test_unsorted_index_outside_the_fast_sort_range(geot, oracle):
    """Keys the 32-bit sort cannot take (a negative key is invalid input and must not be reached silently; keys at or
    above 2^32 with a small index[-1]) go through the generic sort: rows stay index[-1]+1, keys beyond are ignored."""
    rng = np.random.default_rng(9)
    nnz, K, F = 5000, 40, 8
    index = rng.integers(0, K, nnz).astype(np.int64)
    index[7] = (1 << 33) + 5                                                  # beyond rows: ignored, as the reference's rule implies
    index[-1] = K - 1
    src = rng.random((nnz, F), dtype=np.float32)
    got = geot.index_scatter(0, dev(src), dev(index), "sum", sorted=False).cpu().numpy()
    keep = index < K
    want = np.zeros((K, F), dtype=np.float64)
    np.add.at(want, index[keep], src[keep].astype(np.float64))
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)


def test_sorted_false_is_routed_by_the_probe(geot, oracle, monkeypatch):
    """sorted=False promises nothing: an ascending index is served by the atomic-free kernels (all
    reductions), an index with descents by the atomic path (sum only); both sized by index[-1]+1."""
    from geot_amd import hip, ops
    rng = np.random.default_rng(77)
    for nnz, K in ((3000, 200), (300_000, 20_000)):                   # below / above the speculation threshold
        index = np.sort(rng.integers(0, K, nnz)).astype(np.int64)
        index[-1] = K - 1
        src = rng.random((nnz, 16), dtype=np.float32)
        probe = hip.index_probe_out(dev(index), torch.empty(2, dtype=torch.int64, device="cuda")).cpu()
        assert probe.tolist() == [K - 1, 0]
        for red in ("sum", "mean", "max"):
            a = geot.index_scatter(0, dev(src), dev(index), red, sorted=True)
            for _ in range(2):                                        # second call takes the speculated order
                b = geot.index_scatter(0, dev(src), dev(index), red, sorted=False)
                assert torch.equal(a, b), (nnz, red)                  # ascending: the same atomic-free kernels
        shuffled = index.copy()
        shuffled[:-1] = rng.permutation(shuffled[:-1])
        descents = int((shuffled[:-1] > shuffled[1:]).sum())
        probe = hip.index_probe_out(dev(shuffled), torch.empty(2, dtype=torch.int64, device="cuda")).cpu()
        assert probe.tolist() == [K - 1, descents] and descents > 0
        check_index_scatter(geot, oracle, shuffled, src, sorted=False, what=f"probe-routed unsorted {nnz}")
        for red in ("mean", "max"):           # any reduction on an index with descents: sorted-gather path
            got = geot.index_scatter(0, dev(src), dev(shuffled), red, sorted=False).cpu().numpy()
            order = np.argsort(shuffled, kind="stable")
            want = oracle.index_scatter_3pass(shuffled[order], src[order], reduce=red, rows=K)
            if red == "max":
                assert np.array_equal(got, want)
            else:
                np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6)
        # same tensor object, content changed in place between calls (copy_ moves the version counter):
        # the remembered facts must not be reused
        t = dev(index)
        first = geot.index_scatter(0, dev(src), t, "sum", sorted=False)
        t.copy_(dev(shuffled))
        second = geot.index_scatter(0, dev(src), t, "sum", sorted=False)
        want = oracle.index_scatter(shuffled, src, acc64=True)
        np.testing.assert_allclose(second.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
        assert first.shape == second.shape
    for mode in ("atomic", "sort"):                                   # GEOT_UNSORTED forces either unsorted path
        old = ops.set_option("unsorted_mode", mode)
        try:
            index = rng.integers(0, 50, 4000).astype(np.int64)
            index[-1] = 49
            src = rng.random((4000, 8), dtype=np.float32)
            check_index_scatter(geot, oracle, index, src, sorted=False, what=f"unsorted mode {mode}")
        finally:
            ops.set_option("unsorted_mode", old)


def test_unsorted_index_with_sorted_false(geot, oracle):
    rng = np.random.default_rng(14)
    index = rng.integers(0, 5000, 100_000).astype(np.int64)
    index[-1] = 4999                                                  # row rule still reads index[-1]
    for F in (64, 33, 4):
        src = rng.random((100_000, F), dtype=np.float32)
        check_index_scatter(geot, oracle, index, src, sorted=False, what=f"unsorted F={F}")


def test_workspace_reuse_across_shapes_and_streams(geot, oracle):
    """Control words must come back to zero after every call (large-gap list is exercised first)."""
    rng = np.random.default_rng(15)
    big_gap = np.arange(5000, dtype=np.int64) * 50
    normal = sorted_index(rng, 40_000, 3000)
    s1 = torch.cuda.Stream()
    for rep in range(3):
        check_index_scatter(geot, oracle, big_gap, rng.random((5000, 16), dtype=np.float32), what="gap list")
        check_index_scatter(geot, oracle, normal, rng.random((40_000, 64), dtype=np.float32), what="after gap list")
        with torch.cuda.stream(s1):
            check_index_scatter(geot, oracle, big_gap, rng.random((5000, 8), dtype=np.float32), what="side stream")
        s1.synchronize()


def test_int64_keys_beyond_2_31(geot):
    """Row numbers and offsets are 64-bit (the reference truncates keys to int and overflows at 2^31
    elements: csrc/cuda/index_scatter_kernel.cuh:137,159,166)."""
    base = 2**31 + 5
    index = torch.tensor([base, base, base + 1], device="cuda")
    src = torch.tensor([[1.0], [2.0], [4.0]], device="cuda")
    out = geot.index_scatter(0, src, index)
    assert out.shape == (base + 2, 1)
    assert out[base].item() == 3.0 and out[base + 1].item() == 4.0
    assert out[: 1 << 20].abs().sum().item() == 0 and out[base - 1000: base].abs().sum().item() == 0
    assert out.sum(dtype=torch.float64).item() == 7.0


def test_large_element_offsets(geot):
    """nnz * F beyond 2^31 elements (cfg2 x 4 rows wide enough): offsets must be 64-bit."""
    nnz, F = 9_000_000, 256                                             # 2.3e9 elements, 9.2 GB
    index = dev(powerlaw_index(nnz, 500_000, 21))
    src = torch.rand(nnz, F, device="cuda")
    out = geot.index_scatter(0, src, index)
    # float64 reference (torch's fp32 index_add_ adds with atomics in arbitrary order: on the 30 k-edge hub it is itself
    # only good to ~1e-5 relative, which made this check flaky), column block by column block
    for c0 in range(0, F, 64):
        ref = torch.zeros(500_000, 64, device="cuda", dtype=torch.float64).index_add_(0, index, src[:, c0:c0 + 64].double())
        assert bool(((out[:, c0:c0 + 64].double() - ref).abs() <= 1e-5 * ref.abs() + 1e-30).all())   # (non-negative data)
        del ref
    assert torch.allclose(out.double().sum(0), src.double().sum(0), rtol=1e-6)
    # the last segments are fed by rows that live above 2^31 elements
    for k in torch.unique(index[-5000:])[-5:].tolist():
        sel = (index == k).nonzero().flatten()
        assert sel.min().item() * F > 2**31
        assert torch.allclose(out[k].double(), src[sel].double().sum(0), rtol=1e-5)


GATHER = load_golden("gather.npz")


@pytest.mark.parametrize("case", sorted(c for c in GATHER if not c.startswith("mh_")))
def test_gather_ops_golden(geot, oracle, case):
    g = GATHER[case]
    si, di, w, src = g["src_index"], g["dst_index"], g["weight"], g["src"]
    out = geot.gather_scatter(dev(si), dev(di), dev(src))
    hi = oracle.gather_scatter(si, di, src, acc64=True)
    assert_close_to_oracle(out, hi, hi, case + " gs")
    assert torch.allclose(out.cpu(), torch.from_numpy(g["torch_spmm_unweighted"]), atol=1e-4)
    assert torch.equal(geot.gather_scatter(dev(si), dev(di), dev(src), "sum"), out)   # stale 4-arg callers
    out = geot.gather_weight_scatter(dev(si), dev(di), dev(w), dev(src))
    hi = oracle.gather_weight_scatter(si, di, w, src, acc64=True)
    assert_close_to_oracle(out, hi, hi, case + " gws")
    assert torch.allclose(out.cpu(), torch.from_numpy(g["torch_spmm_weighted"]), atol=1e-4)


@pytest.mark.parametrize("case", sorted(c for c in GATHER if c.startswith("mh_")))
def test_mh_spmm_golden(geot, oracle, case):
    g = GATHER[case]
    si, di, w, src = g["src_index"], g["dst_index"], g["weight"], g["src"]
    hi = oracle.mh_spmm(si, di, w, src, acc64=True)
    a = geot.mh_spmm(dev(si), dev(di), dev(w), dev(src), "sum")                       # weight [nnz, H]
    b = geot.mh_spmm_transposed(dev(si), dev(di), dev(w), dev(src), "sum")            # -> [H, nnz]
    assert_close_to_oracle(a, hi, hi, case)
    assert_close_to_oracle(b, hi, hi, case + " transposed")
    assert torch.allclose(a.cpu(), torch.from_numpy(g["torch_mh"]), atol=1e-4)        # test/test_mh_spmm.py:28


@pytest.mark.parametrize("nodes,nnz,F", [(100, 1000, 32), (5000, 200_000, 128), (3000, 100_000, 7), (20_000, 500_000, 64)])
def test_gather_ops_random_graphs(geot, oracle, nodes, nnz, F):
    rng = np.random.default_rng(nodes + F)
    si = rng.integers(0, nodes, nnz).astype(np.int64)
    di = powerlaw_index(nnz, nodes, nodes)
    w = rng.random(nnz, dtype=np.float32)
    src = rng.random((nodes, F), dtype=np.float32)
    hi = oracle.gather_weight_scatter(si, di, w, src, acc64=True)
    assert_close_to_oracle(geot.gather_weight_scatter(dev(si), dev(di), dev(w), dev(src)), hi, hi, "gws")
    hi = oracle.gather_scatter(si, di, src, acc64=True)
    assert_close_to_oracle(geot.gather_scatter(dev(si), dev(di), dev(src)), hi, hi, "gs")
    if F % 4 == 0:
        H = 4
        wh = rng.random((nnz, H), dtype=np.float32)
        s3 = src.reshape(nodes, H, F // H)
        hi = oracle.mh_spmm(si, di, wh, s3, acc64=True)
        assert_close_to_oracle(geot.mh_spmm(dev(si), dev(di), dev(wh), dev(s3)), hi, hi, "mh")
        assert_close_to_oracle(geot.mh_spmm(dev(si), dev(di), dev(np.ascontiguousarray(wh.T)), dev(s3)), hi, hi, "mh^T")


@pytest.mark.parametrize("xcd", [1, 0])
def test_gather_ops_under_both_tile_mappings(geot, oracle, xcd):
    """The XCD-aware block->tile remap of the gather modes is pure placement: results identical either way
    (tile counts that are and are not multiples of 8, partial last tile, hubs)."""
    from geot_amd import hip
    hip.set_option("xcd", xcd)
    try:
        rng = np.random.default_rng(60)
        for nodes, nnz, F in ((4000, 70_000, 32), (900, 8192 + 5, 64), (20_000, 1024 * 8, 16), (50, 30_000, 128)):
            si = rng.integers(0, nodes, nnz).astype(np.int64)
            di = powerlaw_index(nnz, nodes, nodes)
            w = rng.random(nnz, dtype=np.float32)
            src = rng.random((nodes, F), dtype=np.float32)
            hi = oracle.gather_weight_scatter(si, di, w, src, acc64=True)
            assert_close_to_oracle(geot.gather_weight_scatter(dev(si), dev(di), dev(w), dev(src)), hi, hi, f"xcd={xcd}")
    finally:
        hip.set_option("xcd", 1)


@pytest.mark.parametrize("reduce", ["add", "sum", "mean", "max", "min", "prod"])
def test_gather_ops_take_pyg_aggr_as_reduce(geot, reduce):
    """models/conv/spmm.py forwards the layer's aggr as the trailing argument of the gather ops."""
    rng = np.random.default_rng(70)
    for nodes, nnz, F in ((400, 9000, 32), (3000, 50_000, 7), (60, 40_000, 64)):
        di_h = powerlaw_index(nnz, nodes, nodes)
        si = dev(rng.integers(0, nodes, nnz).astype(np.int64))
        di = dev(di_h)
        x = dev((rng.random((nodes, F), dtype=np.float32) * (0.3 if reduce == "prod" else 1.0) + (0.85 if reduce == "prod" else 0.0)))
        w = dev(rng.random(nnz, dtype=np.float32) + 0.5)
        kind = {"add": "sum", "max": "amax", "min": "amin"}.get(reduce, reduce)
        for weight in (None, w):
            msg = x[si] if weight is None else x[si] * weight[:, None]
            ref = torch.zeros(nodes, F, device="cuda").scatter_reduce(0, di[:, None].expand(-1, F), msg, kind, include_self=False)
            out = geot.gather_scatter(si, di, x, reduce) if weight is None else geot.gather_weight_scatter(si, di, weight, x, reduce)
            assert out.shape == ref.shape
            if reduce in ("max", "min"):
                assert torch.equal(out, ref)
            elif reduce == "prod":
                assert torch.allclose(out, ref, rtol=1e-2, atol=1e-30)
            else:
                assert torch.allclose(out, ref, rtol=1e-4, atol=1e-4)
    with pytest.raises(RuntimeError, match="reduce argument must be either"):
        geot.gather_scatter(si, di, x, "median")


def test_sddmm_and_autograd_against_golden(geot, oracle):
    g = load_golden("pyref_autograd.npz")["pyref_autograd"]
    si, di = dev(g["src_index"]), dev(g["dst_index"])
    gout = dev(g["grad_out"])
    # forward + d/dsrc exactly what the reference's Python wrappers produce
    s = dev(g["src"]).requires_grad_(True)
    out = geot.gather_scatter(si, di, s)
    out.backward(gout)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["pyref_gs_fwd"], rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(s.grad.cpu().numpy(), g["pyref_gs_dsrc"], rtol=RTOL, atol=1e-6)
    s = dev(g["src"]).requires_grad_(True)
    w = dev(g["weight"]).requires_grad_(True)
    out = geot.gather_weight_scatter(si, di, w, s)
    out.backward(gout)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["pyref_gws_fwd"], rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(s.grad.cpu().numpy(), g["pyref_gws_dsrc"], rtol=RTOL, atol=1e-6)
    # d/dweight: dense autograd is the truth; the reference's shipped formula is not
    np.testing.assert_allclose(w.grad.cpu().numpy(), g["dense_gws_dweight"], rtol=RTOL, atol=1e-6)
    sd = geot.sddmm_coo_impl(si, di, gout, dev(g["src"]))
    np.testing.assert_allclose(sd.cpu().numpy(), oracle.sddmm_coo(g["src_index"], g["dst_index"], g["grad_out"], g["src"], acc64=True),
                               rtol=RTOL)


def test_index_scatter_backward_is_a_row_gather(geot):
    index = dev(powerlaw_index(30_000, 2000, 4))
    for shape, dim in (((30_000, 48), 0), ((30_000, 3, 5), 0), ((7, 30_000), 1)):
        src = torch.rand(*shape, device="cuda", requires_grad=True)
        ref_src = src.detach().clone().requires_grad_(True)
        out = geot.index_scatter(dim, src, index)
        g = torch.rand_like(out)
        out.backward(g)
        zeros = torch.zeros_like(out)
        zeros.index_add(dim, index, ref_src).backward(g)
        assert torch.equal(src.grad, ref_src.grad)                     # a pure gather: exact
    with pytest.raises(NotImplementedError, match="backward is implemented for reduce='sum'"):
        src = torch.rand(30_000, 4, device="cuda", requires_grad=True)
        geot.index_scatter(0, src, index, "max").sum().backward()


def test_backward_reuses_and_invalidates_the_transposed_edge_list(geot):
    """The source-sorted edge list is remembered per tensor identity+version; an in-place edit (version bump)
    must produce a fresh one.  Gradients are compared with dense autograd each time."""
    torch.manual_seed(3)
    n, nnz, F = 300, 6000, 8
    di = torch.sort(torch.randint(0, n, (nnz,), device="cuda")).values
    di[-1] = n - 1
    si = torch.randint(0, n, (nnz,), device="cuda")
    w0 = torch.rand(nnz, device="cuda")
    x0 = torch.rand(n, F, device="cuda")
    g = torch.rand(n, F, device="cuda")
    for it in range(4):
        if it == 2:
            si[:100] = torch.randint(0, n, (100,), device="cuda")     # in-place edit: bumps si._version
        x1, w1 = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        x2, w2 = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        geot.gather_weight_scatter(si, di, w1, x1).backward(g)
        torch.zeros(n, F, device="cuda").index_add(0, di, x2[si] * w2[:, None]).backward(g)
        assert torch.allclose(x1.grad, x2.grad, rtol=1e-4, atol=1e-4), it
        assert torch.allclose(w1.grad, w2.grad, rtol=1e-4, atol=1e-4), it


def test_backward_when_last_node_has_no_out_edge(geot):
    """The reference returns max(src_index)+1 grad rows (shape error); here grad has src.shape[0] rows."""
    si = torch.tensor([0, 1, 1, 0], device="cuda")
    di = torch.tensor([0, 0, 1, 2], device="cuda")
    src = torch.rand(5, 8, device="cuda", requires_grad=True)           # nodes 2..4 have no out-edge
    geot.gather_scatter(si, di, src).sum().backward()
    assert src.grad.shape == (5, 8) and src.grad[2:].abs().sum().item() == 0
    assert torch.allclose(src.grad[:2], torch.full((2, 8), 2.0, device="cuda"))


@pytest.mark.parametrize("nnz,publish", [(5000, 1), (5000, 0), (1_200_000, 1)])
def test_row_rule_read_back_three_ways(geot, nnz, publish):
    """index[-1] reaches the host (a) from the call's own first kernel, which writes it and a sequence number into pinned
    memory while it runs (calls up to 1 M edges: geot_publish_word - no copy, no event), (b) by an 8-byte copy queued
    ahead of the kernels (GEOT_PUBLISH_ROWS=0, and every larger call).  Each way catches a silent `.data` edit, for
    every kernel family that publishes (tile, lane-sequential F <= 8, weighted gather) and for paths that do not."""
    from geot_amd import ops
    old = ops.set_option("publish_rows", publish)
    try:
        torch.manual_seed(nnz)
        K = 700
        index = torch.sort(torch.randint(0, K, (nnz,), device="cuda")).values
        index[-1] = K - 1
        si = torch.randint(0, K, (nnz,), device="cuda")
        w = torch.rand(nnz, device="cuda")
        for F in (1, 4, 48):
            src = torch.rand(nnz, F, device="cuda")
            x = torch.rand(K, F, device="cuda")
            st0 = ops.stats()
            for last in (K - 1, K - 1, K + 20, K - 1):
                index.data[-1] = last                                # same identity, same version counter
                ref = torch.zeros(last + 1, F, device="cuda", dtype=torch.float64).index_add_(0, index, src.double())
                out = geot.index_scatter(0, src, index)
                assert out.shape == (last + 1, F) and torch.allclose(out.double(), ref, rtol=1e-5, atol=1e-4), (F, last)
                out = geot.gather_weight_scatter(si, index, w, x)
                ref = torch.zeros(last + 1, F, device="cuda", dtype=torch.float64).index_add_(0, index, x.double()[si] * w.double()[:, None])
                assert out.shape == (last + 1, F) and torch.allclose(out.double(), ref, rtol=1e-5, atol=1e-4), (F, last)
            st1 = ops.stats()
            assert st1["row_mismatches"] - st0["row_mismatches"] >= 2              # (the op that runs first after an edit sees it)
            took_publish = st1["published"] - st0["published"]
            assert (took_publish >= 6) if (publish and nnz <= (1 << 20)) else (took_publish == 0), (F, took_publish)
    finally:
        ops.set_option("publish_rows", old)


def test_row_rule_is_verified_on_every_call(geot, oracle, monkeypatch):
    """The row count remembered for an index tensor is only a guess: index[-1] is read back and checked
    on every call.  `.data` writes change the content without bumping the version counter."""
    from geot_amd import ops
    assert ops.get_option("speculate_rows") == 1              # the host layer speculates at every size
    mism = ops.stats()["row_mismatches"]
    rng = np.random.default_rng(21)
    index_h = sorted_index(rng, 4000, 300)
    src_h = rng.random((4000, 32), dtype=np.float32)
    index, src = dev(index_h), dev(src_h)
    for _ in range(3):                                           # first call learns, later ones speculate
        out = geot.index_scatter(0, src, index)
        assert out.shape[0] == 300
    v0 = index._version
    index.data[-1] = 450                                         # more rows, same identity + version
    index_h[-1] = 450
    assert index._version == v0
    out = geot.index_scatter(0, src, index)
    assert out.shape[0] == 451
    assert_close_to_oracle(out, oracle.index_scatter(index_h, src_h, acc64=True),
                           oracle.index_scatter(index_h, np.abs(src_h), acc64=True), "grown")
    index.data[-1] = 299                                         # fewer rows again
    index_h[-1] = 299
    for _ in range(2):
        out = geot.index_scatter(0, src, index)
        assert out.shape[0] == 300
        assert_close_to_oracle(out, oracle.index_scatter(index_h, src_h, acc64=True),
                               oracle.index_scatter(index_h, np.abs(src_h), acc64=True), "shrunk")
    si = dev(rng.integers(0, 300, 4000).astype(np.int64))
    for last in (299, 320, 299):
        index.data[-1] = last
        assert geot.gather_scatter(si, index, src[:300].contiguous()).shape[0] == last + 1
    assert ops.stats()["row_mismatches"] >= mism + 4          # every silent edit was caught by the read-back
    old = ops.set_option("trust_version", 2)                  # GEOT_TRUST_VERSION=2: no read-back at all for a known content
    try:
        idx2 = index.clone()
        for _ in range(3):
            assert geot.index_scatter(0, src, idx2).shape[0] == int(idx2[-1]) + 1
        idx2[-1] = 400                                            # an in-place edit moves the version counter: seen
        assert geot.index_scatter(0, src, idx2).shape[0] == 401
    finally:
        ops.set_option("trust_version", old)
    old = ops.set_option("speculate_rows", 0)                 # GEOT_SPECULATE_ROWS=0: the reference's blocking order
    try:
        index.data[-1] = 310
        assert geot.index_scatter(0, src, index).shape[0] == 311
    finally:
        ops.set_option("speculate_rows", old)


@pytest.mark.parametrize("reduce", ["mean", "min", "amin", "max", "amax", "prod", "sum"])
def test_reductions_match_reference_cpu_semantics(geot, oracle, reduce):
    """All reductions of the reference's CPU path (golden: the compiled reference itself)."""
    g = load_golden("reductions.npz")["reductions"]
    index, src = g["index"], g["src"]
    # the shipped CPU kernel reduces src[index[n]]; feed that operand so the captured output applies
    out = geot.index_scatter(0, dev(src[index]), dev(index), reduce, True).cpu().numpy()
    ref = g["ref_" + {"amin": "min", "amax": "max"}.get(reduce, reduce)]
    if reduce in ("min", "amin", "max", "amax"):
        np.testing.assert_array_equal(out, ref)                          # order-independent: exact
    else:
        np.testing.assert_allclose(out, ref, rtol=2e-5)                  # prod of ~8 factors / mean
    assert np.all(out[17] == 0)                                          # empty key stays 0
    if reduce in ("min", "max", "sum"):                                  # NaN propagation like ATen, NaN positions exact
        n = load_golden("reductions.npz")["reductions_nan"]
        out = geot.index_scatter(0, dev(n["src"][n["index"]]), dev(n["index"]), reduce, True).cpu().numpy()
        np.testing.assert_array_equal(np.isnan(out), np.isnan(n["ref_" + reduce]))
        ok = ~np.isnan(out)
        np.testing.assert_allclose(out[ok], n["ref_" + reduce][ok], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("reduce", ["mean", "min", "max", "prod"])
@pytest.mark.parametrize("shape", ["powerlaw", "hub", "gaps", "units"])
def test_reductions_all_segment_shapes(geot, oracle, reduce, shape):
    rng = np.random.default_rng(31)
    if shape == "powerlaw":
        index = powerlaw_index(200_000, 15_000, 2)
    elif shape == "hub":
        index = np.sort(np.concatenate([np.full(30_000, 5), rng.integers(0, 40, 3000)])).astype(np.int64)
    elif shape == "gaps":
        index = np.sort(rng.integers(0, 300, 5000)).astype(np.int64) * 21 + 40
    else:
        index = np.arange(5000, dtype=np.int64)
    nnz = len(index)
    for F in (64, 5):
        src = (rng.random((nnz, F), dtype=np.float32) * (0.2 if reduce == "prod" else 1.0) + (0.9 if reduce == "prod" else 0.0))
        src[rng.integers(0, nnz, 50)] *= -1.0
        out = geot.index_scatter(0, dev(src), dev(index), reduce, True).cpu().numpy()
        ref = oracle.index_scatter_3pass(index, src, reduce=reduce)
        if reduce in ("min", "max"):
            np.testing.assert_array_equal(out, ref)
        elif reduce == "mean":
            np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-5)
        else:  # prod over up to 30k factors near 1: compare in log space where the factors are positive
            ok = np.isclose(out, ref, rtol=1e-2, atol=1e-30) | (np.abs(ref) < 1e-30)
            assert ok.mean() > 0.999
    nanv = rng.random((nnz, 8), dtype=np.float32)
    nanv[nnz // 2, 3] = np.nan
    if reduce in ("min", "max"):
        out = geot.index_scatter(0, dev(nanv), dev(index), reduce, True).cpu().numpy()
        np.testing.assert_array_equal(np.isnan(out), np.isnan(oracle.index_scatter_3pass(index, nanv, reduce=reduce)))


@pytest.mark.parametrize("hub", [-1, 1, 0])
def test_few_keys_long_chains(geot, oracle, hub):
    """Global-pooling shaped inputs (a handful of keys, runs spanning hundreds of tiles): the chain of tile
    carries is reduced through per-window sums (seg_wsum_kernel; hub=-1: the library's own nnz/K rule,
    1: forced, 0: tile-by-tile walk) - every reduction, fp32 and bf16, streamed and gathered rows."""
    from geot_amd import hip
    rng = np.random.default_rng(41)
    runs = [(0, 150_001), (1, 9), (2, 70_000), (7, 300), (8, 131_072 + 64 * 512), (9, 1), (30, 90_000)]
    index = np.concatenate([np.full(n, k, dtype=np.int64) for k, n in runs])
    nnz = len(index)
    hip.set_option("hub", hub)
    try:
        for F in (64, 6, 24):
            src = rng.random((nnz, F), dtype=np.float32) - 0.25
            for reduce in ("sum", "mean", "max", "min"):
                out = geot.index_scatter(0, dev(src), dev(index), reduce, True).cpu().numpy()
                if reduce == "sum":
                    hi = oracle.index_scatter(index, src, acc64=True)
                    mag = oracle.index_scatter(index, np.abs(src), acc64=True)
                    assert np.all(np.abs(out - hi) <= 2e-5 * mag + 1e-6), (F, reduce)
                else:
                    ref = oracle.index_scatter_3pass(index, src, reduce=reduce)
                    if reduce == "mean":
                        np.testing.assert_allclose(out, ref, rtol=2e-4, atol=2e-5)
                    else:
                        np.testing.assert_array_equal(out, ref)
                assert np.all(out[10:30] == 0) and np.all(out[3:7] == 0)
        src = torch.from_numpy(rng.random((nnz, 64), dtype=np.float32)).to(torch.bfloat16)
        out = geot.index_scatter(0, src.cuda(), dev(index), "sum", True).float().cpu().numpy()
        hi = oracle.index_scatter(index, src.float().numpy(), acc64=True)
        assert np.all(np.abs(out - hi) <= 2.0 ** -8 * np.abs(hi) + 1e-3)
        si = rng.integers(0, 31, nnz).astype(np.int64)
        x = rng.random((31, 32), dtype=np.float32)
        w = rng.random(nnz, dtype=np.float32)
        out = geot.gather_weight_scatter(dev(si), dev(index), dev(w), dev(x)).cpu().numpy()
        hi = oracle.gather_weight_scatter(si, index, w, x, acc64=True)
        assert np.all(np.abs(out - hi) <= 2e-5 * np.abs(hi) + 1e-6)
    finally:
        hip.set_option("hub", -1)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_16bit_storage_fp32_accumulate(geot, oracle, dtype):
    """half / bfloat16 inputs: accumulate in fp32, round once (reference CPU semantics; golden = compiled reference)."""
    case = IS_CASES["f16" if dtype == torch.float16 else "bf16_bits"]
    index = case["index"]
    if dtype == torch.float16:
        src_t = torch.from_numpy(case["src"])
        ref_t = torch.from_numpy(case["ref_out"])
    else:
        src_t = torch.from_numpy(case["src"].view(np.int16)).view(torch.bfloat16)
        ref_t = torch.from_numpy(case["ref_out"].view(np.int16)).view(torch.bfloat16)
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    # the captured reference output was computed on src[index] (its operand quirk): feed the same operand
    out = geot.index_scatter(0, src_t[torch.from_numpy(index)].cuda(), dev(index))
    assert out.dtype == dtype and out.shape == ref_t.shape
    diff = (out.float().cpu() - ref_t.float()).abs()
    assert torch.all(diff <= ulp * ref_t.float().abs() + 1e-6), diff.max()
    # larger shapes, all ops, against the fp32-accumulated oracle rounded once
    rng = np.random.default_rng(5)
    for nnz, keys, F in ((50_000, 3000, 64), (20_000, 900, 40), (30_000, 5000, 6), (9000, 70, 3), (40_000, 100, 256)):
        idx = powerlaw_index(nnz, keys, F)
        s32 = torch.from_numpy(rng.standard_normal((nnz, F)).astype(np.float32)).to(dtype)
        hi = oracle.index_scatter(idx, s32.float().numpy(), acc64=True)
        mag = oracle.index_scatter(idx, s32.float().abs().numpy(), acc64=True)
        for red in ("sum", "max", "mean"):
            out = geot.index_scatter(0, s32.cuda(), dev(idx), red).float().cpu().numpy()
            if red == "sum":
                assert np.all(np.abs(out - hi) <= ulp * np.abs(hi) + 2e-5 * mag + 1e-6), (nnz, F)
            elif red == "max":
                ref = oracle.index_scatter_3pass(idx, s32.float().numpy(), reduce="max")
                np.testing.assert_array_equal(out, ref)                  # exact: max of representable values
        si = rng.integers(0, keys, nnz).astype(np.int64)
        x = torch.from_numpy(rng.standard_normal((keys, F)).astype(np.float32)).to(dtype)
        w = torch.from_numpy(rng.random(nnz, dtype=np.float32)).to(dtype)
        hi = oracle.gather_weight_scatter(si, idx, w.float().numpy(), x.float().numpy(), acc64=True)
        mag = oracle.gather_weight_scatter(si, idx, w.float().numpy(), x.float().abs().numpy(), acc64=True)
        out = geot.gather_weight_scatter(dev(si), dev(idx), w.cuda(), x.cuda()).float().cpu().numpy()
        assert np.all(np.abs(out - hi) <= ulp * np.abs(hi) + 2e-5 * mag + 1e-6), ("gws", nnz, F)
    # sorted=False: the probe finds this index ascending -> the same atomic-free kernels (16-bit storage is fine);
    # an index with descents is reduced over its stable sort (no 16-bit float atomics needed): same multiset of
    # addends per row, fp32 accumulation, one rounding
    fwd = geot.index_scatter(0, s32.cuda(), dev(idx), "sum", sorted=True)
    assert torch.equal(geot.index_scatter(0, s32.cuda(), dev(idx), "sum", sorted=False), fwd)
    rev_idx = idx[::-1].copy()
    rev_idx[-1] = idx[-1]                                                 # the row rule reads index[-1]
    rev_src = torch.from_numpy(s32.float().numpy()[::-1].copy()).to(dtype)
    rev_src[-1] = s32[-1]
    out = geot.index_scatter(0, rev_src.cuda(), dev(rev_idx), "sum", sorted=False)
    order = np.argsort(rev_idx, kind="stable")
    hi = oracle.index_scatter(rev_idx[order], rev_src.float().numpy()[order], rows=int(idx[-1]) + 1, acc64=True)
    mag = oracle.index_scatter(rev_idx[order], np.abs(rev_src.float().numpy()[order]), rows=int(idx[-1]) + 1, acc64=True)
    assert out.dtype == dtype and np.all(np.abs(out.float().cpu().numpy() - hi) <= ulp * np.abs(hi) + 2e-5 * mag + 1e-6)


def test_calls_are_hipgraph_capturable(geot, oracle):
    """The C-ABI calls do no allocation, no host sync and no per-call memset: capture once, replay."""
    from geot_amd import hip
    rng = np.random.default_rng(8)
    index_h = powerlaw_index(200_000, 15_000, 6)
    index = dev(index_h)
    src = torch.rand(200_000, 64, device="cuda")
    out = torch.empty(15_000, 64, device="cuda")
    si = dev(rng.integers(0, 15_000, 200_000).astype(np.int64))
    w = torch.rand(200_000, device="cuda")
    x = torch.rand(15_000, 64, device="cuda")
    out2 = torch.empty(15_000, 64, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        hip.index_scatter_out(index, src, out)                       # warm-up on the capture stream (workspace)
        hip.gather_weight_scatter_out(si, index, w, x, out2)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            hip.index_scatter_out(index, src, out)
            hip.gather_weight_scatter_out(si, index, w, x, out2)
    for rep in range(3):
        src.uniform_()                                               # new inputs, same addresses
        x.uniform_()
        out.fill_(float("nan"))
        out2.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        ref = torch.zeros_like(out).index_add_(0, index, src)
        assert torch.allclose(out, ref, rtol=1e-5, atol=1e-4)
        ref2 = torch.zeros_like(out2).index_add_(0, index, x[si] * w[:, None])
        assert torch.allclose(out2, ref2, rtol=1e-5, atol=1e-4)


def test_inference_mode_and_no_grad(geot, oracle):
    rng = np.random.default_rng(50)
    index_h = sorted_index(rng, 3000, 200)
    src_h = rng.random((3000, 16), dtype=np.float32)
    hi = oracle.index_scatter(index_h, src_h, acc64=True)
    with torch.inference_mode():
        index, src = dev(index_h), dev(src_h)                       # inference tensors: no version counter
        for _ in range(3):
            assert_close_to_oracle(geot.index_scatter(0, src, index), hi, hi, "inference_mode")
        x = torch.rand(200, 16, device="cuda")
        si = torch.randint(0, 200, (3000,), device="cuda")
        assert geot.gather_scatter(si, index, x).shape == (200, 16)
    with torch.no_grad():
        assert_close_to_oracle(geot.index_scatter(0, dev(src_h), dev(index_h)), hi, hi, "no_grad")


def test_out_rows_larger_than_last_key(geot):
    """C ABI: out_rows may exceed index[-1]+1; the extra rows are zero-filled (small and large tails)."""
    from geot_amd import hip
    idx = dev(sorted_index(np.random.default_rng(3), 5000, 100))
    src = torch.rand(5000, 64, device="cuda")
    ref = geot.index_scatter(0, src, idx)
    for extra in (1, 7, 16, 17, 1000, 200_000):
        out = torch.full((100 + extra, 64), float("nan"), device="cuda")
        hip.index_scatter_out(idx, src, out)
        assert torch.equal(out[:100], ref) and out[100:].abs().sum().item() == 0, extra


def test_errors_on_gpu_tensors(geot):
    src = torch.rand(6, 4, device="cuda")
    idx = torch.tensor([0, 0, 1, 1, 2, 2], device="cuda")
    with pytest.raises(RuntimeError, match="expected scalar type Long but found Int"):
        geot.index_scatter(0, src, idx.int())
    with pytest.raises(RuntimeError, match="not implemented for 'Int'"):
        geot.index_scatter(0, (src * 10).int(), idx)
    with pytest.raises(RuntimeError, match="index length must be equal to src dimension size"):
        geot.index_scatter(0, src, idx[:4])
    with pytest.raises(RuntimeError, match="CPU tensors are not supported|same device|no CPU fallback"):
        geot.index_scatter(0, src, idx.cpu())


def test_c_abi_direct_error_codes(geot):
    """Call the C ABI with raw pointers: workspace / argument validation, no exceptions, no crash."""
    from geot_amd import _lib
    L = _lib.load()
    idx = torch.tensor([0, 0, 1], device="cuda")
    src = torch.rand(3, 4, device="cuda")
    out = torch.empty(2, 4, device="cuda")
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert L.geot_index_scatter(idx.data_ptr(), src.data_ptr(), out.data_ptr(), 3, 4, 2, 0, 1, ws.data_ptr(), 16, st) == -2
    assert b"workspace" in L.geot_last_error()
    assert L.geot_index_scatter(idx.data_ptr(), src.data_ptr(), out.data_ptr(), 3, 4, 2, 0, 1, ws.data_ptr() + 8, 1 << 19, st) == -2
    assert L.geot_index_scatter(idx.data_ptr(), src.data_ptr(), out.data_ptr(), -1, 4, 2, 0, 1, ws.data_ptr(), ws.numel(), st) == -1
    assert L.geot_index_scatter(idx.data_ptr(), src.data_ptr(), out.data_ptr(), 3, 4, 2, 7, 1, ws.data_ptr(), ws.numel(), st) == -1
    assert L.geot_index_scatter(None, src.data_ptr(), out.data_ptr(), 3, 4, 2, 0, 1, ws.data_ptr(), ws.numel(), st) == -1
    assert L.geot_index_scatter(idx.data_ptr(), src.data_ptr(), out.data_ptr(), 3, 4, 2, 0, 1, ws.data_ptr(), ws.numel(), st) == 0
    torch.cuda.synchronize()
    assert torch.allclose(out, torch.stack([src[0] + src[1], src[2]]))
    # nnz == 0: dst is zero-filled
    out.fill_(7)
    assert L.geot_index_scatter(idx.data_ptr(), src.data_ptr(), out.data_ptr(), 0, 4, 2, 0, 1, ws.data_ptr(), ws.numel(), st) == 0
    torch.cuda.synchronize()
    assert out.abs().sum().item() == 0


def test_cfg2_full_size_properties(geot):
    """BASELINE.json configs[1]: 10M power-law edges -> 1M nodes, F=64.  Size-independent properties:
    column checksums (linearity), determinism, sampled segments against torch, torch.index_add_."""
    nnz, keys, F = 10_000_000, 1_000_000, 64
    index_h = powerlaw_index(nnz, keys, 0)
    index = dev(index_h)
    torch.manual_seed(1)
    src = torch.rand(nnz, F, device="cuda")
    out = geot.index_scatter(0, src, index, "sum", True)
    assert out.shape == (keys, F)
    assert torch.equal(out, geot.index_scatter(0, src, index, "sum", True))
    # checksum of checksums: every src element lands in exactly one row
    assert torch.allclose(out.double().sum(0), src.double().sum(0), rtol=1e-9)
    # empty keys are exactly zero, non-empty ones are not
    counts = torch.bincount(index, minlength=keys)
    assert out[counts == 0].abs().sum().item() == 0 and (counts == 0).sum().item() > 0
    assert (out[counts > 0].abs().sum(1) > 0).all()
    # the hub and 2000 random segments against a float64 torch reduction
    offs = torch.cumsum(counts, 0) - counts
    pick = torch.cat([counts.argmax().view(1), torch.randint(0, keys, (2000,), device="cuda")])
    for k in pick[:200].tolist():
        seg = src[offs[k]: offs[k] + counts[k]].double().sum(0)
        assert torch.allclose(out[k].double(), seg, rtol=1e-5, atol=1e-7), k
    ref = torch.zeros(keys, F, device="cuda", dtype=torch.float64).index_add_(0, index, src.double())
    assert bool(((out.double() - ref).abs() <= 1e-5 * ref.abs() + 1e-30).all())   # float64 reference, non-negative data
