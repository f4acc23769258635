"""The DEVELOPMENT build of the library (geot_amd/libgeot_hip_dev.so: the sources with -DGEOT_DEV_EXPERIMENTS): the measured-and-rejected
variants the product does not carry, kept honest.  These tests run only in a process started with GEOT_HIP_LIB=dev
(`GEOT_HIP_LIB=dev python -m pytest tests -m gpu`: the whole suite passes on the development build too, plus this file); in a
product process the module is skipped and tests/test_abi_and_host.py checks that the product refuses the switches.
"""
import os

import numpy as np
import pytest
import torch

from conftest import powerlaw_index

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("GEOT_HIP_LIB", "") != "dev", reason="development build only (GEOT_HIP_LIB=dev)")]


@pytest.fixture(scope="module")
def geot():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import geot_amd
    assert "DEVELOPMENT" in geot_amd.hip.build_info()
    return geot_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _dense_graph(rng, nodes, nnz):
    di = powerlaw_index(nnz, nodes, nodes + 1)
    di[: nnz // 20] = di[nnz // 20]                                     # a hub that is split (carry slots)
    di = np.sort(di)
    di[di == 7] = 8                                                     # a destination without edges
    si = rng.integers(0, nodes, nnz).astype(np.int64)
    return si, di



@pytest.mark.parametrize("dtype,H,Fh", [(torch.bfloat16, 4, 64), (torch.float32, 4, 32), (torch.float16, 8, 32), (torch.float32, 2, 64)])
def test_two_rows_per_instruction_kernel_gives_the_sums(geot, dtype, H, Fh):
    """seg_slab_wpair_kernel (option slab_pair; off by default - measured slower, kept as an experiment): multi-head plans over 512-byte
    rows read two edges' rows per wave-instruction, each half of the wave adding its partial sums into the group's LDS rows in turn.
    A hub of 30 000 edges split into pieces (both halves hand pieces of the SAME row in at the same step - the case that lost 18 % of
    the hub before the wave-level fences), rows without edges, both weight layouts, weights in plan order: against float64 and against
    the one-row-per-instruction kernel."""
    from geot_amd import slab
    rng = np.random.default_rng(H * Fh + 1)
    nodes, nnz = 2500, 300_000
    si, di = _dense_graph(rng, nodes, nnz)
    di[: nnz // 10] = di[nnz // 10]
    di = np.sort(di)
    esz = 4 if dtype == torch.float32 else 2
    rowbytes = H * Fh * esz
    assert rowbytes == 512
    v = torch.from_numpy(rng.random((nodes, H, Fh), dtype=np.float32)).to(dtype).cuda()
    w = torch.from_numpy(rng.random((nnz, H), dtype=np.float32)).to(dtype).cuda()
    d_si, d_di = dev(si), dev(di)
    plan = slab.build_plan(d_si, d_di, nodes, nodes, rowbytes, 2, H, rows_per_group=slab.rows_per_group(2, H, dtype, rowbytes))
    assert plan.meta["split_rows"] >= 1
    ref = torch.zeros(nodes, H, Fh, device="cuda", dtype=torch.float64).index_add_(0, d_di, v.double()[d_si] * w.double()[:, :, None])
    tol = 1e-5 if dtype == torch.float32 else (2.0 ** -9 if dtype == torch.float16 else 2.0 ** -6)
    outs = {}
    geot.hip.set_option("slab_spmm_mfma", 0)                            # (the comparison is with the row-per-wave kernel, not the matrix-core one)
    try:
        for pair in (1, 0):
            geot.hip.set_option("slab_pair", pair)
            o = torch.full((nodes, H, Fh), float("nan"), device="cuda", dtype=dtype)
            slab.slab_spmm_out(plan, w, 2, v, o, H, Fh, stage_weights=False)
            assert ("seg_slab_wpair_kernel" if pair else "seg_slab_wrow_kernel") in geot.hip.last_kernel(), geot.hip.last_kernel()
            assert float((o.double() - ref).abs().max()) <= tol * float(ref.abs().max()), pair
            outs[pair] = o
        geot.hip.set_option("slab_pair", 1)
        again = torch.empty_like(outs[1])
        slab.slab_spmm_out(plan, w, 2, v, again, H, Fh, stage_weights=False)
        assert torch.equal(again, outs[1])                                          # fixed by the plan, not by timing
        slab.slab_spmm_out(plan, w.t().contiguous(), 3, v, again, H, Fh)            # head-major weights: the same sums
        assert torch.equal(again, outs[1])
        slab.slab_spmm_out(plan, w[plan.tensors["e_perm"].long()].contiguous(), 5, v, again, H, Fh)   # weights in plan order
        assert torch.equal(again, outs[1])
    finally:
        geot.hip.set_option("slab_pair", 0)
        geot.hip.set_option("slab_spmm_mfma", 1)
    assert float((outs[1].double() - outs[0].double()).abs().max()) <= 2 * tol * float(ref.abs().max())


def test_probe_drops_the_row_reads(geot):
    """slab_probe = 1: the gathered table's descriptor has zero records - every row read returns 0, so a weighted sum of anything is 0
    (wrong by design: a timing experiment, which is why the product does not carry it)."""
    from geot_amd import slab
    rng = np.random.default_rng(3)
    nodes, nnz, H, Fh = 2500, 300_000, 4, 64
    si, di = _dense_graph(rng, nodes, nnz)
    v = torch.rand(nodes, H, Fh, device="cuda").bfloat16() + 1
    w = torch.rand(nnz, H, device="cuda").bfloat16() + 1
    plan = slab.build_plan(dev(si), dev(di), nodes, nodes, 512, 2, H, rows_per_group=slab.rows_per_group(2, H, torch.bfloat16, 512))
    o = torch.empty(nodes, H, Fh, device="cuda", dtype=torch.bfloat16)
    try:
        geot.hip.set_option("slab_probe", 1)
        slab.slab_spmm_out(plan, w, 2, v, o, H, Fh, stage_weights=False)
        assert float(o.float().abs().max()) == 0.0
    finally:
        geot.hip.set_option("slab_probe", 0)
    slab.slab_spmm_out(plan, w, 2, v, o, H, Fh, stage_weights=False)
    assert float(o.float().abs().max()) > 0.0
