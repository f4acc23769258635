#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ (run in THE BUILD CONTAINER, where
/root/reference exists; the outputs are committed, this script documents how they were made).

    python tests/golden/make_golden.py

Sources of truth captured here (nothing below is our own kernel or our own oracle):
  * ``ref_out``      the REFERENCE's own CPU index_scatter, compiled in place into oracle/_ref
                     (csrc/cpu/index_scatter_cpu.cpp), run on (index, src) AS SHIPPED -- i.e. with its
                     src[index[n]] operand quirk.  Parity identity:
                     ref_out == sequential_sum(index, src[index]).
  * ``torch_*``      the comparators the reference's own tests use: zeros.index_add_/scatter_add_
                     (test/test_index_scatter.py:17-23), torch.sparse.mm on the coalesced COO matrix
                     (test/test_gather_scatter.py:4-12, test/test_gather_weight_scatter.py:4-11),
                     index_select * weight -> index_add (test/test_mh_spmm.py:4-10).
  * ``pyref_*``      the reference's Python autograd wrappers geot/gather_scatter.py and
                     geot/gather_weight_scatter.py, imported standalone from /root/reference with
                     probe CPU kernels registered for the geot::*_impl ops (torch index_add_), to
                     capture the forward result and the gradients the reference's backward formulas
                     produce.  Only ARRAYS are stored; no reference source travels.
Inputs are drawn from numpy's PCG64 with the seeds written in each case so tests can regenerate
the large ones (cfg1) instead of storing them.
"""
from __future__ import annotations

import hashlib
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref as oref  # noqa: E402  (compiled reference; test infrastructure)

REFERENCE = "/root/reference"


def sorted_index(rng, nnz, keys, force_last=True):
    idx = np.sort(rng.integers(0, keys, nnz)).astype(np.int64)
    if force_last and nnz:
        idx[-1] = keys - 1
    return idx


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def torch_index_add(index, src, rows):
    out = torch.zeros((rows,) + tuple(src.shape[1:]), dtype=torch.from_numpy(src).dtype)
    return out.index_add_(0, torch.from_numpy(index), torch.from_numpy(src)).numpy()


def torch_scatter_add(index, src, rows):
    t = torch.from_numpy(src)
    out = torch.zeros((rows, src.shape[1]), dtype=t.dtype)
    return out.scatter_add_(0, torch.from_numpy(index).unsqueeze(-1).expand_as(t), t).numpy()


def torch_spmm(src_index, dst_index, weight, src):
    # test/test_gather_weight_scatter.py:4-11 (weight None -> ones, test_gather_scatter.py:9)
    n = int(dst_index[-1]) + 1
    ncol = src.shape[0]
    w = torch.ones(len(dst_index)) if weight is None else torch.from_numpy(weight)
    adj = torch.sparse_coo_tensor(torch.stack([torch.from_numpy(dst_index), torch.from_numpy(src_index)]), w,
                                  (n, ncol)).coalesce()
    return torch.sparse.mm(adj, torch.from_numpy(src)).numpy()


def torch_mh(src_index, dst_index, weight, src):
    # test/test_mh_spmm.py:4-10, with rows = dst_index[-1]+1 instead of zeros_like(src)
    s = torch.from_numpy(src)
    sel = s.index_select(0, torch.from_numpy(src_index))
    mul = torch.from_numpy(weight).unsqueeze(-1) * sel
    out = torch.zeros((int(dst_index[-1]) + 1,) + tuple(src.shape[1:]))
    return out.index_add(0, torch.from_numpy(dst_index), mul).numpy()


def index_scatter_cases():
    cases = {}

    def add(name, index, src, store_inputs=True, **extra):
        rows = int(index[-1]) + 1
        d = dict(rows=np.int64(rows), **extra)
        if store_inputs:
            d.update(index=index, src=src)
        d["torch_index_add"] = torch_index_add(index, src, rows)
        if src.ndim == 2 and src.dtype == np.float32:
            d["torch_scatter_add"] = torch_scatter_add(index, src, rows)
        if index.max() < len(index):  # the shipped kernel reads src[index[n]]: must be in range
            d["ref_out"] = oref.index_scatter_cpu(index, src)
        cases[name] = d

    rng = np.random.default_rng(100)
    # the reference's own test shape: 1000 x 32 -> 10 keys (test/test_index_scatter.py:6-13)
    add("ref_test_1000x32_10", sorted_index(rng, 1000, 10), rng.random((1000, 32), dtype=np.float32))
    for F in (1, 3, 7, 8, 31, 33, 64, 100):
        add(f"uniform_F{F}", sorted_index(rng, 600, 90), rng.random((600, F), dtype=np.float32))
    add("single_segment", np.zeros(700, dtype=np.int64), rng.random((700, 16), dtype=np.float32))
    add("unit_segments", np.arange(300, dtype=np.int64), rng.random((300, 8), dtype=np.float32))
    add("gaps_small", np.arange(200, dtype=np.int64) * 3, rng.random((200, 8), dtype=np.float32))
    add("gaps_large", np.arange(60, dtype=np.int64) * 40, rng.random((60, 4), dtype=np.float32))
    idx = sorted_index(rng, 500, 40, force_last=False) + 25
    add("first_key_gt0", idx, rng.random((500, 8), dtype=np.float32))
    add("nnz1", np.array([3], dtype=np.int64), rng.random((1, 5), dtype=np.float32))
    hub = np.sort(np.concatenate([np.full(5000, 7), rng.integers(0, 20, 600)])).astype(np.int64)
    add("hub_5000", hub, rng.random((len(hub), 8), dtype=np.float32))
    add("fp64", sorted_index(rng, 400, 33), rng.random((400, 6)))
    add("src3d", sorted_index(rng, 120, 9), rng.random((120, 3, 4), dtype=np.float32))
    # mixed-sign data (cancellation), normal distribution
    add("signed", sorted_index(rng, 800, 50), rng.standard_normal((800, 16)).astype(np.float32))

    # 16-bit storage types of the reference CPU path (fp32 accumulate, one rounding at the end);
    # bfloat16 is stored as raw uint16 bit patterns
    idx16 = sorted_index(rng, 700, 60)
    s16 = rng.standard_normal((700, 24)).astype(np.float32)
    h = s16.astype(np.float16)
    cases["f16"] = dict(index=idx16, src=h, rows=np.int64(60), ref_out=oref.index_scatter_cpu(idx16, h))
    b = torch.from_numpy(s16).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    cases["bf16_bits"] = dict(index=idx16, src=b, rows=np.int64(60), ref_out=oref.index_scatter_cpu(idx16, b))

    # cfg1 (BASELINE.json configs[0]): 100k x 32 -> 10k segments; inputs regenerated from the seed
    rng1 = np.random.default_rng(0)
    index = sorted_index(rng1, 100_000, 10_000)
    src = np.random.default_rng(1).random((100_000, 32), dtype=np.float32)
    ref_out = oref.index_scatter_cpu(index, src)
    expect = torch_index_add(index, src, 10_000)
    cases["cfg1_100k_x32_10k"] = dict(
        rows=np.int64(10_000), seed_index=np.int64(0), seed_src=np.int64(1),
        index_sha256=np.array(sha(index)), src_sha256=np.array(sha(src)),
        ref_out_sha256=np.array(sha(ref_out)), ref_out_head=ref_out[:64],
        torch_index_add_sha256=np.array(sha(expect)), torch_index_add_head=expect[:64],
        torch_index_add_rowsum=expect.astype(np.float64).sum(axis=1))
    return cases


def reduce_cases():
    """All five reductions of the reference CPU path (csrc/cpu/index_scatter_cpu.cpp:124-134)."""
    rng = np.random.default_rng(200)
    index = sorted_index(rng, 500, 60)
    index[index == 17] = 18  # leave key 17 empty
    src = (rng.random((500, 6), dtype=np.float32) + 0.5)
    out = dict(index=index, src=src)
    for red in ("sum", "mean", "min", "max", "prod"):
        out[f"ref_{red}"] = oref.index_scatter_cpu(index, src, reduce=red)
    # NaN propagation of min/max (ATen _min/_max): NaNs early, in the middle and at the end of segments,
    # F = 21 so that both the vectorised body and the scalar tail of ATen's map2 are exercised
    index2 = sorted_index(rng, 300, 25)
    src2 = rng.standard_normal((300, 21)).astype(np.float32)
    for pos in (0, 7, 150, 151, 298, 299):
        src2[pos, rng.integers(0, 21, 4)] = np.nan
    nan = dict(index=index2, src=src2)
    for red in ("min", "max", "sum"):
        nan[f"ref_{red}"] = oref.index_scatter_cpu(index2, src2, reduce=red)
    return {"reductions": out, "reductions_nan": nan}


def gather_cases():
    cases = {}
    rng = np.random.default_rng(300)
    for name, nodes, nnz, F in (("ref_test_100n_1000e_F32", 100, 1000, 32), ("small_F5", 37, 400, 5),
                                ("fat_F128", 64, 900, 128)):
        src_index = rng.integers(0, nodes, nnz).astype(np.int64)
        dst_index = np.sort(rng.integers(0, nodes, nnz)).astype(np.int64)
        weight = rng.random(nnz, dtype=np.float32)
        src = rng.random((nodes, F), dtype=np.float32)
        cases[name] = dict(src_index=src_index, dst_index=dst_index, weight=weight, src=src,
                           torch_spmm_unweighted=torch_spmm(src_index, dst_index, None, src),
                           torch_spmm_weighted=torch_spmm(src_index, dst_index, weight, src))
    # multi-head (test/test_mh_spmm.py:12-28): 100 nodes, 1000 edges, H=4, F=32
    for name, nodes, nnz, H, F in (("mh_ref_test_H4_F32", 100, 1000, 4, 32), ("mh_H3_F6", 40, 500, 3, 6)):
        src_index = rng.integers(0, nodes, nnz).astype(np.int64)
        dst_index = np.sort(rng.integers(0, nodes, nnz)).astype(np.int64)
        weight = rng.random((nnz, H), dtype=np.float32)
        src = rng.random((nodes, H, F), dtype=np.float32)
        cases[name] = dict(src_index=src_index, dst_index=dst_index, weight=weight, src=src,
                           torch_mh=torch_mh(src_index, dst_index, weight, src))
    return cases


def pyref_autograd_cases():
    """Run the reference's Python wrappers (forward + backward formulas) on CPU probe kernels."""
    lib = torch.library.Library("geot", "FRAGMENT")
    lib.define("gather_scatter_impl(Tensor src_index, Tensor dst_index, Tensor src) -> Tensor")
    lib.define("gather_weight_scatter_impl(Tensor src_index, Tensor dst_index, Tensor weight, Tensor src) -> Tensor")
    lib.define("sddmm_coo_impl(Tensor src_index, Tensor dst_index, Tensor mat_1, Tensor mat_2) -> Tensor")

    def gs(si, di, s):
        return torch.zeros((int(di[-1]) + 1, s.shape[1]), dtype=s.dtype).index_add_(0, di, s[si])

    def gws(si, di, w, s):
        return torch.zeros((int(di[-1]) + 1, s.shape[1]), dtype=s.dtype).index_add_(0, di, s[si] * w[:, None])

    def sddmm(si, di, m1, m2):  # csrc/cuda/gather_weight_scatter_cuda.cu:41-62: rows=dst_index, cols=src_index
        return (m1[di.long()] * m2[si.long()]).sum(-1)

    lib.impl("gather_scatter_impl", gs, "CPU")
    lib.impl("gather_weight_scatter_impl", gws, "CPU")
    lib.impl("sddmm_coo_impl", sddmm, "CPU")

    def load(name):
        spec = importlib.util.spec_from_file_location(f"_pyref_{name}", os.path.join(REFERENCE, "geot", f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    m_gs = load("gather_scatter")
    m_gws = load("gather_weight_scatter")
    rng = np.random.default_rng(400)
    nodes, nnz, F = 30, 400, 8
    src_index = rng.integers(0, nodes, nnz).astype(np.int64)
    src_index[0] = nodes - 1   # make sure the last node has an out-edge (reference row-count rule)
    dst_index = np.sort(rng.integers(0, nodes, nnz)).astype(np.int64)
    dst_index[-1] = nodes - 1
    weight = rng.random(nnz, dtype=np.float32)
    src = rng.random((nodes, F), dtype=np.float32)
    gout = rng.random((nodes, F), dtype=np.float32)
    si, di = torch.from_numpy(src_index), torch.from_numpy(dst_index)

    s = torch.from_numpy(src).clone().requires_grad_(True)
    out = m_gs.gather_scatter(si, di, s)
    out.backward(torch.from_numpy(gout))
    gs_fwd, gs_dsrc = out.detach().numpy(), s.grad.numpy()

    s = torch.from_numpy(src).clone().requires_grad_(True)
    w = torch.from_numpy(weight).clone().requires_grad_(True)
    out = m_gws.gather_weight_scatter(si, di, w, s)
    out.backward(torch.from_numpy(gout))
    gws_fwd, gws_dsrc, gws_dw_shipped = out.detach().numpy(), s.grad.numpy(), w.grad.numpy()

    # dense autograd (the mathematically correct gradients)
    s = torch.from_numpy(src).clone().requires_grad_(True)
    w = torch.from_numpy(weight).clone().requires_grad_(True)
    dense = torch.zeros((nodes, F)).index_add(0, di, s[si] * w[:, None])
    dense.backward(torch.from_numpy(gout))
    return {"pyref_autograd": dict(
        src_index=src_index, dst_index=dst_index, weight=weight, src=src, grad_out=gout,
        pyref_gs_fwd=gs_fwd, pyref_gs_dsrc=gs_dsrc, pyref_gws_fwd=gws_fwd, pyref_gws_dsrc=gws_dsrc,
        pyref_gws_dweight_as_shipped=gws_dw_shipped,
        dense_gws_dsrc=s.grad.numpy(), dense_gws_dweight=w.grad.numpy(),
        schema_gather_scatter=np.array(str(torch.ops.geot.gather_scatter.default._schema)),
        schema_gather_weight_scatter=np.array(str(torch.ops.geot.gather_weight_scatter.default._schema)))}


def main():
    if not oref.available():
        if not oref.build():
            sys.exit("oracle/_ref is not built and /root/reference is absent")
    groups = {
        "index_scatter.npz": index_scatter_cases(),
        "reductions.npz": reduce_cases(),
        "gather.npz": gather_cases(),
        "pyref_autograd.npz": pyref_autograd_cases(),
    }
    for fname, cases in groups.items():
        flat = {}
        for cname, arrays in cases.items():
            for k, v in arrays.items():
                flat[f"{cname}/{k}"] = np.asarray(v)
        path = os.path.join(HERE, fname)
        np.savez_compressed(path, **flat)
        print(f"{fname}: {len(cases)} cases, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
