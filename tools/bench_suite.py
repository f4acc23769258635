#!/usr/bin/env python3
"""Operator benchmark in the shape of the reference's benchmark/bench_index_scatter.py and
benchmark/bench_spmm.py: per (dataset, feature size) one CSV row with the engine beside the stock PyTorch
formulations, timed the way the reference times them (wall clock over `iter` back-to-back calls, one
synchronize at the end - host overhead included).

No network here, so each dataset of the reference's list (benchmark/bench_index_scatter.py:91,
benchmark/utils.py:17-46) is replaced by a synthetic graph with its published node / edge counts
(+ self loops, as utils.py:49 adds them), power-law destination degrees, uniform sources.  torch_scatter /
torch_sparse (PyG) are not installed in this image and there is no Triton: their columns are left out.  The columns
`geot_via_launch_*` are THIS engine called through the reference's comparator launcher surface
(geot.triton.launch_* -> geot_amd/comparators.py): they measure that call path (accumulate into `output`), they are
NOT a comparison against Triton kernels.

    python tools/bench_suite.py [--out-dir gpurun_out] [--iters 100] [--quick]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import powerlaw_index  # noqa: E402

import geot  # noqa: E402
from geot.triton import (launch_parallel_reduction, launch_pr_spmm, launch_serial_reduction,  # noqa: E402
                         launch_sr_spmm, launch_torch_compile_spmm)

# (nodes, directed edges) as published for the PyG / OGB objects the reference loads
DATASETS = {
    "cora": (19_793, 126_842), "citeseer": (4_230, 10_674), "pubmed": (19_717, 88_648),
    "amazon_photo": (7_650, 238_162), "ppi": (1_767, 32_318), "flickr": (716_847, 13_954_819),
    "ogbn-arxiv": (169_343, 1_166_243), "ogbl-collab": (235_868, 2_358_104), "reddit2": (232_965, 23_213_838),
}


def timeit(func, iters, *args):
    start = time.time()
    for _ in range(iters):
        func(*args)
    torch.cuda.synchronize()
    return (time.time() - start) / iters * 1000


def torch_scatter_reduce(index, src, keys):
    return torch.zeros(keys, src.size(1), device=src.device).scatter_add_(0, index.unsqueeze(-1).expand_as(src), src)


def torch_index_reduce(index, src, keys):
    return torch.zeros(keys, src.size(1), device=src.device).index_add_(0, index, src)


def index_scatter_unsorted(index, src, keys):
    return geot.index_scatter(0, src, index, reduce="sum", sorted=False)   # the flag the reference's bench passes


def index_scatter_sorted(index, src, keys):
    return geot.index_scatter(0, src, index, reduce="sum", sorted=True)


def synth(name, dev):
    nodes, edges = DATASETS[name]
    nnz = edges + nodes  # self loops
    dst = powerlaw_index(nnz, nodes, hash(name) % 1000, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(1 + hash(name) % 1000)
    col = torch.randint(0, nodes, (nnz,), device=dev, generator=g)
    return nodes, dst, col


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out-dir", default=os.path.join(ROOT, "gpurun_out"))
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    os.makedirs(a.out_dir, exist_ok=True)
    names = ["cora", "ogbn-arxiv"] if a.quick else list(DATASETS)
    feats = [1, 32] if a.quick else [1, 2, 4, 8, 16, 32, 64, 128]
    f1 = open(os.path.join(a.out_dir, "benchop_index_scatter.csv"), "w")
    f2 = open(os.path.join(a.out_dir, "benchop_spmm.csv"), "w")
    f1.write("dataset,feature_size,torch_scatter_reduce,torch_index_reduce,index_scatter_reduce,"
             "index_scatter_sorted,geot_via_launch_parallel_reduction,geot_via_launch_serial_reduction\n")
    f2.write("dataset,feature_size,gather_weight_scatter,pytorch_spmm,geot_via_launch_pr_spmm,geot_via_launch_sr_spmm,"
             "geot_via_launch_torch_compile_spmm\n")
    for name in names:
        nodes, dst, col = synth(name, dev)
        nnz = dst.numel()
        value = torch.ones(nnz, device=dev)
        rowptr = torch.zeros(nodes + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.bincount(dst, minlength=nodes).cumsum(0)
        adj = torch.sparse_csr_tensor(rowptr, col, value, (nodes, nodes))
        edges = torch.stack([col, dst]).contiguous()
        iters = max(5, a.iters // (1 + nnz // 4_000_000))
        for F in feats:
            src = torch.rand(nnz, F, device=dev)
            x = torch.rand(nodes, F, device=dev)
            o1 = torch.zeros(nodes, F, device=dev)
            o2 = torch.zeros(nodes, F, device=dev)
            ref = torch_index_reduce(dst, src, nodes)
            for fn in (index_scatter_unsorted, index_scatter_sorted):
                err = (fn(dst, src, nodes) - ref).abs().max().item()
                assert err <= 1e-3 * max(1.0, ref.abs().max().item()), (name, F, fn.__name__, err)
            cells = []
            for fn in (torch_scatter_reduce, torch_index_reduce, index_scatter_unsorted, index_scatter_sorted):
                for _ in range(3):
                    fn(dst, src, nodes)
                cells.append(timeit(fn, iters, dst, src, nodes))
            for fn, o in ((launch_parallel_reduction, o1), (launch_serial_reduction, o2)):
                for _ in range(3):
                    fn(dst, src, o, nnz, F, 32)
                cells.append(timeit(fn, iters, dst, src, o, nnz, F, 32))
            f1.write(f"{name},{F}," + ",".join(f"{c:.4f}" for c in cells) + "\n")
            f1.flush()

            ref = torch.sparse.mm(adj, x)
            got = geot.gather_weight_scatter(col, dst, value, x)
            err = (got - ref).abs().max().item()
            assert err <= 1e-3 * max(1.0, ref.abs().max().item()), (name, F, "gws", err)
            cells = []
            for _ in range(3):
                geot.gather_weight_scatter(col, dst, value, x)
                torch.sparse.mm(adj, x)
            cells.append(timeit(geot.gather_weight_scatter, iters, col, dst, value, x))
            cells.append(timeit(torch.sparse.mm, iters, adj, x))
            for fn in (launch_pr_spmm, launch_sr_spmm, launch_torch_compile_spmm):
                for _ in range(3):
                    fn(edges, x, o1, nnz, F, 32)
                cells.append(timeit(fn, iters, edges, x, o1, nnz, F, 32))
            f2.write(f"{name},{F}," + ",".join(f"{c:.4f}" for c in cells) + "\n")
            f2.flush()
            print(name, F, "done", flush=True)
            del src, x, o1, o2
    f1.close()
    f2.close()
    import pandas as pd
    for fn, base, ours in (("benchop_index_scatter.csv", "torch_index_reduce", "index_scatter_sorted"),
                           ("benchop_spmm.csv", "pytorch_spmm", "gather_weight_scatter")):
        d = pd.read_csv(os.path.join(a.out_dir, fn))
        sp = d[base] / d[ours]
        print(f"{fn}: {ours} vs {base}: geomean speedup {sp.apply('log').mean().__rpow__(2.718281828):.2f}x "
              f"(min {sp.min():.2f}x, max {sp.max():.2f}x) over {len(d)} cells")


if __name__ == "__main__":
    main()
