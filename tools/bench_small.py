#!/usr/bin/env python3
"""Per-call wall time of launch-bound calls: cora / citeseer / pubmed / ppi / amazon_photo-scale
graphs - five of the reference's nine benchmark datasets (benchmark/bench_index_scatter.py:91) - timed the way the
reference times them (wall clock over back-to-back calls, one synchronize at the end): the operator as dispatched
(C++ host layer: row rule read-back + allocation + kernels) beside torch.index_add_ / torch.sparse.mm, which are
handed the row count; and the same dispatched operators replayed from a hipGraph (torch.cuda.graph around 20 calls:
after one eager call they are capturable - what a small-graph model would do around its whole forward pass).  CSV on stdout.
    python tools/bench_small.py [--iters 2000]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import powerlaw_index  # noqa: E402

import geot_amd as geot  # noqa: E402
from geot_amd import hip  # noqa: E402

DATASETS = {"citeseer": (4_230, 10_674), "ppi": (1_767, 32_318), "pubmed": (19_717, 88_648), "cora": (19_793, 126_842),
            "amazon_photo": (7_650, 238_162)}


def wall_us(fn, n):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def graph_us(fn, n, per_graph=20):
    """per-call time of `fn` replayed from a captured graph of `per_graph` calls"""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(per_graph):
                fn()
    reps = max(5, n // per_graph)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * per_graph) * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2000)
    a = ap.parse_args()
    dev = torch.device("cuda")
    print(hip.build_info(), file=sys.stderr)
    print("dataset,nodes,edges,feat,index_scatter_sorted_us,index_scatter_sorted_false_us,doorway_no_row_rule_us,torch_index_add_us,"
          "gather_weight_scatter_us,torch_sparse_mm_us,index_scatter_trust2_us,gather_weight_scatter_trust2_us,"
          "index_scatter_graph_replay_us,gather_weight_scatter_graph_replay_us")
    for name, (nodes, edges) in DATASETS.items():
        nnz = edges + nodes
        dst = powerlaw_index(nnz, nodes, 3, dev)
        col = torch.randint(0, nodes, (nnz,), device=dev)
        val = torch.rand(nnz, device=dev)
        rowptr = torch.zeros(nodes + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.bincount(dst, minlength=nodes).cumsum(0)
        adj = torch.sparse_csr_tensor(rowptr, col, val, (nodes, nodes))
        for F in (1, 16, 64):
            src = torch.rand(nnz, F, device=dev)
            x = torch.rand(nodes, F, device=dev)
            out = torch.empty(nodes, F, device=dev)
            t = [wall_us(lambda: geot.index_scatter(0, src, dst, "sum", True), a.iters),
                 wall_us(lambda: geot.index_scatter(0, src, dst, "sum", False), a.iters),
                 wall_us(lambda: hip.index_scatter_out(dst, src, out), a.iters),
                 wall_us(lambda: torch.zeros(nodes, F, device=dev).index_add_(0, dst, src), a.iters),
                 wall_us(lambda: geot.gather_weight_scatter(col, dst, val, x), a.iters),
                 wall_us(lambda: torch.sparse.mm(adj, x), max(50, a.iters // 10))]
            geot.ops.set_option("trust_version", 2)      # opt-in: the remembered row count is not read back
            t += [wall_us(lambda: geot.index_scatter(0, src, dst, "sum", True), a.iters),
                  wall_us(lambda: geot.gather_weight_scatter(col, dst, val, x), a.iters)]
            geot.ops.set_option("trust_version", 1)
            t += [graph_us(lambda: geot.index_scatter(0, src, dst, "sum", True), a.iters),
                  graph_us(lambda: geot.gather_weight_scatter(col, dst, val, x), a.iters)]
            print(f"{name},{nodes},{nnz},{F}," + ",".join(f"{v:.1f}" for v in t), flush=True)


if __name__ == "__main__":
    main()
