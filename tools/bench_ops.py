#!/usr/bin/env python3
"""Timing of the gather ops on the synthetic stand-ins of BASELINE.json configs[2] / configs[3]
(ogbn-products-scale gws F=128 vs rocSPARSE CSR SpMM; Reddit-scale mh_spmm H=4 F=64), kernels only
(hipEvents via torch.cuda.Event on the current stream; the ops' .item() is outside the bracket because
the *_out doorway is used).  Developer tool: numbers are quoted in DESIGN.md, not a bench.py line.

    python tools/bench_ops.py [--scale 1.0] [--iters 20]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import powerlaw_index  # noqa: E402

import geot_amd  # noqa: E402
from geot_amd import hip  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def graph(nodes, nnz, seed, dev):
    dst = powerlaw_index(nnz, nodes, seed, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(seed + 1)
    src_index = torch.randint(0, nodes, (nnz,), device=dev, generator=g)
    return src_index, dst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda")
    print(hip.build_info())

    # ---- cfg3: gather_weight_scatter, ogbn-products scale (2.45M nodes, ~124M directed edges), F=128
    nodes, nnz, F = int(2_449_029 * args.scale), int(123_718_280 * args.scale), 128
    si, di = graph(nodes, nnz, 7, dev)
    w = torch.rand(nnz, device=dev)
    x = torch.rand(nodes, F, device=dev)
    out = torch.empty(nodes, F, device=dev)
    uniq = torch.unique(si).numel()
    comp = nnz * 20 + uniq * 4 * F + nodes * 4 * F
    for cg in (32, 64, 128, 256):
        hip.tune(edges_per_group=cg)
        tt = timeit(lambda: hip.gather_weight_scatter_out(si, di, w, x, out), max(3, args.iters // 2))
        print(f"   gws tile sweep: edges_per_group={cg}: {tt:.3f} ms")
    for un in (8, 16, 8, 16):
        hip.set_option("unroll", un)
        for cg in (64, 128):
            hip.tune(edges_per_group=cg)
            tt = timeit(lambda: hip.gather_weight_scatter_out(si, di, w, x, out), max(3, args.iters // 2))
            print(f"   gws unroll={un} edges_per_group={cg}: {tt:.3f} ms")
    hip.set_option("unroll", 0)
    hip.tune()
    for nt in (0, 1, 2, 3):
        hip.tune(nontemporal=nt)
        tt = timeit(lambda: hip.gather_weight_scatter_out(si, di, w, x, out), max(3, args.iters // 2))
        print(f"   gws nt policy {nt}: {tt:.3f} ms")
    hip.tune()
    # graph WITH locality: sources within +-2000 rows of the destination (what a reordered real graph looks like)
    g2 = torch.Generator(device=dev)
    g2.manual_seed(99)
    si_loc = (di + torch.randint(-2000, 2001, (nnz,), device=dev, generator=g2)).clamp_(0, nodes - 1)
    for xcd in (0, 1, 0, 1):
        hip.set_option("xcd", xcd)
        tl = timeit(lambda: hip.gather_weight_scatter_out(si_loc, di, w, x, out), max(3, args.iters // 2))
        tr = timeit(lambda: hip.gather_weight_scatter_out(si, di, w, x, out), max(3, args.iters // 2))
        print(f"   gws xcd_swizzle={xcd}: local graph {tl:.3f} ms ({nnz / tl / 1e6:.1f} Gedge/s) | random graph {tr:.3f} ms")
    hip.set_option("xcd", 1)
    del si_loc
    t = timeit(lambda: hip.gather_weight_scatter_out(si, di, w, x, out), args.iters)
    print(f"cfg3 gws   nodes={nodes} nnz={nnz} F={F}: {t:.3f} ms  {nnz / t / 1e6:.2f} Gedge/s  "
          f"compulsory {comp / 1e9:.2f} GB -> {comp / t / 1e9:.2f} TB/s ({comp / t / 1e9 / 8 * 100:.1f}% of 8 TB/s)")
    t2 = timeit(lambda: hip.gather_scatter_out(si, di, x, out), args.iters)
    print(f"cfg3 gs    (same graph, no weights): {t2:.3f} ms  {nnz / t2 / 1e6:.2f} Gedge/s")
    # rocSPARSE CSR SpMM through torch.sparse (hipSPARSE/rocSPARSE backend), same matrix
    crow = torch.zeros(nodes + 1, dtype=torch.int64, device=dev)
    crow[1:] = torch.cumsum(torch.bincount(di, minlength=nodes), 0)
    A = torch.sparse_csr_tensor(crow, si, w, size=(nodes, nodes))
    ref = torch.sparse.mm(A, x)
    hip.gather_weight_scatter_out(si, di, w, x, out)
    err = ((out - ref).abs().max() / ref.abs().max()).item()
    t3 = timeit(lambda: torch.sparse.mm(A, x), max(3, args.iters // 2))
    print(f"cfg3 rocSPARSE CSR SpMM (torch.sparse.mm): {t3:.3f} ms  {nnz / t3 / 1e6:.2f} Gedge/s   "
          f"-> ours is {t3 / t:.2f}x faster; max rel diff {err:.2e}")
    A32 = torch.sparse_csr_tensor(crow.int(), si.int(), w, size=(nodes, nodes))
    t4 = timeit(lambda: torch.sparse.mm(A32, x), max(3, args.iters // 2))
    print(f"cfg3 rocSPARSE CSR SpMM, int32 indices:    {t4:.3f} ms  -> ours is {t4 / t:.2f}x faster")
    # training step: forward + backward of gws through autograd (transposed edge list cached vs re-sorted)
    import geot_amd.ops as gops
    xg = x.clone().requires_grad_(True)
    wg = w.clone().requires_grad_(True)

    def train_step():
        y = geot_amd.gather_weight_scatter(si, di, wg, xg)
        y.backward(torch.ones_like(y))
        xg.grad = None
        wg.grad = None

    t_cached = timeit(train_step, 3)
    keep = gops.set_option("transpose_cache", 0)
    t_resort = timeit(train_step, 3)
    gops.set_option("transpose_cache", keep)
    print(f"cfg3 gws fwd+bwd (autograd): {t_cached:.2f} ms with the cached transposed edge list, {t_resort:.2f} ms re-sorting every call")
    del xg, wg
    # backward pieces (row f1): SDDMM (d/dweight of gws) and the row gather (backward of index_scatter)
    g = torch.rand(nodes, F, device=dev)
    dw = torch.empty(nnz, device=dev)
    si_loc = (di + torch.randint(-2000, 2001, (nnz,), device=dev)).clamp_(0, nodes - 1)
    for xcd in (0, 1):
        hip.set_option("xcd", xcd)
        ta = timeit(lambda: hip.sddmm_coo_out(si_loc, di, g, x, dw), 3)
        tb = timeit(lambda: hip.sddmm_coo_out(si, di, g, x, dw), 3)
        print(f"   sddmm xcd_swizzle={xcd}: local graph {ta:.3f} ms | random graph {tb:.3f} ms")
    hip.set_option("xcd", 1)
    del si_loc
    t5 = timeit(lambda: hip.sddmm_coo_out(si, di, g, x, dw), max(3, args.iters // 2))
    gathered = nnz * 2 * 4 * F
    print(f"cfg3 sddmm_coo F={F}: {t5:.3f} ms  {nnz / t5 / 1e6:.2f} Gedge/s  gathered {gathered / 1e9:.1f} GB -> {gathered / t5 / 1e9:.2f} TB/s")
    sub = 20_000_000
    rows_out = torch.empty(sub, F, device=dev)
    t6 = timeit(lambda: hip.gather_rows_out(di[:sub], x, rows_out), max(3, args.iters // 2))
    print(f"gather_rows {sub} x F={F} (sorted index): {t6:.3f} ms  {(sub * 4 * F * 2) / t6 / 1e9:.2f} TB/s (read+write)")
    del g, dw, rows_out
    del A, A32, ref, x, out, w, si, di, crow
    torch.cuda.empty_cache()

    # ---- cfg4: mh_spmm, Reddit scale (232,965 nodes, 114.6M edges), H=4, F=64
    nodes, nnz, H, F = int(232_965 * args.scale), int(114_615_892 * args.scale), 4, 64
    si, di = graph(nodes, nnz, 11, dev)
    w = torch.rand(nnz, H, device=dev)
    wt = w.t().contiguous()
    x = torch.rand(nodes, H, F, device=dev)
    out = torch.empty(nodes, H, F, device=dev)
    comp = nnz * (16 + 4 * H) + 2 * nodes * 4 * H * F
    t = timeit(lambda: hip.mh_spmm_out(si, di, w, x, out, False), args.iters)
    print(f"cfg4 mh_spmm [nnz,H] nodes={nodes} nnz={nnz} H={H} F={F}: {t:.3f} ms  {nnz / t / 1e6:.2f} Gedge/s  "
          f"compulsory {comp / 1e9:.2f} GB -> {comp / t / 1e9:.2f} TB/s")
    t2 = timeit(lambda: hip.mh_spmm_out(si, di, wt, x, out, True), args.iters)
    print(f"cfg4 mh_spmm [H,nnz]: {t2:.3f} ms  {nnz / t2 / 1e6:.2f} Gedge/s")
    del si, di, w, wt, x, out
    torch.cuda.empty_cache()

    # ---- cfg5: gather_scatter, papers100M scale, ONE GPU's shard of 8: 1/8 of 1.6157 B edges, 1/8 of the
    #      111.06 M dst rows, src (all 111.06 M nodes x 128) replicated = 56.9 GB
    if args.scale >= 1.0:
        nodes_all, F = 111_059_956, 128
        nnz, rows = 1_615_685_872 // 8, 111_059_956 // 8
        di = powerlaw_index(nnz, rows, 13, dev)
        g = torch.Generator(device=dev)
        g.manual_seed(14)
        si = torch.randint(0, nodes_all, (nnz,), device=dev, generator=g)
        x = torch.rand(nodes_all, F, device=dev)
        out = torch.empty(rows, F, device=dev)
        hip.gather_scatter_out(si, di, x, out)
        torch.cuda.synchronize()
        # checks at full size: a handful of rows against a float64 torch reduction, and the hub
        counts = torch.bincount(di, minlength=rows)
        offs = torch.cumsum(counts, 0) - counts
        for k in [int(counts.argmax()), 0, rows - 1, rows // 2, 12345]:
            seg = x[si[offs[k]: offs[k] + counts[k]]].double().sum(0)
            assert torch.allclose(out[k].double(), seg, rtol=1e-5, atol=1e-6), k
        assert out[counts == 0].abs().sum().item() == 0
        t = timeit(lambda: hip.gather_scatter_out(si, di, x, out), 3)
        gathered = nnz * 4 * F
        print(f"cfg5 shard (1/8): gather_scatter nnz={nnz} rows={rows} F={F}, src {x.numel() * 4 / 1e9:.1f} GB resident: "
              f"{t:.3f} ms  {nnz / t / 1e6:.2f} Gedge/s  gathered {gathered / 1e9:.0f} GB -> {gathered / t / 1e9:.2f} TB/s; "
              f"peak memory {torch.cuda.max_memory_allocated() / 1e9:.0f} GB")


if __name__ == "__main__":
    main()
