#!/usr/bin/env python3
"""Does the source-blocked path (csrc/seg_slab.hip) still pay when a DENSE graph has locality?  The density rule that routes a
graph there was calibrated on uniform-random sources.  Reddit-scale graphs (232 965 nodes; 114.6 M and 23.2 M edges) and a
1 M-node / 100 M-edge graph, sources uniform-random or within +-window rows of the destination: per-edge kernels (slab_mode
never), the plan (always), and what the operator does by itself (auto: the plan is TRIED on its first use and kept only if it is
faster - `plan_or_edges` in csrc/torch_ops.cpp).  `profiles/r03/slab_vs_per_edge_by_locality.txt`.

    python tools/bench_slab_locality.py
"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import powerlaw_index
import geot_amd as geot
from geot_amd import hip, ops
dev = torch.device("cuda")
def timeit(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
g = torch.Generator(device=dev); g.manual_seed(1)
ops.set_option("content_guard", 0)
for nodes, nnz in ((232_965, 114_615_892), (232_965, 23_213_838), (1_000_000, 100_000_000)):
    di = powerlaw_index(nnz, nodes, 11, dev)
    for window in (0, 2000, 20000, 100000):
        si = torch.randint(0, nodes, (nnz,), device=dev, generator=g) if window == 0 else (di + torch.randint(-window, window + 1, (nnz,), device=dev, generator=g)).clamp_(0, nodes - 1)
        for op, H, F in (("mh", 4, 64), ("gws", 1, 128)):
            if op == "mh":
                x = torch.rand(nodes, H, F, device=dev); w = torch.rand(nnz, H, device=dev)
                call = lambda: geot.mh_spmm(si, di, w, x)
            else:
                x = torch.rand(nodes, F, device=dev); w = torch.rand(nnz, device=dev)
                call = lambda: geot.gather_weight_scatter(si, di, w, x)
            res = {}
            for mode in ("never", "always"):
                ops.set_option("slab_mode", mode); ops.clear_caches()
                res[mode] = timeit(call)
            ops.set_option("slab_mode", "auto"); ops.clear_caches()
            call(); call(); s0 = ops.stats()["slab_calls"]; call(); auto_slab = ops.stats()["slab_calls"] > s0
            print(f"nodes={nodes} nnz={nnz} window={'random' if window == 0 else window} {op} H={H} F={F}: per-edge {res['never']:.3f} ms  slab {res['always']:.3f} ms ({res['never'] / res['always']:.2f}x)  auto takes {'slab' if auto_slab else 'per-edge'}", flush=True)
            del x, w
            ops.clear_caches(); torch.cuda.empty_cache()
