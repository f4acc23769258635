#!/usr/bin/env python3
"""Group size of the tile kernel on SHORT runs (average run of 1.5 ... 10 edges, uniform keys), streamed fp32 rows: the built-in
rule (auto) against forced 16- and 32-edge groups, 10 M and 2 M edges, F = 32 ... 256.  The measurement behind the short-run
clause of make_plan (csrc/seg_reduce.hip); `profiles/r03/short_run_groups.txt`.

    python tools/bench_short_runs.py
"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geot_amd import hip
dev = torch.device("cuda")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
g = torch.Generator(device=dev); g.manual_seed(1)
for nnz in (10_000_000, 2_000_000):
    for avg in (1.5, 2, 3, 4, 6, 10):
        keys = int(nnz / avg)
        index = torch.sort(torch.randint(0, keys, (nnz,), device=dev, generator=g)).values
        index[-1] = keys - 1
        for F in (32, 64, 128, 256):
            src = torch.rand(nnz, F, device=dev); out = torch.empty(keys, F, device=dev)
            res = {}
            for rep in range(2):
                for cg in (0, 16, 32):
                    hip.tune(cg, 0, -1, -1)
                    t = timeit(lambda: hip.index_scatter_out(index, src, out, True, "sum"))
                    res[cg] = min(res.get(cg, 9e9), t)
            hip.tune(0, 0, -1, -1)
            print(f"nnz={nnz} avg={avg} F={F}: auto {res[0]:.4f}  cg16 {res[16]:.4f} ({res[0] / res[16]:.3f}x)  cg32 {res[32]:.4f} ({res[0] / res[32]:.3f}x)", flush=True)
            del src, out
