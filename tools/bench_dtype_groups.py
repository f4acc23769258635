#!/usr/bin/env python3
"""Group size of the tile kernel for the dtypes the fp32 sweep (tools/sweep_rule.py) does not cover: bf16 rows of 128 ... 1024
bytes and fp64 rows of 256 ... 1024 bytes, streamed (`index_scatter` sum), runs of 2 / 10 / 50 edges, 10 M and 2 M edges: the
built-in rule (auto) against forced 16 ... 128-edge groups.  The measurement behind make_plan's clauses for 512-byte and 1-KiB
rows of every dtype; `profiles/r03/dtype_groups.txt`.

    python tools/bench_dtype_groups.py
"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geot_amd import hip
dev = torch.device("cuda")
def timeit(fn, iters=15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
g = torch.Generator(device=dev); g.manual_seed(1)
for nnz in (10_000_000, 2_000_000):
    for avg in (2, 10, 50):
        keys = int(nnz / avg)
        index = torch.sort(torch.randint(0, keys, (nnz,), device=dev, generator=g)).values
        index[-1] = keys - 1
        for dt, Fs in ((torch.bfloat16, (64, 128, 256, 512)), (torch.float64, (32, 64, 128))):
            for F in Fs:
                src = torch.rand(nnz, F, device=dev).to(dt); out = torch.empty(keys, F, device=dev, dtype=dt)
                res = {}
                for rep in range(2):
                    for cg in (0, 16, 32, 64, 128):
                        hip.tune(cg, 0, -1, -1)
                        t = timeit(lambda: hip.index_scatter_out(index, src, out, True, "sum"))
                        res[cg] = min(res.get(cg, 9e9), t)
                hip.tune(0, 0, -1, -1)
                best = min(res, key=res.get)
                es = src.element_size(); alg = nnz * (es * F + 8) + keys * es * F
                print(f"nnz={nnz} avg={avg} {str(dt)[6:]:8s} F={F:3d} ({es*F:4d} B): auto {res[0]:.4f} ({alg/res[0]/1e9:.2f} TB/s)  best cg{best} {res[best]:.4f} ({res[0]/res[best]:.3f}x)   " + " ".join(f"{c}:{t:.4f}" for c, t in res.items()), flush=True)
                del src, out
