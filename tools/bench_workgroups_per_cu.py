#!/usr/bin/env python3
"""Workgroups per CU of the tile kernel on streamed rows (`index_scatter`): the dynamic LDS a launch asks for is the only handle
on it (160 KB per CU).  The built-in rule (-1) against "what fits" (0) and against floors that leave 5 / 4 / 3 / 2 workgroups
per CU, fp32 / fp64 / bf16, rows of 128 ... 1024 bytes, sum / mean / max, 10 M and 2 M power-law edges.  Wide rows run faster
with FEWER tiles in flight (`tile_lds` in csrc/seg_reduce.hip); `profiles/r03/workgroups_per_cu.txt`.

    python tools/bench_workgroups_per_cu.py
"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import powerlaw_index
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
for nnz, keys in ((10_000_000, 1_000_000), (2_000_000, 200_000)):
    index = powerlaw_index(nnz, keys, 0, dev)
    for dt, Fs in ((torch.float32, (32, 64, 128, 256)), (torch.float64, (32, 64, 128)), (torch.bfloat16, (128, 256, 512))):
        for F in Fs:
            src = torch.rand(nnz, F, device=dev).to(dt); out = torch.empty(keys, F, device=dev, dtype=dt)
            for red in ("sum", "mean", "max"):
                res = {}
                for rep in range(3):
                    for fl in (-1, 0, 27000, 33000, 41000, 54000):     # the rule / what fits / 5 / 4 / 3 / 2 workgroups per CU
                        hip.set_option("lds_floor", fl)
                        res[fl] = min(res.get(fl, 9e9), timeit(lambda: hip.index_scatter_out(index, src, out, True, red)))
                hip.set_option("lds_floor", -1)
                best = min(res, key=res.get)
                print(f"nnz={nnz} {str(dt)[6:]:8s} F={F:3d} ({F * src.element_size():4d} B) {red:4s}: " + " ".join(f"{k}:{v:.4f}" for k, v in res.items()) + f"  best {best}: rule {res[-1] / res[best]:.3f}x of it, what fits {res[0] / res[best]:.3f}x", flush=True)
            del src, out
