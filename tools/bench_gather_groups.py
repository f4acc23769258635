#!/usr/bin/env python3
"""Tile shape of the GATHER modes (gather_weight_scatter fp32 / bf16 at F = 64 ... 512, mh_spmm at several (H, F)), on
uniform-random sources and on sources within +-2000 rows of the destination, 20 M and 2 M edges: the built-in rule (auto) against
forced 16 ... 128-edge groups.  The measurement behind make_plan's three bounds for gathered rows (L2 footprint, per-edge LDS,
grid); `profiles/r03/gather_groups.txt`.

    python tools/bench_gather_groups.py
"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geot_amd import hip
dev = torch.device("cuda")
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
g = torch.Generator(device=dev); g.manual_seed(1)
for nnz, nodes in ((20_000_000, 2_000_000), (20_000_000, 400_000), (2_000_000, 200_000)):
    di = torch.sort(torch.randint(0, nodes, (nnz,), device=dev, generator=g)).values
    di[-1] = nodes - 1
    for local in (False, True):
        if local:
            si = (di + torch.randint(-2000, 2001, (nnz,), device=dev, generator=g)).clamp_(0, nodes - 1)
        else:
            si = torch.randint(0, nodes, (nnz,), device=dev, generator=g)
        cases = []
        for dt, Fs in ((torch.float32, (64, 128, 256)), (torch.bfloat16, (64, 128, 256, 512))):
            for F in Fs:
                cases.append(("gws", dt, 1, F))
        cases += [("mh", torch.float32, 4, 16), ("mh", torch.float32, 4, 64), ("mh", torch.float32, 8, 32), ("mh", torch.bfloat16, 4, 64)]
        for op, dt, H, F in cases:
            if op == "gws":
                x = torch.rand(nodes, F, device=dev).to(dt); w = torch.rand(nnz, device=dev).to(dt); out = torch.empty(nodes, F, device=dev, dtype=dt)
                fn = lambda: hip.gather_weight_scatter_out(si, di, w, x, out)
            else:
                x = torch.rand(nodes, H, F, device=dev).to(dt); w = torch.rand(nnz, H, device=dev).to(dt); out = torch.empty(nodes, H, F, device=dev, dtype=dt)
                fn = lambda: hip.mh_spmm_out(si, di, w, x, out, False)
            res = {}
            for rep in range(2):
                for cg in (0, 16, 32, 64, 128):
                    hip.tune(cg, 0, -1, -1)
                    res[cg] = min(res.get(cg, 9e9), timeit(fn))
            hip.tune(0, 0, -1, -1)
            best = min(res, key=res.get)
            print(f"nnz={nnz} nodes={nodes} {'local' if local else 'random'} {op} {str(dt)[6:]:8s} H={H} F={F:3d}: auto {res[0]:.4f}  best cg{best} {res[best]:.4f} ({res[0]/res[best]:.3f}x)   " + " ".join(f"{c}:{t:.4f}" for c, t in res.items()), flush=True)
            del x, w, out
