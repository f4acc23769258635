#!/usr/bin/env python3
"""Lanes per row of the per-edge SDDMM kernel (sddmm_coo_kernel; the backward of gather_weight_scatter w.r.t. the edge weights):
the built-in rule (-1) against one lane per 16-byte piece (0, the gather kernels' layout) and against half / a quarter / an eighth
of those lanes (1, 2, 3), fp32 and bf16, F = 16 ... 256, uniform-random and local sources, 20 M and 2 M edges.
`profiles/r03/sddmm_lanes.txt`.

    python tools/bench_sddmm_lanes.py
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
        si = (di + torch.randint(-2000, 2001, (nnz,), device=dev, generator=g)).clamp_(0, nodes - 1) if local else torch.randint(0, nodes, (nnz,), device=dev, generator=g)
        for dt in (torch.float32, torch.bfloat16):
            for F in (16, 32, 64, 128, 256):
                x = torch.rand(nodes, F, device=dev).to(dt); y = torch.rand(nodes, F, device=dev).to(dt)
                oe = torch.empty(nnz, device=dev, dtype=dt); ref = None
                res = {}
                for rep in range(2):
                    for sh in (-1, 0, 1, 2, 3):
                        hip.set_option("sddmm_shift", sh)
                        res[sh] = min(res.get(sh, 9e9), timeit(lambda: hip.sddmm_coo_out(si, di, x, y, oe)))
                        if ref is None: ref = oe.float().clone()
                        else: assert torch.allclose(oe.float(), ref, rtol=2e-2 if dt != torch.float32 else 1e-4, atol=1e-3), (sh, (oe.float() - ref).abs().max())
                hip.set_option("sddmm_shift", -1)
                best = min(res, key=res.get)
                print(f"nnz={nnz} nodes={nodes} {'local' if local else 'random'} {str(dt)[6:]:8s} F={F:3d}: " + " ".join(f"{c}:{t:.4f}" for c, t in res.items()) + f"  best shift {best}: rule {res[-1] / res[best]:.3f}x of it, one lane per piece {res[0] / res[best]:.2f}x", flush=True)
                del x, y, oe
