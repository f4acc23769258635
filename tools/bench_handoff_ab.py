#!/usr/bin/env python3
"""A/B of the in-kernel hand-off by row width: hip.index_scatter_out (tile kernel + second launch, HIP events, best of 2 x 60 calls),
`handoff` option 1 vs 0 alternating on one box, F = 16..256 x 1-10 M edges, fp32 / bf16 (profiles/r03/handoff_ab_by_row_width.txt)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import powerlaw_index, device_ms
import geot_amd as geot
from geot_amd import hip
dev = torch.device("cuda")
print("edges,feat,dtype,handoff1_us,handoff0_us,ratio")
for nnz, keys in ((1_000_000, 100_000), (4_000_000, 400_000), (10_000_000, 1_000_000), (10_000_000, 20_000)):
    idx = powerlaw_index(nnz, keys, 1, dev)
    for F, dt in ((16, torch.float32), (32, torch.float32), (64, torch.float32), (128, torch.float32), (256, torch.float32), (64, torch.bfloat16), (128, torch.bfloat16)):
        src = torch.rand(nnz, F, device=dev).to(dt)
        out = torch.empty(keys, F, device=dev, dtype=dt)
        r = {}
        for rep in range(2):
            for mode in (1, 0):
                hip.set_option("handoff", mode)
                t = device_ms(lambda: hip.index_scatter_out(idx, src, out), 60, warmup=5) * 1e3
                r[mode] = min(r.get(mode, 1e9), t)
        hip.set_option("handoff", 1)
        print(f"{nnz},{keys},{F},{str(dt)[6:]},{r[1]:.1f},{r[0]:.1f},{r[1]/r[0]:.3f}", flush=True)
