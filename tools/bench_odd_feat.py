#!/usr/bin/env python3
"""Feature widths that are not a multiple of the 16-byte vector (Reddit's raw features: F = 602 -> GraphSAGE aggregates them as they
are): per-edge gather_scatter and index_scatter at F = 600 / 602 / 604 / 608, Reddit2-sized graph (23 M edges, 233 k nodes).
    python tools/bench_odd_feat.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import device_ms, powerlaw_index  # noqa: E402
import geot_amd as geot  # noqa: E402
from geot_amd import hip, ops  # noqa: E402

dev = torch.device("cuda")
nodes, nnz = 232_965, 23_213_838
di = powerlaw_index(nnz, nodes, 11, dev)
g = torch.Generator(device=dev)
g.manual_seed(1)
si = torch.randint(0, nodes, (nnz,), device=dev, generator=g)
ops.set_option("slab_mode", "never")
print(hip.build_info())
for F in (600, 601, 602, 604, 608, 64, 66, 100, 128, 130):
    x = torch.rand(nodes, F, device=dev, generator=g)
    out = torch.empty(nodes, F, device=dev)
    ms = device_ms(lambda: hip.gather_scatter_out(si, di, x, out), 3, warmup=1)
    k = hip.last_kernel()
    ms_op = device_ms(lambda: geot.gather_scatter(si, di, x), 3, warmup=1)     # the operator: pads rows that are not whole 16-byte vectors
    n2 = 4_000_000
    src = torch.rand(n2, F, device=dev, generator=g)
    idx = di[:: nnz // n2][:n2].contiguous()
    o2 = torch.empty(int(idx[-1]) + 1, F, device=dev)
    ms2 = device_ms(lambda: hip.index_scatter_out(idx, src, o2), 5, warmup=2)
    k2 = hip.last_kernel()
    print(f"F={F:4d}: gather_scatter kernel as is {ms:7.3f} ms = {nnz * F * 4 / ms / 1e9:6.2f} TB/s of row reads [{k}]  operator {ms_op:7.3f} ms   "
          f"index_scatter 4 M edges {ms2:6.3f} ms = {n2 * F * 4 / ms2 / 1e9:5.2f} TB/s [{k2}]", flush=True)
    del x, out, src, o2
