#!/usr/bin/env python3
"""Kernel-level view of launch-bound calls: run under `rocprofv3 --kernel-trace --stats -- python3 tools/trace_small.py`
to see what the GPU side of a small call is made of (tile kernel, fix-up kernel, the gap between them)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import powerlaw_index  # noqa: E402

import geot_amd as geot  # noqa: E402

dev = torch.device("cuda")
geot.ops.set_option("trust_version", 2)
for name, nodes, nnz, F in (("citeseer", 4230, 14904, 16), ("cora", 19793, 146635, 64), ("amazon_photo", 7650, 245812, 64)):
    dst = powerlaw_index(nnz, nodes, 3, dev)
    col = torch.randint(0, nodes, (nnz,), device=dev)
    val = torch.rand(nnz, device=dev)
    src = torch.rand(nnz, F, device=dev)
    x = torch.rand(nodes, F, device=dev)
    for _ in range(300):
        geot.index_scatter(0, src, dst, "sum", True)
    torch.cuda.synchronize()
    for _ in range(300):
        geot.gather_weight_scatter(col, dst, val, x)
    torch.cuda.synchronize()
print("done")
