#!/usr/bin/env python3
"""Why does a 20-step bench run read 0.47 ms per kernel and a 100-step run 0.44 on the same box?  Times batches of 20 calls of the graded
operator after different kinds of preceding activity.   python tools/bench_warm_state.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import FEAT, KEYS, NNZ, powerlaw_index  # noqa: E402
import geot_amd as geot  # noqa: E402

dev = torch.device("cuda")
index = powerlaw_index(NNZ, KEYS, 0, dev)
src = torch.rand(NNZ, FEAT, device=dev)


def batch(n=20):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        geot.index_scatter(0, src, index, "sum", True)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for _ in range(5):
    geot.index_scatter(0, src, index, "sum", True)
print("after 5 warm-up calls, consecutive batches of 20 calls (ms per call):", " ".join(f"{batch():.4f}" for _ in range(8)), flush=True)
time.sleep(1.0)
print("after 1 s of idle:", " ".join(f"{batch():.4f}" for _ in range(4)), flush=True)
time.sleep(1.0)
big = torch.empty(1 << 28, device=dev)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    big.add_(1.0)
torch.cuda.synchronize()
print("after 1 s of idle + 0.3 s of a streaming torch kernel:", " ".join(f"{batch():.4f}" for _ in range(4)), flush=True)
time.sleep(1.0)
x = torch.rand(4096, 4096, device=dev)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    x = (x @ x).clamp_(-1, 1)
torch.cuda.synchronize()
print("after 1 s of idle + 0.3 s of matmuls:", " ".join(f"{batch():.4f}" for _ in range(4)), flush=True)
