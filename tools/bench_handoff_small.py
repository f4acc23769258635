#!/usr/bin/env python3
"""Does the in-kernel hand-off pay on launch-bound calls too?  index_scatter (F=16 / 64) on the small graphs of
tools/bench_small.py and on mid sizes, `handoff` = 1 vs 0, wall time per call over back-to-back calls and replayed from a graph;
plus the cost of a REPAIRED call (an index scrambled behind the version counter) at the graded size."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import powerlaw_index  # noqa: E402
import geot_amd as geot  # noqa: E402
from geot_amd import hip  # noqa: E402
from tools.bench_small import graph_us, wall_us  # noqa: E402

dev = torch.device("cuda")
print("edges,rows,feat,handoff1_us,handoff0_us,handoff1_graph_us,handoff0_graph_us")
for nodes, nnz in ((4_230, 15_000), (19_793, 147_000), (7_650, 246_000), (100_000, 1_000_000), (300_000, 3_000_000), (1_000_000, 10_000_000)):
    dst = powerlaw_index(nnz, nodes, 3, dev)
    for F in (16, 64):
        src = torch.rand(nnz, F, device=dev)
        res = []
        for mode in (1, 0):
            hip.set_option("handoff", mode)
            res.append(wall_us(lambda: geot.index_scatter(0, src, dst, "sum", True), 1000 if nnz < 2_000_000 else 200))
        for mode in (1, 0):
            hip.set_option("handoff", mode)
            res.append(graph_us(lambda: geot.index_scatter(0, src, dst, "sum", True), 400))
        hip.set_option("handoff", 1)
        print(f"{nnz},{nodes},{F}," + ",".join(f"{v:.1f}" for v in res), flush=True)
# a repaired call at the graded size
import warnings
warnings.simplefilter("ignore")
nnz, keys, F = 10_000_000, 1_000_000, 64
index = powerlaw_index(nnz, keys, 0, dev)
src = torch.rand(nnz, F, device=dev)
geot.index_scatter(0, src, index, "sum", True)
bad = index.clone()
at = torch.randint(1, nnz - 2000, (50,), device=dev)
bad[at], bad[at + 1000] = index[at + 1000], index[at]
index.data.copy_(bad)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = geot.index_scatter(0, src, index, "sum", True)
torch.cuda.synchronize()
print(f"repaired call (descents behind the version counter, 10 M edges x 64): {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
ok = torch.allclose(out.double(), torch.zeros(keys, F, device=dev, dtype=torch.float64).index_add_(0, index, src.double()), rtol=1e-5, atol=1e-5)
t0 = time.perf_counter()
out = geot.index_scatter(0, src, index, "sum", True)
torch.cuda.synchronize()
print(f"next call (alarm seen: probe + sort + gather kernels): {(time.perf_counter() - t0) * 1e3:.2f} ms; repaired result correct: {ok}", flush=True)
