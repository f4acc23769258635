#!/usr/bin/env python3
"""Host-side cost of the sharded call path on ONE rank (a 1-rank RCCL group: every collective is there, nothing crosses a link):
`geot.index_scatter` vs `sharding.sharded_index_scatter` at the graded shape, wall time per call over a pipelined loop.  What the
multi-GPU line pays per step on top of the kernels, whatever the links do.   python tools/bench_sharded_overhead.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import FEAT, KEYS, NNZ, powerlaw_index  # noqa: E402

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
import geot_amd as geot  # noqa: E402
from geot_amd import sharding  # noqa: E402

index = powerlaw_index(NNZ, KEYS, 0, dev)
src = torch.rand(NNZ, FEAT, device=dev)


def wall(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


a = wall(lambda: geot.index_scatter(0, src, index, "sum", True))
b = wall(lambda: sharding.sharded_index_scatter(index, src, key_offset=0))
c = wall(lambda: sharding.sharded_index_scatter(index, src, key_offset=0, collective="reduce_scatter"))
print(f"graded shape, one rank: geot.index_scatter {a:.4f} ms per call; sharded_index_scatter {b:.4f} ms (all_gather form), {c:.4f} ms (reduce_scatter form)"
      f"  -> the sharded path adds {(b - a) * 1e3:.0f} us of host / queue time per step")
small_i = powerlaw_index(200_000, 20_000, 1, dev)
small_s = torch.rand(200_000, FEAT, device=dev)
a = wall(lambda: geot.index_scatter(0, small_s, small_i, "sum", True), 500)
b = wall(lambda: sharding.sharded_index_scatter(small_i, small_s, key_offset=0), 500)
print(f"200 k edges (launch-bound): geot.index_scatter {a * 1e3:.1f} us; sharded {b * 1e3:.1f} us")
dist.destroy_process_group()
