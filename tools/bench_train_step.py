#!/usr/bin/env python3
"""A gather_weight_scatter training step (forward + d/dsrc + d/dweight through autograd) on a DENSE graph - Reddit scale, F=128 -
as dispatched: forward over the source-blocked plan, d/dsrc over the plan of the TRANSPOSED list (sorted by source; its
destinations ascend inside every row, as in any list made by a stable sort), d/dweight by the SDDMM over the forward's plan.

    python tools/bench_train_step.py [--nodes 232965 --edges 114615892 --feat 128]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import powerlaw_index  # noqa: E402
import geot_amd as geot  # noqa: E402
from geot_amd import hip, ops  # noqa: E402


def timed(fn, iters=5, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=232_965)
    ap.add_argument("--edges", type=int, default=114_615_892)
    ap.add_argument("--feat", type=int, default=128)
    a = ap.parse_args()
    dev = torch.device("cuda")
    print(hip.build_info())
    di = powerlaw_index(a.edges, a.nodes, 11, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(12)
    si = torch.randint(0, a.nodes, (a.edges,), device=dev, generator=g)
    for coalesced in (False, True):
        s_idx = (torch.sort(di * a.nodes + si).values % a.nodes).contiguous() if coalesced else si
        x = torch.rand(a.nodes, a.feat, device=dev, generator=g, requires_grad=True)
        w = torch.rand(a.edges, device=dev, generator=g, requires_grad=True)
        cot = torch.rand(a.nodes, a.feat, device=dev, generator=g)
        for mode in ("never", "auto"):
            ops.set_option("slab_mode", mode)
            ops.clear_caches()

            def step():
                x.grad = None
                w.grad = None
                geot.gather_weight_scatter(s_idx, di, w, x).backward(cot)
            fwd = timed(lambda: geot.gather_weight_scatter(s_idx, di, w.detach(), x.detach()))
            full = timed(step)
            st = ops.stats()
            print(f"sources {'ascending inside every row' if coalesced else 'in random order inside a row'}; slab_mode {mode:5s}: forward {fwd:7.3f} ms, "
                  f"forward + backward (d/dsrc, d/dweight) {full:7.3f} ms   plans {st['plans']} trials {st['plan_trials']} rejected {st['plans_rejected']} "
                  f"declined {st['plans_declined']}", flush=True)
        del x, w, cot
    ops.set_option("slab_mode", "auto")


if __name__ == "__main__":
    main()
