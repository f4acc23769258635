#!/usr/bin/env python3
"""Round 5: the hand-off without its second launch ("single_launch" = 1, the default) against the two-launch form, alternating in
one process on one box: the graded shape (10 M power-law edges -> 1 M rows, F=64), 1 M edges, and configs[0]'s launch-bound shape
(100 k edges -> 10 k rows, F=32) eager and replayed from a hipGraph.  Device time per call (events around batches of calls) and,
for the launch-bound shape, wall time per call.  `profiles/r05/bench_single_launch.txt`

    python tools/bench_single_launch.py [--reps 6]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import powerlaw_index  # noqa: E402
import geot_amd as geot  # noqa: E402
from geot_amd import hip  # noqa: E402


def batch_ms(fn, calls):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(calls):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / calls


def wall_us(fn, calls):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / calls * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=6)
    a = ap.parse_args()
    dev = torch.device("cuda")
    print(f"# {hip.build_info()}")
    for name, nnz, keys, F, calls in (("graded: 10 M edges -> 1 M rows, F=64", 10_000_000, 1_000_000, 64, 40),
                                      ("1 M edges -> 100 k rows, F=64", 1_000_000, 100_000, 64, 200),
                                      ("configs[0]: 100 k edges -> 10 k rows, F=32", 100_000, 10_000, 32, 500)):
        index = powerlaw_index(nnz, keys, 0, dev)
        src = torch.rand(nnz, F, device=dev)
        out = torch.empty(keys, F, device=dev)
        raw = lambda: hip.index_scatter_out(index, src, out, sorted=True)          # noqa: E731  (the C ABI: kernels only)
        op = lambda: geot.index_scatter(0, src, index, "sum", True)                # noqa: E731  (the operator: + row rule, allocation)
        res = {1: [], 0: []}
        for rep in range(a.reps):
            for single in (1, 0):
                hip.set_option("single_launch", single)
                raw()
                res[single].append(batch_ms(raw, calls))
        hip.set_option("single_launch", 1)
        one, two = min(res[1]), min(res[0])
        print(f"{name:48s} C ABI, device ms per call: single launch {one * 1e3:9.2f} us   two launches {two * 1e3:9.2f} us   ({(two - one) * 1e3:+.2f} us, "
              f"{(two / one - 1) * 100:+.1f} %)   all reps single {['%.4f' % v for v in res[1]]} two {['%.4f' % v for v in res[0]]}", flush=True)
        if nnz <= 1_000_000:
            for label, fn in (("operator, eager", op),):
                w = {}
                for single in (1, 0, 1, 0):
                    hip.set_option("single_launch", single)
                    fn()
                    w.setdefault(single, []).append(wall_us(fn, calls))
                hip.set_option("single_launch", 1)
                print(f"{name:48s} {label}, wall us per call: single launch {min(w[1]):8.2f}   two launches {min(w[0]):8.2f}", flush=True)
            w = {}
            for single in (1, 0, 1, 0):
                hip.set_option("single_launch", single)
                op()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    op()
                g.replay()
                w.setdefault(single, []).append(wall_us(g.replay, calls))
                del g
            hip.set_option("single_launch", 1)
            print(f"{name:48s} operator replayed from a hipGraph, wall us per replay: single launch {min(w[1]):8.2f}   two launches {min(w[0]):8.2f}", flush=True)
        del index, src, out
    print("# done")


if __name__ == "__main__":
    main()
