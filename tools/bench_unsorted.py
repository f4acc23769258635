#!/usr/bin/env python3
"""sorted=False on an index WITH descents: the two paths side by side.
  atomic : tile kernel, runs pre-reduced in registers, float atomics into a zeroed dst (non-deterministic)
  sort   : stable sort of the index once (kept with the index facts) -> gather-mode kernels over (keys, perm);
           deterministic, any reduction.  Timed with the sort cached (steady state) and including the sort:
           geot_sort_index (32-bit radix passes over the bits the keys use) beside ATen's generic 64-bit sort, and
           the operator as dispatched on a first sighting (caches cleared: probe + sort + reduce, wall clock).
    python tools/bench_unsorted.py [--nnz 10000000] [--keys 1000000]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import time  # noqa: E402

import geot_amd as geot  # noqa: E402
from geot_amd import hip, ops  # noqa: E402
from tools.rocsparse import device_ms  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nnz", type=int, default=10_000_000)
    ap.add_argument("--keys", type=int, default=1_000_000)
    a = ap.parse_args()
    dev = torch.device("cuda")
    print(hip.build_info())
    print("nnz,keys,F,atomic_ms,sort_cached_ms,aten_sort_ms,geot_sort_ms,sort_incl_ms,first_sight_op_wall_ms,torch_index_add_ms,bit_reproducible")
    for keys in (a.keys, 10_000, 100):
        index = torch.randint(0, keys, (a.nnz,), device=dev)
        index[-1] = keys - 1
        for F in (1, 8, 32, 64, 128):
            src = torch.rand(a.nnz, F, device=dev)
            out = torch.empty(keys, F, device=dev)
            t_atomic = device_ms(lambda: hip.index_scatter_out(index, src, out, sorted=False), 5)
            t_aten = device_ms(lambda: torch.sort(index, stable=True), 3)
            t_sort = device_ms(lambda: hip.sort_index(index, keys - 1), 3)
            k, perm = hip.sort_index(index, keys - 1)
            k2, perm2 = torch.sort(index, stable=True)
            assert torch.equal(k, k2) and torch.equal(perm, perm2)
            del k2, perm2

            def first_sight():
                ops.clear_caches()
                return geot.index_scatter(0, src, index, "sum", sorted=False)
            for _ in range(2):
                first_sight()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                first_sight()
            torch.cuda.synchronize()
            t_first = (time.perf_counter() - t0) / 5 * 1e3
            t_gather = device_ms(lambda: hip.gather_reduce_out(perm, k, None, src, out, "sum"), 5)
            a1 = hip.gather_reduce_out(perm, k, None, src, torch.empty_like(out), "sum")
            a2 = hip.gather_reduce_out(perm, k, None, src, torch.empty_like(out), "sum")
            ref = torch.zeros(keys, F, device=dev)
            t_torch = device_ms(lambda: ref.zero_().index_add_(0, index, src), 3)
            ok = torch.equal(a1, a2) and torch.allclose(a1, ref, rtol=1e-4, atol=1e-3)
            print(f"{a.nnz},{keys},{F},{t_atomic:.3f},{t_gather:.3f},{t_aten:.3f},{t_sort:.3f},{t_gather + t_sort:.3f},{t_first:.3f},{t_torch:.3f},{ok}",
                  flush=True)
            del src, out, ref, a1, a2


if __name__ == "__main__":
    main()
