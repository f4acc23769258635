#!/usr/bin/env python3
"""sorted=False on an index WITH descents: the two paths side by side.
  atomic : tile kernel, runs pre-reduced in registers, float atomics into a zeroed dst (non-deterministic)
  sort   : stable sort of the index once (kept with the index facts) -> gather-mode kernels over (keys, perm);
           deterministic, any reduction.  Timed with the sort cached (steady state) and including the sort.
    python tools/bench_unsorted.py [--nnz 10000000] [--keys 1000000]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geot_amd import hip  # noqa: E402
from tools.rocsparse import device_ms  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nnz", type=int, default=10_000_000)
    ap.add_argument("--keys", type=int, default=1_000_000)
    a = ap.parse_args()
    dev = torch.device("cuda")
    print(hip.build_info())
    print("nnz,keys,F,atomic_ms,sort_cached_ms,sort_ms,sort_incl_ms,torch_index_add_ms,bit_reproducible")
    for keys in (a.keys, 10_000, 100):
        index = torch.randint(0, keys, (a.nnz,), device=dev)
        index[-1] = keys - 1
        for F in (1, 8, 32, 64, 128):
            src = torch.rand(a.nnz, F, device=dev)
            out = torch.empty(keys, F, device=dev)
            t_atomic = device_ms(lambda: hip.index_scatter_out(index, src, out, sorted=False), 5)
            t_sort = device_ms(lambda: torch.sort(index, stable=True), 3)
            k, perm = torch.sort(index, stable=True)
            t_gather = device_ms(lambda: hip.gather_reduce_out(perm, k, None, src, out, "sum"), 5)
            a1 = hip.gather_reduce_out(perm, k, None, src, torch.empty_like(out), "sum")
            a2 = hip.gather_reduce_out(perm, k, None, src, torch.empty_like(out), "sum")
            ref = torch.zeros(keys, F, device=dev)
            t_torch = device_ms(lambda: ref.zero_().index_add_(0, index, src), 3)
            ok = torch.equal(a1, a2) and torch.allclose(a1, ref, rtol=1e-4, atol=1e-3)
            print(f"{a.nnz},{keys},{F},{t_atomic:.3f},{t_gather:.3f},{t_sort:.3f},{t_gather + t_sort:.3f},{t_torch:.3f},{ok}", flush=True)
            del src, out, ref, a1, a2


if __name__ == "__main__":
    main()
