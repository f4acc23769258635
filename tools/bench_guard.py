#!/usr/bin/env python3
"""What the content guard costs (csrc/seg_guard.hip; "content guard" in csrc/host_cache.cpp).

1. the fingerprint kernel alone on two int64 index arrays of E edges (HIP events, streamed bytes / time);
2. operators that use a remembered product, content_guard on / off alternating in ONE process on one box:
   mh_spmm at Reddit scale over the slab plan (BASELINE.json configs[3]), its bf16 form, gather_weight_scatter's backward
   over the remembered transposed edge list (ogbn-products scale would not fit beside the forward buffers: 40 M edges).

    python tools/bench_guard.py [--edges 114615892]
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import powerlaw_index  # noqa: E402

import geot_amd as geot  # noqa: E402
from geot_amd import _lib, ops  # noqa: E402


def events(fn, iters):
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
    ap.add_argument("--edges", type=int, default=114_615_892)
    ap.add_argument("--nodes", type=int, default=232_965)
    args = ap.parse_args()
    dev = torch.device("cuda")
    E, N = args.edges, args.nodes
    di = powerlaw_index(E, N, 3, dev)
    si = torch.randint(0, N, (E,), device=dev)
    L = _lib.load()
    fp = torch.zeros(2, dtype=torch.int64, device=dev)
    scratch = torch.zeros(L.geot_content_fingerprint_scratch_bytes() // 8, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def launch(bufs):
        n = len(bufs)
        ptrs = (ctypes.c_void_p * n)(*[b.data_ptr() for b in bufs])
        sizes = (ctypes.c_size_t * n)(*[b.numel() * b.element_size() for b in bufs])
        assert L.geot_content_fingerprint(ptrs, sizes, n, fp.data_ptr(), 0, None, 0, scratch.data_ptr(), st) == 0

    for name, bufs in (("src_index + dst_index (int64)", [si, di]), ("dst_index alone", [di]), ("int32 copy of both", [si.int(), di.int()]),
                       ("dst_index from element 1 (8-byte aligned only)", [di[1:]])):
        ms = events(lambda: launch(bufs), 20)
        gb = sum(b.numel() * b.element_size() for b in bufs) / 1e9
        print(f"fingerprint  {name:48s} {gb:6.3f} GB  {ms:7.4f} ms  {gb / ms:6.2f} TB/s")

    H, F = 4, 64
    for dtype in (torch.float32, torch.bfloat16):
        x = torch.rand(N, H, F, device=dev).to(dtype)
        w = torch.rand(E, H, device=dev).to(dtype)
        for _ in range(3):
            geot.mh_spmm(si, di, w, x)
        row = []
        for rep in range(3):
            for g in (1, 0):
                ops.set_option("content_guard", g)
                row.append((g, events(lambda: geot.mh_spmm(si, di, w, x), 10)))
        ops.set_option("content_guard", 1)
        on = min(t for g, t in row if g == 1)
        off = min(t for g, t in row if g == 0)
        print(f"mh_spmm {E} edges H={H} F={F} {str(dtype)[6:]:9s}: guard on {on:.3f} ms, off {off:.3f} ms  (+{(on / off - 1) * 100:.1f} %)   "
              + " ".join(f"{g}:{t:.3f}" for g, t in row))
        del x, w
    # backward over the remembered transposed edge list
    E2 = min(E, 40_000_000)
    si2, di2 = si[:E2].clone(), di[:E2].clone()
    di2[-1] = N - 1
    w = torch.rand(E2, device=dev)
    g_out = torch.rand(N, 128, device=dev)

    def fwd_bwd():
        x = torch.rand(N, 128, device=dev, requires_grad=True)
        geot.gather_weight_scatter(si2, di2, w, x).backward(g_out)

    for _ in range(3):
        fwd_bwd()
    row = []
    for rep in range(3):
        for g in (1, 0):
            ops.set_option("content_guard", g)
            row.append((g, events(fwd_bwd, 5)))
    ops.set_option("content_guard", 1)
    on = min(t for g, t in row if g == 1)
    off = min(t for g, t in row if g == 0)
    print(f"gather_weight_scatter forward + backward {E2} edges F=128 float32: guard on {on:.3f} ms, off {off:.3f} ms  (+{(on / off - 1) * 100:.1f} %)")
    print("stats:", {k: v for k, v in ops.stats().items() if k in ("guard_checks", "stale_products", "plans", "transposed")})


if __name__ == "__main__":
    main()
