#!/usr/bin/env python3
"""A/B of two builds of libgeot_hip.so on ONE box (boxes of the pool differ by several percent, see
profiles/r02/sessions.md): the graded workload (BASELINE.json configs[1]) through the C ABI of each library, alternately,
`rounds` x `iters` launches, HIP events around each batch.
    python tools/ab_libs.py geot_amd/libgeot_hip.so tools/_ab/libgeot_old.so
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import FEAT, KEYS, NNZ, powerlaw_index  # noqa: E402


def load(path):
    L = ctypes.CDLL(os.path.abspath(path))
    L.geot_workspace_bytes.restype = ctypes.c_size_t
    L.geot_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
    L.geot_index_scatter.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                                                  ctypes.c_size_t, ctypes.c_void_p]
    return L


def main():
    paths = sys.argv[1:3]
    dev = torch.device("cuda")
    index = powerlaw_index(NNZ, KEYS, 0, dev)
    src = torch.rand(NNZ, FEAT, device=dev)
    out = torch.empty(KEYS, FEAT, device=dev)
    libs = []
    for p in paths:
        L = load(p)
        ws = torch.zeros(L.geot_workspace_bytes(NNZ, FEAT, KEYS, 0), dtype=torch.uint8, device=dev)
        libs.append((p, L, ws))
    st = torch.cuda.current_stream().cuda_stream

    def run(L, ws, n):
        for _ in range(n):
            rc = L.geot_index_scatter(index.data_ptr(), src.data_ptr(), out.data_ptr(), NNZ, FEAT, KEYS, 0, 1, ws.data_ptr(), ws.numel(), st)
            assert rc == 0
    ref = torch.zeros(KEYS, FEAT, device=dev, dtype=torch.float64).index_add_(0, index, src.double())
    for p, L, ws in libs:
        run(L, ws, 20)
        torch.cuda.synchronize()
        assert torch.allclose(out.double(), ref, rtol=1e-5, atol=1e-4), p
    for r in range(4):
        for p, L, ws in libs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(L, ws, 100)
            e1.record()
            torch.cuda.synchronize()
            print(f"round {r} {p}: {e0.elapsed_time(e1) * 10:.1f} us per call (tile + fix-up)", flush=True)


if __name__ == "__main__":
    main()
