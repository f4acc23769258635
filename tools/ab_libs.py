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
    paths = [a for a in sys.argv[1:] if not a.startswith("--")][:2]
    check = "--no-check" not in sys.argv
    dev = torch.device("cuda")
    index = powerlaw_index(NNZ, KEYS, 0, dev)
    src = torch.rand(NNZ, FEAT, device=dev)
    out = torch.empty(KEYS, FEAT, device=dev)
    libs = []
    ws = None                                   # ONE workspace for both builds: same addresses, same physical pages
    for p in paths:
        L = load(p)
        need = L.geot_workspace_bytes(NNZ, FEAT, KEYS, 0)
        if ws is None or ws.numel() < need:
            ws = torch.zeros(need, dtype=torch.uint8, device=dev)
        libs.append([p, L, None])
    for entry in libs:
        entry[2] = ws
    st = torch.cuda.current_stream().cuda_stream

    def run(L, ws, n):
        for _ in range(n):
            rc = L.geot_index_scatter(index.data_ptr(), src.data_ptr(), out.data_ptr(), NNZ, FEAT, KEYS, 0, 1, ws.data_ptr(), ws.numel(), st)
            assert rc == 0
    ref = torch.zeros(KEYS, FEAT, device=dev, dtype=torch.float64).index_add_(0, index, src.double())
    for p, L, ws in libs:
        run(L, ws, 20)
        torch.cuda.synchronize()
        ok = torch.allclose(out.double(), ref, rtol=1e-5, atol=1e-4)
        print(f"{p}: result {'matches' if ok else 'DIFFERS from'} the float64 reference", flush=True)
        assert ok or not check, p
    import ctypes as ct
    for r in range(6):
        for p, L, ws in (libs if r % 2 == 0 else libs[::-1]):                  # alternate the order too
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(L, ws, 100)
            e1.record()
            torch.cuda.synchronize()
            print(f"round {r} {p}: {e0.elapsed_time(e1) * 10:.1f} us per call (tile + fix-up)", flush=True)
    # per kernel, from each library's own event brackets (geot_profile_*): tile kernel / what follows it
    for p, L, ws in libs:
        L.geot_profile_enable.argtypes = [ct.c_int]
        L.geot_profile_read.argtypes = [ct.POINTER(ct.c_double)] * 3 + [ct.POINTER(ct.c_int64)]
        L.geot_profile_enable(1)
        L.geot_profile_reset()
        run(L, ws, 200)
        torch.cuda.synchronize()
        m, f, a, n = ct.c_double(), ct.c_double(), ct.c_double(), ct.c_int64()
        L.geot_profile_read(ct.byref(m), ct.byref(f), ct.byref(a), ct.byref(n))
        L.geot_profile_enable(0)
        if n.value:
            print(f"{p}: tile kernel {m.value / n.value * 1e3:.1f} us, fix-up launch {f.value / n.value * 1e3:.1f} us  ({n.value} calls, HIP events)", flush=True)


if __name__ == "__main__":
    main()
