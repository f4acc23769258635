#!/usr/bin/env python3
"""Direct rocSPARSE CSR SpMM (generic API, via ctypes) on the same matrix as geot.gather_weight_scatter:
the comparison BASELINE.json's north_star asks for ("gather_weight_scatter SpMM beating rocSPARSE CSR SpMM on
the same graph").  torch.sparse.mm is also a rocSPARSE/hipSPARSE call, but through torch's own layout
choices; this calls librocsparse's rocsparse_spmm with each CSR algorithm (row split, row split + shfl,
nnz split, merge path), int32 indices (rocSPARSE's native width), row-major dense operands, preprocess
stage excluded from the timing (it is per-matrix work, like our cached structures).

    python tools/rocsparse_spmm.py [--scale 1.0] [--feat 128] [--local]
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import powerlaw_index  # noqa: E402
from geot_amd import hip  # noqa: E402

OP_NONE, BASE0, I32, F32R, ROW = 111, 0, 2, 151, 0
ALGS = {"default": 0, "csr (row split, shared mem)": 1, "csr_row_split (shfl)": 4, "csr_nnz_split/merge": 5,
        "csr_merge_path": 9}
STAGE_BUF, STAGE_PRE, STAGE_COMPUTE = 1, 2, 3


def timeit(fn, iters):
    for _ in range(2):
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
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--feat", type=int, default=128)
    ap.add_argument("--local", action="store_true", help="sources within +-2000 rows of the destination")
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda")
    nodes, nnz, F = int(2_449_029 * args.scale), int(123_718_280 * args.scale), args.feat
    di = powerlaw_index(nnz, nodes, 7, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(8)
    if args.local:
        si = (di + torch.randint(-2000, 2001, (nnz,), device=dev, generator=g)).clamp_(0, nodes - 1)
    else:
        si = torch.randint(0, nodes, (nnz,), device=dev, generator=g)
    w = torch.rand(nnz, device=dev)
    x = torch.rand(nodes, F, device=dev)
    out = torch.empty(nodes, F, device=dev)
    t_geot = timeit(lambda: hip.gather_weight_scatter_out(si, di, w, x, out), args.iters)
    print(f"graph: {nodes} nodes, {nnz} edges, F={F}, {'local' if args.local else 'uniform-random'} sources")
    print(f"geot gather_weight_scatter (int64 COO, dst-sorted): {t_geot:.3f} ms  {nnz / t_geot / 1e6:.2f} Gedge/s")

    rowptr = torch.zeros(nodes + 1, dtype=torch.int32, device=dev)
    rowptr[1:] = torch.cumsum(torch.bincount(di, minlength=nodes), 0).int()
    col = si.int()
    L = ctypes.CDLL("librocsparse.so")
    vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    handle = vp()
    assert L.rocsparse_create_handle(ctypes.byref(handle)) == 0
    L.rocsparse_set_stream(handle, vp(torch.cuda.current_stream().cuda_stream))
    A, B, C = vp(), vp(), vp()
    L.rocsparse_create_csr_descr.argtypes = [ctypes.POINTER(vp), i64, i64, i64, vp, vp, vp, ci, ci, ci, ci]
    assert L.rocsparse_create_csr_descr(ctypes.byref(A), nodes, nodes, nnz, rowptr.data_ptr(), col.data_ptr(),
                                        w.data_ptr(), I32, I32, BASE0, F32R) == 0
    L.rocsparse_create_dnmat_descr.argtypes = [ctypes.POINTER(vp), i64, i64, i64, vp, ci, ci]
    y = torch.empty(nodes, F, device=dev)
    assert L.rocsparse_create_dnmat_descr(ctypes.byref(B), nodes, F, F, x.data_ptr(), F32R, ROW) == 0
    assert L.rocsparse_create_dnmat_descr(ctypes.byref(C), nodes, F, F, y.data_ptr(), F32R, ROW) == 0
    alpha, beta = ctypes.c_float(1.0), ctypes.c_float(0.0)
    L.rocsparse_spmm.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, ci, ci, ci, ctypes.POINTER(ctypes.c_size_t), vp]
    ref = None
    for name, alg in ALGS.items():
        size = ctypes.c_size_t(0)
        rc = L.rocsparse_spmm(handle, OP_NONE, OP_NONE, ctypes.byref(alpha), A, B, ctypes.byref(beta), C, F32R, alg,
                              STAGE_BUF, ctypes.byref(size), None)
        if rc != 0:
            print(f"rocsparse_spmm alg {name}: buffer_size stage returned status {rc} (not supported for this layout)")
            continue
        buf = torch.empty(max(size.value, 16), dtype=torch.uint8, device=dev)
        rc = L.rocsparse_spmm(handle, OP_NONE, OP_NONE, ctypes.byref(alpha), A, B, ctypes.byref(beta), C, F32R, alg,
                              STAGE_PRE, ctypes.byref(size), buf.data_ptr())
        if rc != 0:
            print(f"rocsparse_spmm alg {name}: preprocess returned status {rc}")
            continue

        def run():
            r = L.rocsparse_spmm(handle, OP_NONE, OP_NONE, ctypes.byref(alpha), A, B, ctypes.byref(beta), C, F32R, alg,
                                 STAGE_COMPUTE, ctypes.byref(size), buf.data_ptr())
            assert r == 0, r
        run()
        torch.cuda.synchronize()
        hip.gather_weight_scatter_out(si, di, w, x, out)
        err = ((y - out).abs().max() / out.abs().max()).item()
        t = timeit(run, args.iters)
        print(f"rocsparse_spmm CSR alg={name:30s}: {t:8.3f} ms  {nnz / t / 1e6:6.2f} Gedge/s   geot is {t / t_geot:5.2f}x faster   "
              f"(max rel diff {err:.1e}, buffer {size.value / 1e6:.0f} MB)")


if __name__ == "__main__":
    main()
