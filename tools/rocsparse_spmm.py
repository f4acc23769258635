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
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import powerlaw_index  # noqa: E402
from geot_amd import hip  # noqa: E402
from tools import rocsparse  # noqa: E402


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
    t_geot = rocsparse.device_ms(lambda: hip.gather_weight_scatter_out(si, di, w, x, out), args.iters)
    print(f"graph: {nodes} nodes, {nnz} edges, F={F}, {'local' if args.local else 'uniform-random'} sources")
    print(f"geot gather_weight_scatter (int64 COO, dst-sorted): {t_geot:.3f} ms  {nnz / t_geot / 1e6:.2f} Gedge/s")
    best, table, y = rocsparse.best_csr_spmm(di, si, w, x, nodes, iters=args.iters, algs=tuple(rocsparse.ALGS))
    hip.gather_weight_scatter_out(si, di, w, x, out)
    err = ((y - out).abs().max() / out.abs().max()).item()
    for r in table:
        if r["ms"] is None:
            print(f"rocsparse_spmm CSR alg={r['algorithm']:16s}: {r['note']}")
            continue
        print(f"rocsparse_spmm CSR alg={r['algorithm']:16s}: {r['ms']:8.3f} ms  {nnz / r['ms'] / 1e6:6.2f} Gedge/s   geot is "
              f"{r['ms'] / t_geot:5.2f}x faster   (preprocess buffer {r['preprocess_buffer_bytes'] / 1e6:.0f} MB, int32 indices)")
    print(f"best: {best['algorithm']}; max rel diff of its result vs geot: {err:.1e}")


if __name__ == "__main__":
    main()
