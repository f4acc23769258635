#!/usr/bin/env python3
"""Hand-off across storage types on one workspace.  The tile kernel's hand-off marks a published carry with the call's tag and the second
launch lowers the flags again; calls of two 16-bit types with the same shapes share the workspace AND its layout.  Alternate them
with different data (a carry taken from the other type's call would show) and check every output word.
python tools/repro_epoch.py [--pairs 40]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import powerlaw_index  # noqa: E402
import geot_amd as geot  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=40)
    ap.add_argument("--nnz", type=int, default=10_000_000)
    ap.add_argument("--feat", type=int, default=128)
    a = ap.parse_args()
    dev = torch.device("cuda")
    nnz, keys, F = a.nnz, a.nnz // 10, a.feat
    index = powerlaw_index(nnz, keys, 0, dev)
    counts = torch.bincount(index, minlength=keys)
    xa = (0.125 + 0.125 * torch.rand(nnz, F, device=dev)).half()
    xb = (1.0 + torch.rand(nnz, F, device=dev)).bfloat16()          # different data: a stale carry of the other type shows
    ref_b = torch.segment_reduce(xb.double(), "sum", lengths=counts, axis=0, unsafe=True)
    ref_a = torch.segment_reduce(xa.double(), "sum", lengths=counts, axis=0, unsafe=True)
    bad = 0
    worst_a = worst_b = 0.0
    for k in range(a.pairs):
        oa = geot.index_scatter(0, xa, index, "sum", True)
        ob = geot.index_scatter(0, xb, index, "sum", True)
        ea = ((oa.double() - ref_a).abs() / (ref_a.abs() + 1e-30)).max().item()
        eb = ((ob.double() - ref_b).abs() / (ref_b.abs() + 1e-30)).max().item()
        worst_a, worst_b = max(worst_a, ea), max(worst_b, eb)
        if ea > 2.0 ** -10 or eb > 2.0 ** -7:
            bad += 1
            print(f"pair {k}: f16 max rel err {ea:.3e}, bf16 max rel err {eb:.3e}", flush=True)
    print(f"{'OK' if bad == 0 else 'MISMATCH'}: {a.pairs} pairs, worst f16 {worst_a:.3e} bf16 {worst_b:.3e}, last kernel {geot.hip.last_kernel()}, "
          f"stats {geot.ops.stats()}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
