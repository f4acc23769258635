#!/usr/bin/env python3
"""Phase A (the plan of the source-blocked kernels) at BASELINE.json configs[3] scale: wall time of one plan build on an
idle stream (stream synchronised before and after), device builder (csrc/seg_plan.hip) vs the ATen formulation, first
build of the process and repeats.   python tools/bench_plan.py [--scale 1.0] [--aten]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import powerlaw_index  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--aten", action="store_true", help="also time the ATen formulation")
    ap.add_argument("--reps", type=int, default=4)
    a = ap.parse_args()
    from geot_amd import ops, slab
    dev = torch.device("cuda")
    nodes, nnz, H, F = int(232_965 * a.scale), int(114_615_892 * a.scale), 4, 64
    di = powerlaw_index(nnz, nodes, 11, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(12)
    si = torch.randint(0, nodes, (nnz,), device=dev, generator=g)
    builders = [("device", 0)] + ([("aten", 1)] if a.aten else [])
    for name, code in builders:
        ops.set_option("slab_builder", code)
        for rep in range(a.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            plan = slab.build_plan(si, di, nodes, nodes, H * F * 4, 2, H)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
            print(f"{name:6s} build {rep}: {ms:8.2f} ms   groups {plan.struct.n_groups} vrows {plan.struct.n_vrows} split {plan.struct.n_split} "
                  f"plan bytes {plan.nbytes() / 1e6:.0f} MB  peak alloc {torch.cuda.max_memory_allocated() / 1e9:.2f} GB", flush=True)
            del plan
            torch.cuda.reset_peak_memory_stats()
    ops.set_option("slab_builder", 0)


if __name__ == "__main__":
    main()
