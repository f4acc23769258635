#!/usr/bin/env python3
"""Source-blocked kernel (csrc/seg_slab.hip) vs the per-edge gather kernel on the BASELINE.json configs[3] stand-in
(Reddit scale: 232 965 nodes, 114.6 M edges, H=4 x F=64) and on denser / sparser variants; slab size sweep; Phase A time.
    python tools/bench_slab.py [--scale 1.0]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import device_ms, powerlaw_index  # noqa: E402
from geot_amd import hip, slab  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--rule", action="store_true", help="density sweep around the routing threshold (gws F=128, 120 M edges)")
    a = ap.parse_args()
    dev = torch.device("cuda")
    print(hip.build_info())
    cases = [("cfg4 mh_spmm H=4 F=64", int(232_965 * a.scale), int(114_615_892 * a.scale), 4, 64, 2),
             ("cfg4 graph, gws F=128", int(232_965 * a.scale), int(114_615_892 * a.scale), 1, 128, 1),
             ("cfg4 graph, gs F=64", int(232_965 * a.scale), int(114_615_892 * a.scale), 1, 64, 0),
             ("1M nodes x deg 100, gws F=128", int(1_000_000 * a.scale), int(100_000_000 * a.scale), 1, 128, 1)]
    if os.environ.get("BENCH_SLAB_NOW"):
        cases = [("cfg4 graph, gs F=256 (1-KiB rows, no weights)", 232_965, 114_615_892, 1, 256, 0),
                 ("cfg4 graph, gws F=256 (1-KiB rows, weight[e])", 232_965, 114_615_892, 1, 256, 1)]
    if a.rule:
        cases = [(f"{n // 1000}k nodes x 120M edges, gws F=128", n, 120_000_000, 1, 128, 1) for n in (300_000, 450_000, 600_000, 800_000)] + \
                [(f"{n // 1000}k nodes x 120M edges, mh H=4 F=64", n, 120_000_000, 4, 64, 2) for n in (450_000, 700_000)]
    for name, nodes, nnz, H, Fh, wmode in cases:
        di = powerlaw_index(nnz, nodes, 11, dev)
        g = torch.Generator(device=dev)
        g.manual_seed(12)
        si = torch.randint(0, nodes, (nnz,), device=dev, generator=g)
        x = torch.rand(nodes, H, Fh, device=dev, generator=g)
        out = torch.empty(nodes, H, Fh, device=dev)
        ref = torch.empty(nodes, H, Fh, device=dev)
        if wmode == 2:
            w = torch.rand(nnz, H, device=dev, generator=g)
            base = lambda: hip.mh_spmm_out(si, di, w, x, ref, False)  # noqa: E731
        elif wmode == 1:
            w = torch.rand(nnz, device=dev, generator=g)
            base = lambda: hip.gather_weight_scatter_out(si, di, w, x.view(nodes, Fh), ref.view(nodes, Fh))  # noqa: E731
        else:
            w = None
            base = lambda: hip.gather_scatter_out(si, di, x.view(nodes, Fh), ref.view(nodes, Fh))  # noqa: E731
        t_base = device_ms(base, 3, warmup=1)
        roc = ""
        if wmode == 1:      # the comparator of the north star on THIS graph too
            from tools import rocsparse
            best, table, y = rocsparse.best_csr_spmm(di, si, w, x.view(nodes, Fh), nodes, iters=3, algs=("csr_nnz_split", "csr_merge_path"))
            roc = "  | rocSPARSE CSR SpMM on this matrix: " + ", ".join(f"{r['algorithm']} {r['ms']:.3f} ms" for r in table if r['ms'])
            del y
        print(f"{name}: nodes={nodes} nnz={nnz} rowbytes={H * Fh * 4} worthwhile={slab.worthwhile(nnz, nodes, nodes, H * Fh * 4)}  "
              f"per-edge gather kernel {t_base:.3f} ms{roc}", flush=True)
        for slab_mib, blocks in ((2.0, 3),):
            hip.set_option("slab_blocks", blocks)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            plan = slab.build_plan(si, di, nodes, nodes, H * Fh * 4, wmode, H, slab_bytes=int(slab_mib * (1 << 20)))
            torch.cuda.synchronize()
            t_plan = time.perf_counter() - t0
            times = {}
            for window in (-1, 2, -2, 2):
                hip.set_option("slab_window", window)
                times[window] = device_ms(lambda: slab.slab_spmm_out(plan, w, wmode, x, out, H, Fh), 3, warmup=1)
            hip.set_option("slab_window", -2)
            hip.set_option("slab_blocks", 3)
            base()
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            best = min(times.values())
            print(f"   slab {slab_mib} MiB, {blocks} workgroups per CU: window(-1=free,2,rule) " + " ".join(f"{times[k]:.3f}" for k in (-1, 2, -2)) +
                  f" ms  best {t_base / best:.2f}x  phase A {t_plan * 1e3:.0f} ms  plan {plan.nbytes() / 1e6:.0f} MB  "
                  f"max rel diff {err:.1e}  {plan.meta}", flush=True)
            del plan
        del di, si, x, out, ref, w
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
