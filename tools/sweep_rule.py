#!/usr/bin/env python3
"""Sweep of the tile-shape knobs against the built-in shape-keyed rule (`make_plan`, seg_reduce.hip) - the
MI355X counterpart of the reference's tuning pipeline (benchmark/benchmark_cpp/run_benchmark.py -> sr_result.csv
-> data/process/dtregression.py -> csrc/cuda/wrapper/*_rule.h).  For every (dataset, op, feature size) it times
the rule's own choice and a grid of forced configurations, writes one CSV row per measurement with the
reference's column vocabulary (dataname, feature_size, size, max, std, mean, config..., time, gflops) and
prints the rule's regret (auto time / best time).

Datasets are synthetic (no network): segment lengths ~ N(avg, avg*cv) clamped to [min_seg, max_seg], the
generator of csrc/dataloader/dataloader.hpp:21-62, plus the power-law generator of bench.py.

    python tools/sweep_rule.py [--out gpurun_out/sweep_rule.csv] [--iters 20] [--quick]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import powerlaw_index  # noqa: E402

import geot_amd  # noqa: E402,F401
from geot_amd import hip  # noqa: E402


def normal_cv_index(keys, avg, cv, seed, dev, min_seg=0, max_seg=1 << 30):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    ln = torch.normal(float(avg), float(avg * cv), (keys,), generator=g, device=dev).round_().clamp_(min_seg, max_seg)
    ln = ln.to(torch.int64)
    if int(ln[-1]) == 0:
        ln[-1] = 1  # the row rule reads index[-1]+1: keep the last key present
    return torch.repeat_interleave(torch.arange(keys, device=dev, dtype=torch.int64), ln)


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(2):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep_rule.csv"))
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    os.makedirs(os.path.dirname(a.out), exist_ok=True)

    datasets = []
    sizes = [1_000_000] if a.quick else [100_000, 300_000, 1_000_000, 10_000_000]    # (the two small sizes: the launch-bound tile shapes)
    for nnz in sizes:
        for avg in ([10, 500] if a.quick else [2, 10, 50, 500]):
            for cv in ([1.0] if a.quick else [0.3, 1.5]):
                keys = max(nnz // avg, 1)
                datasets.append((f"normal_n{nnz}_avg{avg}_cv{cv}", lambda k=keys, v=avg, c=cv: normal_cv_index(k, v, c, 7, dev)))
        datasets.append((f"powerlaw_n{nnz}_avg10", lambda n=nnz: powerlaw_index(n, n // 10, 0, dev)))
    feats = [4, 64] if a.quick else [1, 2, 4, 8, 16, 32, 64, 128, 256]

    rows_out = []
    regret = []
    with open(a.out, "w") as f:
        f.write("dataname,op,feature_size,size,keys,max,std,mean,edges_per_group,lpr_log2,unroll,auto,time_ms,gbs,gflops\n")
        for name, gen in datasets:
            index = gen()
            nnz = index.numel()
            K = int(index[-1]) + 1
            cnt = torch.bincount(index, minlength=K).double()
            mx, sd, mean = int(cnt.max()), float(cnt.std()), float(cnt.mean())
            g = torch.Generator(device=dev)
            g.manual_seed(3)
            src_index = torch.randint(0, K, (nnz,), device=dev, generator=g)
            for F in feats:
                if nnz * F * 4 > 12 << 30:
                    continue
                src = torch.rand(nnz, F, device=dev)
                x = torch.rand(K, F, device=dev)
                out = torch.empty(K, F, device=dev)
                ops = {
                    "index_scatter": (lambda: hip.index_scatter_out(index, src, out, True), nnz * (4 * F + 8) + K * 4 * F),
                    "gather_scatter": (lambda: hip.gather_scatter_out(src_index, index, x, out), nnz * 16 + 2 * K * 4 * F),
                }
                for op, (fn, alg) in ops.items():
                    cfgs = [(0, -1, 0)]
                    if F % 4 == 0:
                        n2 = max(2, (F // 4 - 1).bit_length())
                        for l in (n2, n2 + 1):
                            for cg in (16, 32, 64, 128):
                                if l <= 6 and (256 >> l) * cg <= 2048 and not (a.quick and l > n2):
                                    cfgs.append((cg, l, 0))
                        if op == "index_scatter":
                            for cg in (32, 64):
                                cfgs.append((cg, -1, 16 if F != 64 else 8))
                    else:
                        for cg in (16, 32, 64, 128, 256):
                            cfgs.append((cg, -1, 0))
                    res = []
                    for cg, lpr, unroll in cfgs:
                        hip.tune(cg, 0, -1, lpr)
                        hip.set_option("unroll", unroll)
                        try:
                            ms = timeit(fn, a.iters)
                        except RuntimeError as e:  # a forced shape the library refuses
                            print("skip", name, op, F, cg, lpr, unroll, str(e)[:80])
                            continue
                        auto = int(cg == 0)
                        res.append((ms, cg, lpr, unroll, auto))
                        f.write(f"{name},{op},{F},{nnz},{K},{mx},{sd:.2f},{mean:.2f},{cg},{lpr},{unroll},{auto},"
                                f"{ms:.5f},{alg / ms / 1e6:.1f},{nnz * F / ms / 1e6:.1f}\n")
                    hip.tune(0, 0, -1, -1)
                    hip.set_option("unroll", 0)
                    # the rule's own choice is timed first in its cell (cold buffers, cold code) AND once more at the end: on
                    # launch-bound cells the first slot alone read 1.2-2x slow on some boxes, whatever shape ran in it
                    again = timeit(fn, a.iters)
                    res = [(min(r[0], again),) + r[1:] if r[4] else r for r in res]
                    t_auto = [r[0] for r in res if r[4]][0]
                    best = min(res)
                    regret.append((t_auto / best[0], name, op, F, t_auto, best))
                del src, x, out
            f.flush()
    regret.sort(reverse=True)
    import math
    gm = math.exp(sum(math.log(r[0]) for r in regret) / len(regret))
    print(f"{len(regret)} (dataset, op, F) cells; rule regret geomean {gm:.3f}, "
          f"median {sorted(r[0] for r in regret)[len(regret) // 2]:.3f}, max {regret[0][0]:.3f}")
    print("worst cells (auto_ms / best_ms : best config = edges_per_group, lpr_log2, unroll):")
    for r in regret[:25]:
        print(f"  {r[0]:.3f}  {r[1]:<28s} {r[2]:<15s} F={r[3]:<4d} auto {r[4]:.4f} ms  best {r[5][0]:.4f} ms @ cg={r[5][1]} lpr={r[5][2]} U={r[5][3]}")
    with open(a.out.replace(".csv", "_regret.txt"), "w") as f:
        f.write(f"cells {len(regret)} geomean {gm:.4f} max {regret[0][0]:.4f}\n")
        for r in regret:
            f.write(f"{r[0]:.3f} {r[1]} {r[2]} F={r[3]} auto_ms={r[4]:.5f} best_ms={r[5][0]:.5f} cg={r[5][1]} lpr={r[5][2]} U={r[5][3]}\n")


if __name__ == "__main__":
    main()
