#!/usr/bin/env python3
"""Long stress of the in-kernel hand-off at the GRADED configuration (10 M power-law edges -> 1 M rows, F=64: 19 532 tiles, ~17 k
straddling runs per call, hubs over up to 65 tiles): thousands of calls, new data every call, every output word checked on the
device against float64, a second stream injecting bursts of other work.  `python tools/stress_handoff.py [--calls 4000]`"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import FEAT, KEYS, NNZ, powerlaw_index  # noqa: E402
import geot_amd as geot  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=4000)
    a = ap.parse_args()
    dev = torch.device("cuda")
    index = powerlaw_index(NNZ, KEYS, 0, dev)
    base = torch.rand(NNZ, FEAT, device=dev)
    ref = torch.segment_reduce(base.double(), "sum", lengths=torch.bincount(index, minlength=KEYS), axis=0, unsafe=True)
    noise = torch.cuda.Stream()
    m = torch.rand(2048, 2048, device=dev)
    big = torch.rand(32 << 20, device=dev)
    src = torch.empty_like(base)
    worst, bad = 0.0, 0
    g = torch.Generator(device="cpu").manual_seed(0)
    for it in range(a.calls):
        scale = 0.5 + float(torch.rand(1, generator=g))
        torch.mul(base, scale, out=src)
        if it % 3 == 0:
            with torch.cuda.stream(noise):
                for _ in range(int(torch.randint(1, 5, (1,), generator=g))):
                    if it % 2:
                        m = torch.mm(m, m).clamp_(-1, 1)
                    else:
                        big.mul_(1.0001)
        out = geot.index_scatter(0, src, index, "sum", True)
        err = ((out.double() - ref * scale).abs() / (ref * scale + 1e-30)).max().item()
        worst = max(worst, err)
        if err > 1e-5:
            bad += 1
            print(f"call {it}: max rel err {err:.3e}", flush=True)
            if bad > 5:
                break
        if it % 500 == 0:
            print(f"call {it}: worst rel err so far {worst:.2e}", flush=True)
    torch.cuda.synchronize()
    print(f"{'STRESS PASSED' if bad == 0 else 'STRESS FAILED'}: {a.calls} calls, worst max-relative error {worst:.2e}, stats {geot.ops.stats()}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
