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


def matrix(calls):
    """The shapes that USE the hand-off (launcher rule, seg_reduce.hip run_segment_op: rows >= 256 B, or <= 2 M edges): F in
    {64, 128, 256} x {fp32, bf16, f16} x {sum, mean} x {10 M, 1 M edges}, `calls` calls each.  Data change on every call (an
    exact power-of-two scale: a stale carry line shows), a second stream injects bursts of streaming and matmul work, every
    output word is checked on the device against float64.  (The consumer reads the handed-off rows with sc1 loads only - they
    bypass L1, so a warm L1 has nothing stale to offer; MI355X_MICROARCH.md, valid forms.)"""
    dev = torch.device("cuda")
    noise = torch.cuda.Stream()
    m = torch.rand(2048, 2048, device=dev)
    big = torch.rand(32 << 20, device=dev)
    g = torch.Generator(device="cpu").manual_seed(0)
    failures = 0
    print("edges keys F dtype reduce rowbytes hand-off(by rule) calls worst_rel_err bound verdict", flush=True)
    for nnz in (10_000_000, 1_000_000):
        keys = nnz // 10
        index = powerlaw_index(nnz, keys, 0, dev)
        counts = torch.bincount(index, minlength=keys)
        for F in (64, 128, 256):
            # values in [0.125, 0.25): x 2^-2 .. 2^2 is EXACT in every format (no float16 subnormals) and a hub row of 33 k edges x 2^2 stays
            # inside the float16 range
            base32 = 0.125 + 0.125 * torch.rand(nnz, F, device=dev)
            for dtype, eps in ((torch.float32, 1e-5), (torch.bfloat16, 2.0 ** -8), (torch.float16, 2.0 ** -11)):
                base = base32.to(dtype)
                ref_sum = torch.segment_reduce(base.double(), "sum", lengths=counts, axis=0, unsafe=True)
                rowbytes = F * base.element_size()
                by_rule = rowbytes >= 256 or nnz <= 2_000_000
                for reduce in ("sum", "mean"):
                    ref = ref_sum if reduce == "sum" else ref_sum / counts.clamp(min=1).double()[:, None]
                    src = torch.empty_like(base)
                    worst, bad = 0.0, 0
                    for it in range(calls):
                        k = int(torch.randint(-2, 3, (1,), generator=g))
                        torch.mul(base, 2.0 ** k, out=src)                   # exact in every float format
                        if it % 3 == 0:
                            with torch.cuda.stream(noise):
                                for _ in range(int(torch.randint(1, 5, (1,), generator=g))):
                                    if it % 2:
                                        m = torch.mm(m, m).clamp_(-1, 1)
                                    else:
                                        big.mul_(1.0001)
                        out = geot.index_scatter(0, src, index, reduce, True)
                        want = ref * (2.0 ** k)
                        err = ((out.double() - want).abs() / (want + 1e-30)).max().item()
                        worst = max(worst, err)
                        if err > 1.01 * eps:
                            bad += 1
                            if bad > 3:
                                break
                    failures += bad
                    print(f"{nnz} {keys} {F} {str(dtype).split('.')[-1]} {reduce} {rowbytes} {'yes' if by_rule else 'no (classic second pass)'} "
                          f"{calls} {worst:.3e} {eps:.1e} {'ok' if bad == 0 else 'MISMATCH x%d' % bad}", flush=True)
                    del ref, src
                del ref_sum, base
            del base32
    torch.cuda.synchronize()
    print(f"{'STRESS PASSED' if failures == 0 else 'STRESS FAILED'}: stats {geot.ops.stats()}")
    sys.exit(1 if failures else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=4000)
    ap.add_argument("--matrix", action="store_true", help="the shapes that use the hand-off x dtypes x sum / mean, --calls each")
    a = ap.parse_args()
    if a.matrix:
        return matrix(a.calls)
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
