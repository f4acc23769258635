#!/usr/bin/env python3
"""Where a training step through geot_amd.Graph spends its time at configs[3]'s graph (gws F=128, sources ascending inside every row):
every piece of forward + backward by itself, then the whole step, next to the dispatched operators with and without the content guard.
    python tools/bench_handle_parts.py [--scale 1.0]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import device_ms, powerlaw_index  # noqa: E402
import geot_amd as geot  # noqa: E402
from geot_amd import hip, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    a = ap.parse_args()
    dev = torch.device("cuda")
    nodes, nnz, F = int(232_965 * a.scale), int(114_615_892 * a.scale), 128
    di = powerlaw_index(nnz, nodes, 11, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(12)
    si = torch.randint(0, nodes, (nnz,), device=dev, generator=g)
    si = (torch.sort(di * nodes + si).values % nodes).contiguous()
    x = torch.rand(nodes, F, device=dev, generator=g, requires_grad=True)
    w = torch.rand(nnz, device=dev, generator=g, requires_grad=True)
    cot = torch.rand(nodes, F, device=dev, generator=g)
    h = geot.Graph(si, di, num_src=nodes, num_dst=nodes)

    def hstep():
        x.grad = None
        w.grad = None
        h.gather_weight_scatter(w, x).backward(cot)
    for _ in range(3):
        hstep()
    print(f"# {hip.build_info()}  handle after warm-up: {h.stats}")
    xd, wd = x.detach(), w.detach()
    print(f"forward, edge-order weight                 {device_ms(lambda: h._spmm('fwd', wd, xd), 4):8.3f} ms  {hip.last_kernel()}")
    fplan = h._plan('fwd', F * 4, 1, 1, torch.float32, nodes)
    bplan = h._plan('bwd', F * 4, 1, 1, torch.float32, nodes)
    print(f"plans: fwd {None if fplan is None else fplan.meta}  bwd {None if bplan is None else bplan.meta}  verdicts {h._verdict}")
    idx = h._values_for_bwd(None, bplan)
    out = torch.empty_like(wd)
    print(f"weights into the transposed plan's order   {device_ms(lambda: hip.gather_rows_out(idx, wd, out), 4):8.3f} ms")
    wt = hip.gather_rows_out(idx, wd, out)
    print(f"d/dsrc over the transposed plan (mode 4)   {device_ms(lambda: h._spmm('bwd', (bplan, wt) if bplan is not None else wt, cot), 4):8.3f} ms  {hip.last_kernel()}")
    print(f"d/dweight: SDDMM, edge order               {device_ms(lambda: h._sddmm(cot, xd, False), 4):8.3f} ms  {hip.last_kernel()}")
    print(f"d/dweight: SDDMM, plan order               {device_ms(lambda: h._sddmm(cot, xd, True), 4):8.3f} ms")
    print(f"whole step through the handle              {device_ms(hstep, 4):8.3f} ms   {h.stats}")

    def step():
        x.grad = None
        w.grad = None
        geot.gather_weight_scatter(si, di, w, x).backward(cot)
    for guard in (1, 0):
        ops.set_option("content_guard", guard)
        ops.clear_caches()
        for _ in range(4):
            step()
        print(f"whole step, operators, content_guard={guard}    {device_ms(step, 4):8.3f} ms")
    ops.set_option("content_guard", 1)
    print(f"operators' pieces: d/dweight {device_ms(lambda: torch.ops.geot.sddmm_coo_impl(si, di, cot, xd), 4):8.3f} ms ({hip.last_kernel()}); "
          f"forward {device_ms(lambda: geot.gather_weight_scatter(si, di, wd, xd), 4):8.3f} ms")


if __name__ == "__main__":
    main()
