#!/usr/bin/env python3
"""Joint sweep of the source-blocked kernel's knobs at BASELINE.json configs[3] (mh_spmm, Reddit scale) and its gws sibling:
slab size x lockstep window x workgroups per CU (x rows per group), at the FINAL R of the plan - round 3 swept them one at a
time.  The table is 238 MB: with 2-MiB slabs and a window of 2 a wave may be three slabs (6 MiB) away from the slowest wave of
its XCD, more than the XCD's 4 MiB L2.

    python tools/sweep_slab.py [--sources uniform|powerlaw] [--quick]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# (the development build - the experiment switches live there only - when the command line names one of them; the product otherwise:
#  the development build's kernels carry knock-out branches and time a few per cent slower)
if any(n in " ".join(sys.argv) for n in ("slab_probe", "slab_pair", "slab_wrow_all", "slab_nt", "slab_tight", "slab_stage", "slab_unroll")):
    os.environ.setdefault("GEOT_HIP_LIB", "dev")
from bench import device_ms, powerlaw_index  # noqa: E402
from geot_amd import hip, slab  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sources", default="uniform")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--case", default="mh")
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--coalesced", action="store_true", help="sources sorted inside every dst row (what torch_geometric's coalesce() / a CSR leaves)")
    ap.add_argument("--seq", default="", help="fixed sequence for a PMC pass: 'MiB:window,MiB:window,...' - two launches of the slab kernel per "
                    "setting, nothing else of that name launched (tools/pmc_slab_grid.sh maps the dispatches back)")
    ap.add_argument("--ab", default="", help="A/B of one library option at the default plan: NAME=v0,v1 (e.g. slab_nt=0,1 or slab_far=4,12,1000000)")
    a = ap.parse_args()
    dev = torch.device("cuda")
    print(hip.build_info(), flush=True)
    nodes, nnz = 232_965, 114_615_892
    sddmm = a.case.startswith("sddmm")                   # d/dweight of gather_weight_scatter over the forward's plan (weight_mode 1)
    H, Fh, wmode = {"mh": (4, 64, 2), "gws": (1, 128, 1), "gs64": (1, 64, 0), "gs128": (1, 128, 0), "gws256": (1, 256, 1),
                    "sddmm128": (1, 128, 1), "sddmm256": (1, 256, 1)}[a.case]
    di = powerlaw_index(nnz, nodes, 11, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(12)
    si = di[torch.randperm(nnz, device=dev, generator=g)].contiguous() if a.sources == "powerlaw" else torch.randint(0, nodes, (nnz,), device=dev, generator=g)
    if a.coalesced:
        si = (torch.sort(di * nodes + si).values % nodes).contiguous()
    dt = getattr(torch, a.dtype)
    esz = 2 if dt != torch.float32 else 4
    x = torch.rand(nodes, H, Fh, device=dev, generator=g).to(dt)
    w = (torch.rand(nnz, H, device=dev, generator=g) if wmode == 2 else torch.rand(nnz, device=dev, generator=g)).to(dt)
    if wmode == 0:
        w = None
    out = torch.empty(nodes, H, Fh, device=dev, dtype=dt)
    ref = torch.empty(nodes, H, Fh, device=dev, dtype=dt)
    if sddmm:
        m1 = torch.rand(nodes, Fh, device=dev, generator=g).to(dt)
        eo, eref = torch.empty(nnz, device=dev, dtype=dt), torch.empty(nnz, device=dev, dtype=dt)
        base = lambda: hip.sddmm_coo_out(si, di, m1, x.view(nodes, Fh), eref)  # noqa: E731
    elif wmode == 2:
        base = lambda: hip.mh_spmm_out(si, di, w, x, ref, False)  # noqa: E731
    elif wmode == 1:
        base = lambda: hip.gather_weight_scatter_out(si, di, w, x.view(nodes, Fh), ref.view(nodes, Fh))  # noqa: E731
    else:
        base = lambda: hip.gather_scatter_out(si, di, x.view(nodes, Fh), ref.view(nodes, Fh))  # noqa: E731
    if a.seq:
        last = None
        for item in a.seq.split(","):
            mib, k = item.split(":")
            if mib != last:
                plan = slab.build_plan(si, di, nodes, nodes, H * Fh * esz, wmode, H, slab_bytes=int(float(mib) * (1 << 20)), rows_per_group=slab.rows_per_group(wmode, H, dt, H * Fh * esz))
                last = mib
            hip.set_option("slab_window", int(k))
            for _ in range(2):
                if sddmm:
                    slab.slab_sddmm_out(plan, m1, x.view(nodes, Fh), eo)
                else:
                    slab.slab_spmm_out(plan, w, wmode, x, out, H, Fh)
            torch.cuda.synchronize()
            print(f"setting {item}: R={plan.meta['rows_per_group']} rounds={plan.meta['rounds']} slabs={plan.meta['slabs']}", flush=True)
        return
    print(f"case={a.case} dtype={a.dtype} sources={a.sources} coalesced={a.coalesced} per-edge {device_ms(base, 3, warmup=1):.3f} ms", flush=True)
    if a.ab:
        name, vals = a.ab.split("=")
        vals = [int(v) for v in vals.split(",")]
        plan = slab.build_plan(si, di, nodes, nodes, H * Fh * esz, wmode, H, rows_per_group=slab.rows_per_group(wmode, H, dt, H * Fh * esz))
        for rep in range(3):
            row = []
            for v in vals:
                if name != "staged":                      # ("staged=0,1": the two forms of the SDDMM's output, not a library option)
                    hip.set_option(name, v)
                run = (lambda: slab.slab_sddmm_out(plan, m1, x.view(nodes, Fh), eo, staged=bool(v) if name == "staged" else True)) if sddmm else (lambda: slab.slab_spmm_out(plan, w, wmode, x, out, H, Fh))
                row.append(device_ms(run, 5, warmup=1))
            print(f"{name}: " + "  ".join(f"{v}: {m:.3f} ms" for v, m in zip(vals, row)), flush=True)
        return
    slabs = (1.0, 2.0) if a.quick else (0.5, 1.0, 1.5, 2.0, 3.0, 4.0)
    windows = (1, 2) if a.quick else (0, 1, 2, 3, 4)
    print("blocks slab_MiB " + " ".join(f"w={k:<6d}" for k in windows) + "  (ms; plan meta)")
    best = (1e9, None)
    for blocks in (3, 2) if not a.quick else (3,):
        hip.set_option("slab_blocks", blocks)
        for mib in slabs:
            plan = slab.build_plan(si, di, nodes, nodes, H * Fh * esz, wmode, H, slab_bytes=int(mib * (1 << 20)), rows_per_group=slab.rows_per_group(wmode, H, dt, H * Fh * esz))
            row = []
            for k in windows:
                hip.set_option("slab_window", k)
                run = (lambda: slab.slab_sddmm_out(plan, m1, x.view(nodes, Fh), eo)) if sddmm else (lambda: slab.slab_spmm_out(plan, w, wmode, x, out, H, Fh))
                ms = device_ms(run, 4, warmup=1)
                row.append(ms)
                if ms < best[0]:
                    best = (ms, (blocks, mib, k))
            base()
            err = (((eo.float() - eref.float()).abs().max() / eref.float().abs().max()) if sddmm else
                   ((out.float() - ref.float()).abs().max() / ref.float().abs().max())).item()
            print(f"{blocks:6d} {mib:8.1f} " + " ".join(f"{m:8.3f}" for m in row) + f"   R={plan.meta['rows_per_group']} rounds={plan.meta['rounds']} "
                  f"slabs={plan.meta['slabs']} err={err:.1e}", flush=True)
            del plan
    hip.set_option("slab_window", -2)
    hip.set_option("slab_blocks", 3)
    print("best", best)


if __name__ == "__main__":
    main()
