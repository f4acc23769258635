#!/usr/bin/env python3
"""The source-blocked kernels at BASELINE.json configs[3]'s graph (Reddit scale stand-in: 232 965 nodes, 114.6 M power-law edges,
uniform-random sources), one line per (operator, storage type, row width, weight form): milliseconds per launch through the
pointer-level doorway (geot_amd/slab.py), plans built with the library's own units and rows per group.  Round 5: weights may arrive
in plan order (modes 4 / 5), one weight per edge in edge order is staged into plan order by a pre-pass when the workspace has room
("as the ABI serves them"; `stage_weights=False` = read through the permutation in the row loop), the multi-head SDDMM runs over the
plan; `--options slab_wrow_all=1`: every plan of 512 / 256-byte rows cut into waves (`profiles/r05/slab_cases_*.txt`).

    python tools/bench_slab_cases.py [--scale 1.0] [--only mh,gws,...] [--options slab_unroll=16,slab_tight=0]
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
from geot_amd import graph, hip, slab  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--only", default="")
    ap.add_argument("--options", default="", help="comma-separated name=value library options set for the whole run")
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--slab-mib", type=float, default=0.0, help="slab size of the plans (0 = the host layer's rule)")
    ap.add_argument("--dtypes", default="fp32,bf16")
    ap.add_argument("--rows-per-group", type=int, default=0, help="R of every plan (0 = the library's rule): fewer rows = more rounds")
    ap.add_argument("--mh-lane-groups", action="store_true", help="multi-head plans over 512 / 256-byte rows cut into lane groups (16 bytes a lane) "
                                                                  "instead of waves")
    ap.add_argument("--mh-shape", default="4,64", help="heads,features per head of the multi-head cases (configs[3]: 4,64)")
    ap.add_argument("--gws-wave-cut", action="store_true", help="single-weight / weightless plans over 512 / 256-byte rows cut into WAVES (what the "
                                                                "matrix-core kernels run) instead of the rule's lane groups")
    a = ap.parse_args()
    dev = torch.device("cuda")
    for item in filter(None, a.options.split(",")):
        k, v = item.split("=")
        hip.set_option(k, int(v))
    only = set(filter(None, a.only.split(",")))
    nodes, nnz = int(232_965 * a.scale), int(114_615_892 * a.scale)
    print(f"# {hip.build_info()}  nodes={nodes} nnz={nnz} options={a.options or '-'} slab_mib={a.slab_mib or 'rule'}", flush=True)
    di = powerlaw_index(nnz, nodes, 11, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(12)
    si = torch.randint(0, nodes, (nnz,), device=dev, generator=g)
    plans = {}

    def plan_for(rowbytes, wmode, H, dtype):
        R = slab.rows_per_group(wmode, H, dtype, rowbytes)
        units = 0
        if a.mh_lane_groups and wmode == 2 and rowbytes in (256, 512):
            R = slab.rows_per_group(wmode, H, dtype)
            units = int(slab._lib.load().geot_slab_units_for(0, rowbytes))
        if a.gws_wave_cut and wmode != 2 and rowbytes in (256, 512):
            R = slab.rows_per_group(2, 1, dtype, rowbytes)
            units = int(slab._lib.load().geot_slab_units_for(2, rowbytes))
        R = a.rows_per_group or R
        key = (rowbytes, R, units)
        if key not in plans:
            plans.clear()                                  # (one plan at a time: 1 GB each)
            plans[key] = slab.build_plan(si, di, nodes, nodes, rowbytes, wmode, H, rows_per_group=R, units=units, slab_bytes=int(a.slab_mib * (1 << 20)))
        return plans[key]

    def line(name, fn, extra=""):
        ms = device_ms(fn, a.iters, warmup=2)
        print(f"{name:64s} {ms:8.3f} ms   {hip.last_kernel()} {extra}", flush=True)
        return ms

    for dtype, tname in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        if tname not in a.dtypes.split(","):
            continue
        esz = 4 if dtype == torch.float32 else 2
        # ---- multi-head SpMM, H = 4 x F = 64 (configs[3])
        if not only or "mh" in only:
            H, Fh = (int(v) for v in a.mh_shape.split(","))
            x = torch.rand(nodes, H, Fh, device=dev, generator=g).to(dtype)
            w = torch.rand(nnz, H, device=dev, generator=g).to(dtype)
            out = torch.empty(nodes, H, Fh, device=dev, dtype=dtype)
            plan = plan_for(H * Fh * esz, 2, H, dtype)
            line(f"mh_spmm H={H} F={Fh} {tname} rows {H * Fh * esz} B, weights [nnz,H] in edge order, as the ABI serves them", lambda: slab.slab_spmm_out(plan, w, 2, x, out, H, Fh),
                 f"R={plan.meta['rows_per_group']} rounds={plan.meta['rounds']} units={plan.meta['units']}")
            line(f"mh_spmm H={H} F={Fh} {tname} rows {H * Fh * esz} B, weights [nnz,H] in edge order, through e_perm", lambda: slab.slab_spmm_out(plan, w, 2, x, out, H, Fh, stage_weights=False))
            wp = graph._rows_at(w, plan.tensors["e_perm"].long())   # (the library's row gather: torch's indexing of [115 M, 8] 16-bit rows is wrong on this stack)
            line(f"mh_spmm H={H} F={Fh} {tname} rows {H * Fh * esz} B, weights in PLAN order (mode 5)", lambda: slab.slab_spmm_out(plan, wp, 5, x, out, H, Fh))
            del wp
            if not only or "sddmm" in only:
                q = torch.rand(nodes, H, Fh, device=dev, generator=g).to(dtype)
                s_edge = torch.empty(nnz, H, device=dev, dtype=dtype)
                staging = torch.empty(nnz, H, device=dev, dtype=dtype)
                line(f"mh_sddmm H={H} F={Fh} {tname}, results in edge order (staged + unstage)", lambda: slab.slab_mh_sddmm_out(plan, q, x, s_edge, staging))
                line(f"mh_sddmm H={H} F={Fh} {tname}, results left in plan order", lambda: slab.slab_mh_sddmm_out(plan, q, x, None, staging))
                line(f"mh_sddmm H={H} F={Fh} {tname}, per-edge kernel", lambda: hip.mh_sddmm_coo_out(si, di, q, x, s_edge, False))
                del q, s_edge, staging
            del x, w, out
        # ---- single weight / no weight, F = 128 and F = 64
        for F in (256, 128, 64):
            if only and "gws" not in only:
                break
            if F * esz < 256 or F * esz > 512:
                continue
            x = torch.rand(nodes, F, device=dev, generator=g).to(dtype)
            w = torch.rand(nnz, device=dev, generator=g).to(dtype)
            out = torch.empty(nodes, F, device=dev, dtype=dtype)
            plan = plan_for(F * esz, 1, 1, dtype)
            line(f"gws F={F} {tname} rows {F * esz} B, weight[e] in edge order, as the ABI serves them", lambda: slab.slab_spmm_out(plan, w, 1, x, out, 1, F),
                 f"R={plan.meta['rows_per_group']} rounds={plan.meta['rounds']} units={plan.meta['units']}")
            line(f"gws F={F} {tname} rows {F * esz} B, weight[e] in edge order, through e_perm", lambda: slab.slab_spmm_out(plan, w, 1, x, out, 1, F, stage_weights=False))
            wp = graph._rows_at(w, plan.tensors["e_perm"].long())   # (the library's row gather: torch's indexing of [115 M, 8] 16-bit rows is wrong on this stack)
            line(f"gws F={F} {tname} rows {F * esz} B, weight in PLAN order (mode 4)", lambda: slab.slab_spmm_out(plan, wp, 4, x, out, 1, F))
            line(f"gs  F={F} {tname} rows {F * esz} B, no weight", lambda: slab.slab_spmm_out(plan, None, 0, x, out, 1, F))
            line(f"gs  F={F} {tname} rows {F * esz} B, mean", lambda: slab.slab_spmm_out(plan, None, 0, x, out, 1, F, reduce="mean"))
            if not only or "sddmm" in only:
                m1 = torch.rand(nodes, F, device=dev, generator=g).to(dtype)
                o = torch.empty(nnz, device=dev, dtype=dtype)
                line(f"sddmm F={F} {tname}, staged", lambda: slab.slab_sddmm_out(plan, m1, x, o, staged=True))
                del m1, o
            del x, w, wp, out
    print("# done")


if __name__ == "__main__":
    main()
