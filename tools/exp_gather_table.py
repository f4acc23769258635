#!/usr/bin/env python3
"""Why does the per-edge gather kernel read 512-byte rows at 4.6 TB/s from configs[4]'s 57 GB table and at 6.6 TB/s from
configs[2]'s 1.25 GB one?  (VERDICT round 4, weak #3.)  One edge list shape (dst power-law-ish runs of ~14.5 edges, F=128 fp32,
gather_scatter through the C ABI), sources drawn four ways:

  size      uniform over a table of S GB, S = 0.3 .. 57         -> rate against table size
  window    uniform over a contiguous W-GB window of the 57 GB table -> footprint, not the allocation's size
  page      every run of P consecutive edges draws from ONE random 2-MiB page of the 57 GB table (rows random inside it):
            DRAM-random rows, translation-friendly                -> address translation vs DRAM
  sorted    uniform over the 57 GB table, sources sorted inside every block of B consecutive edges (what an in-tile
            ordering could offer)

and the launcher's knobs at the full table: edges per lane group, loads in flight (via "unroll" where instantiated), non-temporal
row loads.  Prints one line per cell: ms, G edges/s, TB/s of row reads.

    python tools/exp_gather_table.py [--edges 100e6] [--quick]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from geot_amd import hip  # noqa: E402


def ms_of(fn, iters=4, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


def mix64(e, seed):
    m1, m2, m3 = 0x9E3779B97F4A7C15 - (1 << 64), 0xBF58476D1CE4E5B9 - (1 << 64), 0x94D049BB133111EB - (1 << 64)
    x = e * m1 + seed
    x = (x ^ ((x >> 30) & 0x3FFFFFFFF)) * m2
    x = (x ^ ((x >> 27) & 0x1FFFFFFFFF)) * m3
    x = x ^ ((x >> 31) & 0x1FFFFFFFF)
    return x & 0x7FFFFFFFFFFFFFFF


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--edges", type=float, default=100e6)
    ap.add_argument("--feat", type=int, default=128)
    ap.add_argument("--table-gb", type=float, default=56.9)
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    nnz, F = int(args.edges), args.feat
    rb = 4 * F
    rows_full = int(args.table_gb * 1e9 / rb)
    K = int(nnz / 14.5)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    # run lengths: randomised rounding of a mild power law, like bench.py's global list
    w = torch.arange(1, K + 1, device=dev, dtype=torch.float64) ** (-1.0 / 1.5)
    w *= nnz / float(w.sum())
    cnt = torch.floor(w[torch.randperm(K, generator=g, device=dev)] + torch.rand(K, generator=g, device=dev, dtype=torch.float64)).long()
    cnt[-1].clamp_(min=1)
    dst = torch.repeat_interleave(torch.arange(K, device=dev), cnt)
    nnz = dst.numel()
    del w, cnt
    src = torch.empty(rows_full, F, device=dev)
    src.uniform_(generator=g)
    out = torch.empty(K, F, device=dev)
    e = torch.arange(nnz, device=dev)
    h = mix64(e, 77)
    print(f"# {nnz} edges -> {K} rows, F={F}, table {rows_full} rows = {rows_full * rb / 1e9:.1f} GB; {hip.build_info()}", flush=True)

    def run(label, si, table=src):
        t = ms_of(lambda: hip.gather_scatter_out(si, dst, table, out))
        print(f"{label:58s} {t:8.3f} ms  {nnz / t / 1e6:7.2f} G edges/s  {nnz * rb / t / 1e9:6.2f} TB/s of rows   {hip.last_kernel()}", flush=True)
        return t

    sizes = [0.3, 1.25, 4, 8, 16, 32, args.table_gb] if not args.quick else [1.25, args.table_gb]
    for s in sizes:
        r = min(rows_full, int(s * 1e9 / rb))
        run(f"size    uniform over the first {s:5.2f} GB", h % r)
    for wgb in ([1.25, 4, 8, 16] if not args.quick else [4]):
        r = int(wgb * 1e9 / rb)
        base = (rows_full - r) // 2
        run(f"window  uniform over {wgb:5.2f} GB in the middle of the table", base + h % r)
    rows_per_page = (2 << 20) // rb
    pages = rows_full // rows_per_page
    for P in ([64, 640, 5120, 65536] if not args.quick else [640]):
        page = mix64(e // P, 99) % pages
        run(f"page    {P:6d} consecutive edges share one 2-MiB page", page * rows_per_page + h % rows_per_page)
    for gb in ([1, 8] if not args.quick else []):          # regions larger than a page: edges of a block share a region of gb GiB
        rr = int(gb * (1 << 30) / rb)
        regions = max(1, rows_full // rr)
        for P in (5120, 655360):
            reg = mix64(e // P, 101) % regions
            run(f"region  {P:6d} consecutive edges share one {gb}-GiB region", reg * rr + h % rr)
    si_full = h % rows_full
    for B in ([80, 640, 5120] if not args.quick else [640]):
        pad = (-nnz) % B
        x = torch.cat([si_full, si_full.new_full((pad,), rows_full - 1)]).view(-1, B).sort(dim=1).values.reshape(-1)[:nnz].contiguous()
        run(f"sorted  sources ascending inside blocks of {B} edges", x)
    # launcher knobs at the full table
    base = run("knobs   rule", si_full)
    for cg in (32, 48, 64, 128, 160, 256):
        hip.tune(edges_per_group=cg)
        run(f"knobs   {cg} edges per lane group", si_full)
    hip.tune()
    for nt in (1,):
        hip.tune(nontemporal=nt)
        run(f"knobs   nontemporal={nt}", si_full)
    hip.tune()
    for lpr in (6,):
        hip.tune(lpr_log2=lpr)
        run(f"knobs   lanes per row log2 = {lpr}", si_full)
    hip.tune()
    hip.set_option("xcd", 0)
    run("knobs   xcd = 0", si_full)
    hip.set_option("xcd", 1)
    hip.set_option("unroll", 16)
    run("knobs   16 row loads in flight per lane", si_full)
    for cg in (32, 64, 128):
        hip.tune(edges_per_group=cg)
        run(f"knobs   16 loads in flight, {cg} edges per lane group", si_full)
    hip.tune()
    run("knobs   16 loads in flight, 1.25 GB table", h % int(1.25e9 / rb))
    hip.set_option("unroll", 0)
    print(f"# done {time.strftime('%T')}")


if __name__ == "__main__":
    main()
