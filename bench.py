#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

Metric (BASELINE.json): aggregated edges/sec + HBM GB/s, index_scatter feat=64 sorted sum.
Workload (BASELINE.json configs[1]): synthetic power-law 10M edges -> 1M nodes, feat=64, fp32,
int64 index, generated on the device from fixed seeds (SURVEY.md section 8d generator).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself: the
parent touches no GPU, spawns `python -m torch.distributed.run ... bench.py --gpus N ...` as a CHILD process,
relays rank 0's JSON line and exits with the child's code.

A step = one call of the drop-in operator ``geot.index_scatter(0, src, index, 'sum', True)`` on the
resident inputs: the D2H read of index[-1] (the reference's row rule), the output allocation, the
tile kernel and the fix-up kernel.  N > 1 = weak scaling: every rank reduces its own 10M-edge shard
of a dst-sorted edge list whose neighbouring shards share their boundary key, and the partial
boundary rows are exchanged with one small RCCL all_gather per step (geot_amd/sharding.py).

--workload cfg5 (BASELINE.json configs[4]): gather_scatter, papers100M scale, feat=128; every rank holds 1/8 of
the 1.6157 B edges and of the 111.06 M dst rows, src (all 111.06 M nodes = 56.9 GB) replicated; N = 8 is the full
configuration, N = 1 one GPU's shard.  Not the headline line.

One JSON line on rank 0.  `roofline` is the tile kernel alone (HIP events around it on its stream,
over K extra steps); `cpu_baseline` is the reference's own CPU index_scatter (oracle/_ref, compiled
from /root/reference in the build container) - or the oracle port when that library is absent -
timed on this host on the same full-size inputs; `secondary` (N = 1, headline workload) holds the two other
single-GPU configurations of BASELINE.json, bounded to a couple of seconds: gather_weight_scatter at
ogbn-products scale next to rocSPARSE's CSR SpMM on the same matrix, and mh_spmm at Reddit scale.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NNZ, KEYS, FEAT = 10_000_000, 1_000_000, 64
HBM_PEAK_GBPS = 8000.0           # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
HBM_COPY_GBPS = 6290.0           # measured float4 copy ceiling (same table)
METRIC = "aggregated edges/sec + HBM GB/s, index_scatter feat=64 sorted sum"
# BASELINE.json configs[4]: ogbn-papers100M scale, one GPU's share of 8
CFG5_NODES, CFG5_EDGES, CFG5_FEAT = 111_059_956, 1_615_685_872, 128


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=["cfg2", "cfg5"], default="cfg2")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink cfg5 / the secondary workloads (tests)")
    ap.add_argument("--src-sharding", choices=["replicated", "node"], default="replicated",
                    help="cfg5 at N > 1: source features replicated on every rank (56.9 GB each) or sharded by contiguous node ranges, the rows a "
                         "rank's edge range references fetched per step with one all_to_all_single (SURVEY 8e, second option)")
    ap.add_argument("--strong", action="store_true", help="N > 1: strong scaling (the 10 M-edge problem split over the ranks)")
    ap.add_argument("--cuts", choices=["equal", "aligned"], default="equal",
                    help="N > 1: equal = neighbouring shards share their boundary key (one small all_gather of partial rows per "
                         "step); aligned = segment-aligned cuts, no data-path collective")
    ap.add_argument("--collective", choices=["all_gather", "reduce_scatter"], default="all_gather",
                    help="N > 1, equal cuts: how the boundary-row partials travel in the TIMED steps (the other form is timed "
                         "in a few extra steps and reported beside it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--only-secondary", default="", help="comma-separated entries of `secondary` to run (profiling sessions)")
    ap.add_argument("--secondary", choices=["default", "all"], default="default",
                    help="default: the entries the compact line's `extras` quote (configs[0], [2] vs rocSPARSE, [3] fp32 + bf16, [4]'s shard); "
                         "all: every stand-in and the training steps as well (profiling sessions; minutes)")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "bench_secondary.json"),
                    help="where the FULL record (headline + every secondary workload, with notes) is written; stdout carries only the compact line")
    ap.add_argument("--full", action="store_true", help="print the full record as the stdout line instead of the compact one (tools/profile_round.sh)")
    return ap.parse_args()


def spawn_ranks(args):
    """N > 1 called as a plain script: start one process per GPU as CHILDREN (never re-exec: the parent has not
    touched the GPU and never will), relay rank 0's JSON line, propagate failure."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        sys.stderr.write(p.stdout)
        raise SystemExit(p.returncode or 1)
    print(lines[-1])
    raise SystemExit(0)


def powerlaw_index(nnz, keys, seed, device):
    """Sorted int64 keys, w_k ~ rank^(-1/1.5), ranks randomly permuted, index[-1] = keys-1."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    w = torch.arange(1, keys + 1, device=device, dtype=torch.float64) ** (-1.0 / 1.5)
    cdf = torch.cumsum(w, 0)
    perm = torch.randperm(keys, generator=g, device=device)
    u = torch.rand(nnz, generator=g, device=device, dtype=torch.float64) * cdf[-1]
    r = torch.searchsorted(cdf, u).clamp_(max=keys - 1)
    idx = perm[r].sort().values.contiguous()
    idx[-1] = keys - 1
    idx[0] = 0                   # shard r's first key = shard r-1's last key (boundary segment)
    return idx


def _mix64(e, seed):
    """splitmix64-style hash of int64 edge numbers (wrapping int64 arithmetic), 63-bit non-negative result: edge e of the
    global list gets the same source whatever rank generates it and however many ranks there are."""
    m1, m2, m3 = 0x9E3779B97F4A7C15 - (1 << 64), 0xBF58476D1CE4E5B9 - (1 << 64), 0x94D049BB133111EB - (1 << 64)
    x = e * m1 + seed
    x = (x ^ ((x >> 30) & 0x3FFFFFFFF)) * m2
    x = (x ^ ((x >> 27) & 0x1FFFFFFFFF)) * m3
    x = x ^ ((x >> 31) & 0x1FFFFFFFF)
    return x & 0x7FFFFFFFFFFFFFFF


def global_list_shard(rows_global, nnz_target, src_nodes, world, rank, cuts, seed, device):
    """Rank `rank`'s contiguous slice of ONE global dst-sorted power-law edge list, generated WITHOUT materialising the list:
    every rank derives the same per-row edge counts from the same seed (w_k ~ rank^(-1/1.5), ranks randomly permuted,
    randomised rounding of the expected count - rows_global values, cheap), cuts the edge range [0, nnz) like
    geot_amd.sharding.equal_edge_cuts / segment_aligned_cuts would, and expands only its own rows; the source of edge e is a
    hash of e.  Returns (dst_index_local, src_index, first_key, my_rows_upper, nnz_global, cut_edges):
    dst_index_local = global key - first_key (what sharded_gather_scatter's key_offset expects)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    w = torch.arange(1, rows_global + 1, device=device, dtype=torch.float64) ** (-1.0 / 1.5)
    w *= nnz_target / float(w.sum())
    perm = torch.randperm(rows_global, generator=g, device=device)
    counts = torch.floor(w[perm] + torch.rand(rows_global, generator=g, device=device, dtype=torch.float64)).to(torch.int64)
    del w, perm
    counts[-1].clamp_(min=1)                                  # the last row exists: rows = index[-1] + 1 = rows_global
    ends = torch.cumsum(counts, 0)                            # ends[k] = first edge of row k + 1
    nnz = int(ends[-1].item())
    edges = [(nnz * r) // world for r in range(world + 1)]
    if cuts == "aligned":                                     # move every cut to the next row start (no key shared by two ranks)
        probe = torch.tensor(edges[1:-1], device=device, dtype=torch.int64)
        k = torch.searchsorted(ends, probe, right=True)       # row holding edge e
        starts = torch.where(k > 0, ends[(k - 1).clamp_(min=0)], torch.zeros_like(probe))
        snapped = torch.where(starts == probe, probe, ends[k])
        edges = [0] + [int(v) for v in snapped.tolist()] + [nnz]
        for i in range(1, len(edges)):
            edges[i] = max(edges[i], edges[i - 1])
    e0, e1 = edges[rank], edges[rank + 1]
    if e1 <= e0:
        raise SystemExit(f"rank {rank}: empty shard (too few edges for {world} ranks)")
    k0 = int(torch.searchsorted(ends, torch.tensor([e0], device=device), right=True).item())
    k1 = int(torch.searchsorted(ends, torch.tensor([e1 - 1], device=device), right=True).item())
    mine = counts[k0:k1 + 1].clone()
    mine[0] = int(ends[k0].item()) - e0                       # the part of row k0 that lies in [e0, e1)
    if k1 > k0:
        mine[-1] = e1 - (int(ends[k1 - 1].item()))
    else:
        mine[0] = e1 - e0
    del counts, ends
    dst_local = torch.repeat_interleave(torch.arange(k1 - k0 + 1, device=device, dtype=torch.int64), mine, output_size=e1 - e0)
    del mine
    src_index = _mix64(torch.arange(e0, e1, device=device, dtype=torch.int64), seed + 1) % src_nodes
    return dst_local, src_index, k0, k1 - k0 + 1, nnz, edges


def block_model(nodes, nnz, intra, device, seed=3):
    """A graph WITH community structure whose ids are shuffled, as public datasets ship: communities of 2-20 k nodes, power-law
    destinations, `intra` of a node's edges from inside its community, every id randomly permuted; dst-sorted int64 COO.
    Returns (src_index, dst_index, rank of the true community order, number of communities)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    gs = torch.Generator()
    gs.manual_seed(seed)
    sizes, left = [], nodes
    while left > 0:
        s = min(int(torch.randint(2000, 20001, (1,), generator=gs).item()), left)
        sizes.append(s)
        left -= s
    sizes_t = torch.tensor(sizes, device=device)
    starts = torch.cumsum(sizes_t, 0) - sizes_t
    comm = torch.repeat_interleave(torch.arange(len(sizes), device=device), sizes_t)
    w = torch.arange(1, nodes + 1, device=device, dtype=torch.float64) ** (-1.0 / 1.5)
    cdf = torch.cumsum(w, 0)
    pr = torch.randperm(nodes, generator=g, device=device)
    u = torch.rand(nnz, generator=g, device=device, dtype=torch.float64) * cdf[-1]
    dst = pr[torch.searchsorted(cdf, u).clamp_(max=nodes - 1)]
    del w, cdf, u
    c = comm[dst]
    inside = torch.rand(nnz, device=device, generator=g) < intra
    src_in = starts[c] + (torch.rand(nnz, device=device, generator=g) * sizes_t[c]).long().clamp_(max=nodes - 1)
    src = torch.where(inside, src_in, torch.randint(0, nodes, (nnz,), device=device, generator=g))
    del c, inside, src_in
    shuffle = torch.randperm(nodes, generator=g, device=device)
    dst_s, src_s = shuffle[dst], shuffle[src]
    order = torch.argsort(dst_s, stable=True)
    truth = torch.empty(nodes, dtype=torch.int64, device=device)
    truth[shuffle] = torch.arange(nodes, device=device)
    di = dst_s[order].contiguous()
    di[-1] = nodes - 1
    return src_s[order].contiguous(), di, truth, len(sizes)


def algorithmic_bytes(nnz, feat, rows):
    """each src row and index read once, each dst row written once."""
    return nnz * (4 * feat + 8) + rows * 4 * feat


def cpu_baseline(index, src, budget_s=25.0):
    """Reference CPU index_scatter on this host, same inputs (bounded: at most ~budget_s seconds)."""
    idx = index.cpu().numpy()
    s = src.cpu().numpy()
    cores = os.cpu_count() or 1
    res = {"unit": "edges/s", "sample": f"full workload ({NNZ} edges x {FEAT} feat -> {KEYS} rows), best of <=3 passes"}

    def best_of(fn, reps=3):
        best, spent = None, 0.0
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            spent += dt
            if spent > budget_s / 3:
                break
        return best

    from oracle import api as oracle, ref
    # thread counts to try: all cores is not always the fastest on a many-socket host
    tries = sorted({cores, min(cores, 64), min(cores, 16)}, reverse=True)
    port_t, port_n = min((best_of(lambda n=n: oracle.index_scatter_3pass(idx, s, "sum", threads=n, rows=KEYS), reps=2), n)
                         for n in tries)
    res["port_edges_per_s"] = NNZ / port_t
    res["port_threads"] = port_n
    if ref.available(omp=True):
        t_omp, n_omp = min((best_of(lambda n=n: ref.index_scatter_cpu(idx, s, omp=True, threads=n, rows=KEYS), reps=2), n)
                           for n in tries)
        res.update(value=NNZ / t_omp, cores=n_omp, kind="reference", host_cores=cores,
                   note="reference csrc/cpu/index_scatter_cpu.cpp compiled in place with -fopenmp, best of "
                        f"{tries} threads; as shipped it sums src[index[n]] (touches only K distinct "
                        "rows: optimistic for the CPU) and is single-threaded (setup.py has no -fopenmp)")
        if ref.available(omp=False):
            t_ser = best_of(lambda: ref.index_scatter_cpu(idx, s, omp=False, rows=KEYS), reps=2)
            res["reference_as_shipped_serial_edges_per_s"] = NNZ / t_ser
    else:
        res.update(value=NNZ / port_t, cores=port_n, kind="port", host_cores=cores,
                   note="oracle/_ref absent: timed the C restatement of the reference's 3-pass algorithm (OpenMP)")
    return res


def device_ms(fn, iters, warmup=2):
    """Device time per call from HIP events on the current stream (the stream the *_out doorway launches on)."""
    import torch
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


SECONDARY_DEFAULT = ("cfg1", "gws_cfg3", "mh_spmm_cfg4", "mh_spmm_cfg4_bf16")      # what the compact line's `extras` quote
SECONDARY = ("cfg1", "gws_cfg3", "gws_cfg3_local", "gws_cfg3_powerlaw_src", "gws_cfg3_blockmodel", "mh_spmm_cfg4",
             "mh_spmm_cfg4_powerlaw_src", "mh_spmm_cfg4_coalesced", "gws_cfg3_bf16", "mh_spmm_cfg4_bf16", "gws_train_step_cfg4_graph",
             "mh_train_step_cfg4_graph")


def profiled(entry):
    """PMC-derived fabric traffic of a secondary workload's kernel, from the round's profiling session (committed under
    profiles/; tools/profile_round.sh + tools/derive_traffic.py) - a recorded measurement of the same kernel on the same
    workload, not a measurement of this run: labelled with its source."""
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        f = os.path.join(ROOT, "profiles", rnd, "gather_kernels.json")
        try:
            k = json.load(open(f))["kernels"].get(entry)
        except Exception:
            k = None
        if k and k.get("fabric_bytes_per_launch"):
            return {"traffic": k["fabric_bytes_per_launch"], "traffic_over_compulsory": k.get("traffic_over_compulsory"),
                    "l2_hit_rate": k.get("l2_hit_rate"), "kernel_ms_under_rocprofv3": k.get("average_ms"),
                    "traffic_source": f"profiles/{rnd}/gather_kernels.json[{entry!r}] (rocprofv3 --pmc, same workload, recorded session)"}
    return {"traffic": None, "traffic_source": None}


def profiled_kernel_time(entry):
    """Kernel duration of a launch-bound workload from the round's rocprofv3 kernel trace (committed; tools/derive_traffic.py)."""
    for rnd in ("r06", "r05"):
        try:
            k = json.load(open(os.path.join(ROOT, "profiles", rnd, "gather_kernels.json")))["kernels"].get(entry)
        except Exception:
            k = None
        if k and k.get("average_ms"):
            return {"kernel_us": k["average_ms"] * 1e3, "second_launch_us": k.get("second_launch_us"),
                    "source": f"profiles/{rnd}/gather_kernels.json[{entry!r}] (rocprofv3 --kernel-trace of this workload, recorded session, {k.get('calls')} launches)"}
    return None


def secondary(dev, scale=1.0, iters=5, only=None):
    """BASELINE.json configs[2] and configs[3] on synthetic stand-ins (same generator as the headline workload).
    Roofline = SURVEY.md 8(d) COMPULSORY bytes / kernel time / 8 TB/s - re-gathered rows served by the caches are not
    credited.  configs[2] comes twice: sources uniform-random (the worst case: no locality at all, 13.5x compulsory traffic)
    and sources within +-2000 rows of the destination (`gws_cfg3_local`: what a graph with community structure, like
    ogbn-products itself, offers the XCD-aware tile order), each next to rocSPARSE's best CSR algorithm on the same matrix.
    `*_bf16`: the same graphs with bfloat16 storage (fp32 accumulation) - ADDITIONAL lines, never the fp32 headline."""
    import torch
    import geot_amd as geot
    from geot_amd import hip, ops, slab
    from tools import rocsparse
    want = set(only) if only else set(SECONDARY_DEFAULT)
    res = {}

    def sources(kind, di, nodes, g):
        """src_index of a stand-in graph: "uniform" = no structure at all (every row gather misses: the worst case); "local" =
        within +-2000 rows of the destination (nodes numbered by community); "powerlaw" = SURVEY.md 8(d)'s "permuted endpoints":
        src_index = dst_index[randperm(nnz)] - sources exactly as skewed as the destinations (the named graphs are symmetrised:
        a hub is a hub on both sides), ids unordered."""
        nnz = di.numel()
        if kind == "local":
            return (di + torch.randint(-2000, 2001, (nnz,), device=dev, generator=g)).clamp_(0, nodes - 1)
        if kind == "powerlaw":
            return di[torch.randperm(nnz, device=dev, generator=g)].contiguous()
        return torch.randint(0, nodes, (nnz,), device=dev, generator=g)

    SRC_DESC = {"uniform": "uniform-random src", "local": "sources within +-2000 rows of the destination (uniform offset, clamped)",
                "powerlaw": "src = a random permutation of the dst endpoints (as skewed as the destinations, ids unordered)"}

    def gws(name, kind, dtype):
        nodes, nnz, F = int(2_449_029 * scale), int(123_718_280 * scale), 128
        esize = 4 if dtype == torch.float32 else 2
        di = powerlaw_index(nnz, nodes, 7, dev)
        g = torch.Generator(device=dev)
        g.manual_seed(8)
        si = sources(kind, di, nodes, g)
        w = torch.rand(nnz, device=dev, generator=g).to(dtype)
        x = torch.rand(nodes, F, device=dev, generator=g).to(dtype)
        out = torch.empty(nodes, F, device=dev, dtype=dtype)
        ms = device_ms(lambda: hip.gather_weight_scatter_out(si, di, w, x, out), iters)
        kernel = hip.last_kernel()
        op_ms = device_ms(lambda: geot.gather_weight_scatter(si, di, w, x), iters)
        uniq = int(torch.unique(si).numel())
        comp = nnz * (16 + esize) + uniq * esize * F + nodes * esize * F
        entry = {"workload": f"gather_weight_scatter, power-law dst / {SRC_DESC[kind]}, {nodes} nodes, {nnz} edges, feat={F}, "
                             f"{str(dtype).split('.')[-1]}, int64 COO dst-sorted (stand-in of ogbn-products)",
                 "kernel": kernel, "kernel_ms": ms, "op_ms_with_row_rule": op_ms, "edges_per_s": nnz / ms * 1e3,
                 "compulsory_bytes": comp, "distinct_src_rows": uniq,
                 "roofline": {"bound": "hbm", "achieved": comp / ms / 1e6, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": comp / ms / 1e6 / HBM_PEAK_GBPS, **profiled(name)},
                 "gathered_row_bytes": nnz * esize * F, "gathered_row_gbps": nnz * esize * F / ms / 1e6}
        if kind == "uniform":
            # the yardstick of a per-edge gather over THIS table (configs[4]'s, bench_cfg5): the box's own rate for uniform-random rows of
            # it with nothing else going on, and with one row written per `run` rows read (run = edges per output row) - the table is
            # 1.25 GB (0.63 in bf16), so a part of every probe's reads comes out of the 256 MiB Infinity Cache, as the kernel's do
            try:
                rb = hip.profile_box_rows(x, out=torch.empty_like(out), run=max(1, round(nnz / max(nodes, 1))))
                rate = nnz * esize * F / ms / 1e6
                entry["roofline"].update(box_random_row_gbps=rb["best_gbps"], box_random_row_gbps_with_write_mix=rb["mix_row_gbps"],
                                         row_gather_frac_of_box_random_row=rate / rb["best_gbps"] if rb["best_gbps"] else None,
                                         row_gather_frac_of_box_row_mix=rate / rb["mix_row_gbps"] if rb["mix_row_gbps"] else None)
            except Exception as e:  # noqa: BLE001
                entry["roofline"]["box_random_row_error"] = repr(e)
        if dtype == torch.float32:
            try:
                best, table, y = rocsparse.best_csr_spmm(di, si, w, x, nodes, iters=max(2, iters // 2))
                hip.gather_weight_scatter_out(si, di, w, x, out)
                torch.cuda.synchronize()
                entry.update(rocsparse_best_ms=best["ms"], rocsparse_best_algorithm=best["algorithm"],
                             rocsparse_index_width="int32 CSR (geot: int64 COO)", rocsparse_preprocess="excluded from timing",
                             rocsparse_preprocess_buffer_bytes=best.get("preprocess_buffer_bytes"),
                             rocsparse_all=table, speedup_vs_rocsparse_best=best["ms"] / ms,
                             max_rel_diff_vs_rocsparse=float(((y - out).abs().max() / out.abs().max()).item()))
                del y
            except Exception as e:               # the comparator must never take the measurement down with it
                entry["rocsparse_error"] = repr(e)
        res[name] = entry
        del di, si, w, x, out
        torch.cuda.empty_cache()

    def mh(name, dtype, kind="uniform", coalesced=False):
        nodes, nnz, H, F = int(232_965 * scale), int(114_615_892 * scale), 4, 64
        esize = 4 if dtype == torch.float32 else 2
        di = powerlaw_index(nnz, nodes, 11, dev)
        g = torch.Generator(device=dev)
        g.manual_seed(12)
        si = sources(kind, di, nodes, g)
        if coalesced:       # sources ascending inside every dst row: what a CSR / torch_geometric's coalesce() / a stable sort by dst leaves
            si = (torch.sort(di * nodes + si).values % nodes).contiguous()
        w = torch.rand(nnz, H, device=dev, generator=g).to(dtype)
        x = torch.rand(nodes, H, F, device=dev, generator=g).to(dtype)
        out = torch.empty(nodes, H, F, device=dev, dtype=dtype)
        ms_gather = device_ms(lambda: hip.mh_spmm_out(si, di, w, x, out, False), iters)     # per-edge gather kernel (seg_tile_kernel)
        wt = w.t().contiguous()
        ms_t = device_ms(lambda: hip.mh_spmm_out(si, di, wt, x, out, True), iters)
        # Phase A by itself (csrc/seg_plan.hip, device code): wall time of one plan build on an idle stream, synchronised on
        # both sides - the first plan this process builds, then a rebuild
        phase_a, plan_bytes = [], None
        if slab.worthwhile(nnz, nodes, nodes, H * F * esize):
            R = slab.rows_per_group(2, H, dtype)
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                plan = slab.build_plan(si, di, nodes, nodes, H * F * esize, 2, H, rows_per_group=R)
                torch.cuda.synchronize()
                phase_a.append((time.perf_counter() - t0) * 1e3)
                plan_bytes = plan.nbytes()
                del plan
        # the operator as dispatched: a graph this dense is re-arranged once (Phase A, on the second call with the same
        # edge list) and then served by the source-blocked kernel (csrc/seg_slab.hip); device time, steady state
        st0 = ops.stats()
        ms = device_ms(lambda: geot.mh_spmm(si, di, w, x), iters, warmup=3)
        kernel = hip.last_kernel()
        st1 = ops.stats()
        # the same with the content guard off: what the fingerprint re-check of the remembered plan costs (csrc/seg_guard.hip)
        guard_was = ops.set_option("content_guard", 0)
        ms_unguarded = device_ms(lambda: geot.mh_spmm(si, di, w, x), iters, warmup=1)
        ops.set_option("content_guard", guard_was)
        # ... and through the opt-in static-graph handle (owns its index clones and its plan: no fingerprint, no lookup)
        handle = geot.Graph(si, di, num_src=nodes, num_dst=nodes)
        ms_handle = device_ms(lambda: handle.mh_spmm(w, x), iters, warmup=4)
        # ... with the (static) coefficients handed over in plan order once - what the dispatched operator's cache does behind the
        # guard, here the caller's explicit choice
        w_plan = handle.plan_order(w, x)
        ms_handle_static = device_ms(lambda: handle.mh_spmm(w_plan, x), iters, warmup=2)
        del w_plan
        handle_stats = dict(handle.stats)
        del handle
        hip.mh_spmm_out(si, di, w, x, out, False)
        torch.cuda.synchronize()
        diff = float(((geot.mh_spmm(si, di, w, x).float() - out.float()).abs().max() / out.float().abs().max()).item())
        slab_used = st1["slab_calls"] > st0["slab_calls"]
        uniq = int(torch.unique(si).numel())
        comp = nnz * (16 + esize * H) + uniq * esize * H * F + nodes * esize * H * F
        res[name] = {
            "workload": f"mh_spmm, power-law dst / {SRC_DESC[kind]}{', ascending inside every dst row (coalesced COO)' if coalesced else ''}, {nodes} nodes, {nnz} edges, heads={H} feat={F}, "
                        f"{str(dtype).split('.')[-1]} (stand-in of Reddit)",
            "kernel_ms": ms,
            "content_guard": "on: every call re-reads both index arrays and compares their fingerprint with the plan's (kernel_ms includes it)",
            "kernel_ms_without_content_guard": ms_unguarded,
            "kernel_ms_with_graph_handle": ms_handle_static, "kernel_ms_with_graph_handle_edge_order_weights": ms_handle,
            "graph_handle": "geot_amd.Graph: owns index clones and plans, no fingerprint, no cache lookup.  kernel_ms_with_graph_handle: coefficients handed over in "
                            "plan order once (Graph.plan_order) - like-for-like with kernel_ms_without_content_guard, whose host-layer cache keeps a static "
                            "weight tensor in plan order; ..._edge_order_weights: a fresh [nnz, H] tensor in edge order every call, read through the plan's edge "
                            "permutation in the row loop",
            "graph_handle_stats": handle_stats,
            "kernel": kernel + (" (+ memset, combine)" if slab_used else ""),     # what the operator launched (geot_last_kernel)
            "plan_trial_ms": {"plan": st1["trial_plan_us"] / 1e3, "per_edge": st1["trial_edges_us"] / 1e3} if st1["plan_trials"] > st0["plan_trials"] else None,
            "source_blocked_path": slab_used,
            "phase_a": "device builder csrc/seg_plan.hip; wall ms of one build on an idle stream, synchronised before and after",
            "phase_a_ms_once_per_edge_list": phase_a[0] if phase_a else None,          # the first plan this process builds
            "phase_a_ms_rebuilt_in_a_warm_process": min(phase_a[1:]) if len(phase_a) > 1 else None,
            "phase_a_host_ms_as_dispatched": (st1["plan_us"] - st0["plan_us"]) / 1e3,   # host side of the second call (stage 3 runs on)
            "plan_bytes": plan_bytes,
            "kernel_ms_per_edge_gather": ms_gather, "kernel_ms_per_edge_gather_head_major_weights": ms_t,
            "speedup_vs_per_edge_gather": ms_gather / ms, "max_rel_diff_between_the_two_kernels": diff,
            "edges_per_s": nnz / ms * 1e3, "compulsory_bytes": comp, "distinct_src_rows": uniq,
            "roofline": {"bound": "hbm", "achieved": comp / ms / 1e6, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": comp / ms / 1e6 / HBM_PEAK_GBPS, **profiled(name)},
            "gathered_row_bytes": nnz * esize * H * F, "gathered_row_gbps": nnz * esize * H * F / ms / 1e6}
        del di, si, w, wt, x, out
        ops.clear_caches()
        torch.cuda.empty_cache()

    def cfg1(name):
        """BASELINE.json configs[0]: the reference's own CPU-runnable case (test/test_index_scatter.py scaled to 100 k x 32 ->
        10 k segments; SURVEY.md 8(d) generator).  Launch-bound on the GPU: microseconds per call AS DISPATCHED (the drop-in
        operator: read-back of index[-1], allocation, two launches), the kernel alone, and the reference's CPU kernel on the
        same inputs."""
        nnz, K, F = 100_000, 10_000, 32
        g = torch.Generator(device=dev)
        g.manual_seed(0)
        index = torch.randint(0, K, (nnz,), device=dev, generator=g).sort().values.contiguous()
        index[-1] = K - 1
        g.manual_seed(1)
        src = torch.rand(nnz, F, device=dev, generator=g)
        for _ in range(20):
            out = geot.index_scatter(0, src, index, "sum", True)
        torch.cuda.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            out = geot.index_scatter(0, src, index, "sum", True)
        torch.cuda.synchronize()
        wall_us = (time.perf_counter() - t0) / n * 1e6
        hip.profile_enable(True)
        hip.profile_reset()
        for _ in range(50):
            out = geot.index_scatter(0, src, index, "sum", True)
        torch.cuda.synchronize()
        prof = hip.profile_read()
        hip.profile_enable(False)
        kernel = hip.last_kernel()
        # HIP events around a ~6 us kernel bracket its launch gap too (round 4: "kernel" 12.5 + "fix-up" 6.9 us inside a 12.5 us call): the
        # kernel's own duration is the round's rocprofv3 kernel trace of this very workload (profiles/rNN/gather_kernels.json["cfg1"],
        # tools/profile_round.sh), labelled as recorded; the live event brackets stay beside it under their real name
        b_us = prof["main_ms"] / max(prof["calls"], 1) * 1e3
        f_us = prof["fixup_ms"] / max(prof["calls"], 1) * 1e3
        rec = profiled_kernel_time("cfg1")
        # LIVE kernel time: the launcher's two kernels replayed back to back out of a captured HIP graph (no host in the loop, launch gaps
        # shrink to the graph's own dispatch) - an upper bound of the tile kernel + second launch that moves with the binary and the
        # box; the recorded rocprofv3 figure of the profiling session stays beside it under its own name
        k_us = b_us
        try:
            out_g = torch.empty(K, F, device=dev)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    hip.index_scatter_out(index, src, out_g)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for _ in range(20):
                    hip.index_scatter_out(index, src, out_g)
            g_ms = device_ms(graph.replay, 20, warmup=3)
            k_us = g_ms / 20 * 1e3
            live_how = "20 calls of the *_out doorway captured in one HIP graph, replayed 20 times, HIP events around the replays: tile kernel + second launch + the graph's dispatch gap"
            del graph
        except Exception as e:  # noqa: BLE001
            live_how = f"HIP events around the launch inside the library (graph capture failed: {e!r}); includes the launch gap"
        alg = algorithmic_bytes(nnz, F, K)
        entry = {"workload": "index_scatter dim=0 sum, sorted, 100k src rows x feat=32 -> 10k dst segments, fp32, int64 index (BASELINE.json configs[0])",
                 "us_per_call_as_dispatched": wall_us, "calls_timed": n,
                 "kernel": kernel, "kernel_us": k_us, "kernel_us_source": live_how,
                 "recorded_kernel_us": rec["kernel_us"] if rec else None, "recorded_kernel_us_source": rec["source"] if rec else None,
                 "event_bracket_us": {"tile_kernel": b_us, "second_launch": f_us, "note": "brackets of HIP events, launch gaps included"},
                 "edges_per_s": nnz / wall_us * 1e6, "algorithmic_bytes": alg,
                 "roofline": {"bound": "hbm", "achieved": alg / (k_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": alg / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                              "traffic_source": None, "note": "launch-bound: 391 tiles on 256 CUs; the call is host + launch latency"}}
        try:
            from oracle import api as oracle, ref
            idx, s_np = index.cpu().numpy(), src.cpu().numpy()
            cores = os.cpu_count() or 1

            def best(fn, reps=5):
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    fn()
                    ts.append(time.perf_counter() - t0)
                return min(ts)
            cb = {"unit": "us per call", "sample": "the full configs[0] workload, best of 5 calls", "host_cores": cores}
            if ref.available(omp=True):
                t, nthr = min((best(lambda n=n: ref.index_scatter_cpu(idx, s_np, omp=True, threads=n, rows=K)), n)
                              for n in sorted({min(cores, 16), min(cores, 64), 1}))
                cb.update(value=t * 1e6, cores=nthr, kind="reference",
                          note="reference csrc/cpu/index_scatter_cpu.cpp compiled in place with -fopenmp, best thread count of {1, 16, 64}")
                if ref.available(omp=False):
                    cb["reference_as_shipped_serial_us"] = best(lambda: ref.index_scatter_cpu(idx, s_np, omp=False, rows=K)) * 1e6
            else:
                t = best(lambda: oracle.index_scatter_3pass(idx, s_np, "sum", threads=min(cores, 16), rows=K))
                cb.update(value=t * 1e6, cores=min(cores, 16), kind="port")
            got = out.cpu().numpy()
            hi = oracle.index_scatter(idx, s_np, acc64=True)
            entry["max_rel_err_vs_oracle_f64"] = float(abs(got - hi).max() / abs(hi).max())
            entry["cpu_baseline"] = cb
            entry["speedup_vs_cpu_reference"] = cb["value"] / wall_us
        except Exception as e:
            entry["cpu_baseline"] = {"error": repr(e)}
        res[name] = entry

    def blockmodel(name, mode="both"):        # mode "asis" / "renum": one half only (profiling passes: per-kernel averages stay pure)
        """configs[2]'s size on a graph that HAS community structure but ships with shuffled ids (what public datasets look like;
        VERDICT r03 item 4): the operator as it is, rocSPARSE's best, and geot_amd.reorder - a one-time device-built node order
        (label propagation), then per call: permute x in, the same kernels on the renumbered list, permute y out."""
        from geot_amd import reorder
        nodes, nnz, F = int(2_449_029 * scale), int(123_718_280 * scale), 128
        si, di, truth, ncomm = block_model(nodes, nnz, 0.9, dev)
        g = torch.Generator(device=dev)
        g.manual_seed(9)
        w = torch.rand(nnz, device=dev, generator=g)
        x = torch.rand(nodes, F, device=dev, generator=g)
        out = torch.empty(nodes, F, device=dev)
        ms = device_ms(lambda: hip.gather_weight_scatter_out(si, di, w, x, out), iters) if mode != "renum" else float("nan")
        kernel = hip.last_kernel()
        uniq = int(torch.unique(si).numel())
        comp = nnz * 20 + uniq * 4 * F + nodes * 4 * F
        entry = {"workload": f"gather_weight_scatter on a shuffled block model: {nodes} nodes, {nnz} edges, {ncomm} communities of 2-20 k nodes, "
                             f"90 % of a node's edges inside its community, power-law dst, ids randomly permuted, feat={F}, float32",
                 "kernel": kernel, "kernel_ms": ms, "edges_per_s": nnz / ms * 1e3, "compulsory_bytes": comp,
                 "roofline": {"bound": "hbm", "achieved": comp / ms / 1e6, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": comp / ms / 1e6 / HBM_PEAK_GBPS, **profiled(name)}}
        try:
            if mode != "both":
                raise RuntimeError("comparator skipped in a profiling pass")
            from tools import rocsparse
            best, table, y = rocsparse.best_csr_spmm(di, si, w, x, nodes, iters=max(2, iters // 2), algs=("csr_nnz_split", "csr_merge_path"))
            entry.update(rocsparse_best_ms=best["ms"], rocsparse_best_algorithm=best["algorithm"], speedup_vs_rocsparse_best=best["ms"] / ms)
            del y
        except Exception as e:
            entry["rocsparse_error"] = repr(e)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rg = reorder.renumber(si, di, nodes) if mode != "asis" else None
        torch.cuda.synchronize()
        once_ms = (time.perf_counter() - t0) * 1e3
        if rg is None:
            entry["renumbered"] = None
        else:
            w_new = rg.edge_values(w)
            ms_static = device_ms(lambda: rg.gather_weight_scatter(w_new, x, in_new_order=True), iters)
            ms_dyn = device_ms(lambda: rg.gather_weight_scatter(w, x), iters) if mode == "both" else None
            y = rg.gather_weight_scatter(w_new, x, in_new_order=True)
            ms_ceiling = None
            if mode == "both":
                hip.gather_weight_scatter_out(si, di, w, x, out)
                ceiling = reorder.RenumberedGraph(si, di, nodes, truth)
                w_c = ceiling.edge_values(w)
                ms_ceiling = device_ms(lambda: ceiling.gather_weight_scatter(w_c, x, in_new_order=True), iters)
                del ceiling, w_c
            fair = {}
            if mode == "both":          # the comparator on the SAME renumbered matrix (kernel against kernel, permutations outside both)
                try:
                    from tools import rocsparse
                    xp, yp = rg.rows_in(x), torch.empty_like(x)
                    k_ms = device_ms(lambda: hip.gather_weight_scatter_out(rg.src_index, rg.dst_index, w_new, xp, yp), iters)
                    best, _, yr = rocsparse.best_csr_spmm(rg.dst_index, rg.src_index, w_new, xp, nodes, iters=max(2, iters // 2),
                                                          algs=("csr_nnz_split", "csr_merge_path"))
                    fair = {"kernel_ms_on_the_renumbered_list": k_ms, "rocsparse_best_ms_on_the_renumbered_matrix": best["ms"],
                            "speedup_vs_rocsparse_on_the_renumbered_matrix": best["ms"] / k_ms}
                    del xp, yp, yr
                except Exception as e:  # noqa: BLE001
                    fair = {"rocsparse_error": repr(e)}
            entry["renumbered"] = {
                **fair,
                "what": "geot_amd.reorder.renumber: label propagation on the device, then per call rows of x permuted in, the same "
                        "kernels on the renumbered + re-sorted edge list, rows of y permuted back (both permutations inside the timings)",
                "call_ms_static_weight": ms_static, "call_ms_weight_permuted_per_call": ms_dyn,
                "call_ms_with_the_true_community_order": ms_ceiling,
                "one_time_ms_label_propagation_and_renumbering": once_ms,
                "edges_within_16k_rows_before": rg.locality_before, "edges_within_16k_rows_after": rg.locality_after,
                "speedup_vs_as_shipped": ms / ms_static, "calls_to_amortise_the_one_time_cost": once_ms / max(ms - ms_static, 1e-9),
                "max_rel_diff_vs_as_shipped": float(((y - out).abs().max() / out.abs().max()).item()) if mode == "both" else None,
                "roofline_frac_on_compulsory_bytes": comp / ms_static / 1e6 / HBM_PEAK_GBPS}
            del w_new, y
        res[name] = entry
        del si, di, truth, w, x, out, rg
        torch.cuda.empty_cache()

    def train_step(name):
        """SURVEY 8(f1), the backward path: a gather_weight_scatter TRAINING STEP through autograd on configs[3]'s graph (Reddit scale,
        F=128, sources ascending inside every row as in a coalesced COO): forward over the source-blocked plan, d/dsrc over the plan
        of the transposed list, d/dweight by the SDDMM over the forward's plan (results staged, then brought into edge order) - as
        dispatched, and on the per-edge kernels (slab_mode never)."""
        import geot_amd as geot
        nodes, nnz, F = int(232_965 * scale), int(114_615_892 * scale), 128
        di = powerlaw_index(nnz, nodes, 11, dev)
        g = torch.Generator(device=dev)
        g.manual_seed(12)
        si = torch.randint(0, nodes, (nnz,), device=dev, generator=g)
        si = (torch.sort(di * nodes + si).values % nodes).contiguous()
        x = torch.rand(nodes, F, device=dev, generator=g, requires_grad=True)
        w = torch.rand(nnz, device=dev, generator=g, requires_grad=True)
        cot = torch.rand(nodes, F, device=dev, generator=g)

        def touch():
            # the weights CHANGE every step, as under an optimiser (an in-place update bumps the version counter): nothing derived from
            # their values - a transposed or plan-order copy - survives from one step to the next, for the operators or the handle
            with torch.no_grad():
                w.mul_(1.0)

        def step():
            x.grad = None
            w.grad = None
            touch()
            geot.gather_weight_scatter(si, di, w, x).backward(cot)
        out = {"workload": f"gather_weight_scatter forward + backward (d/dsrc, d/dweight) through autograd, {nodes} nodes, {nnz} edges "
                           f"(configs[3]'s graph, sources ascending inside every row), feat={F}, float32; the weight is updated in place before every step"}
        old = ops.set_option("slab_mode", "auto")
        try:
            for mode, key in (("auto", "as_dispatched"), ("never", "per_edge_kernels")):
                ops.set_option("slab_mode", mode)
                ops.clear_caches()
                for _ in range(3):           # (plans are built on the second sighting of an edge list and tried on first use)
                    step()
                out[key] = {"forward_ms": device_ms(lambda: geot.gather_weight_scatter(si, di, w.detach(), x.detach()), max(2, iters // 2)),
                            "forward_backward_ms": device_ms(step, max(2, iters // 2))}
                if mode == "auto":
                    st = ops.stats()
                    out["plans"] = {k: st[k] for k in ("plans", "plan_trials", "plans_rejected", "plans_declined") if k in st}
            out["speedup_forward_backward"] = out["per_edge_kernels"]["forward_backward_ms"] / out["as_dispatched"]["forward_backward_ms"]
            # the same step with the content guard off, and through the opt-in static-graph handle (geot_amd.Graph: owns its clones of
            # the index arrays and its plans - no fingerprint, no cache lookup)
            ops.set_option("slab_mode", "auto")
            guard_was = ops.set_option("content_guard", 0)
            for _ in range(3):
                step()
            out["without_content_guard"] = {"forward_backward_ms": device_ms(step, max(2, iters // 2))}
            ops.set_option("content_guard", guard_was)
            ops.clear_caches()
            handle = geot.Graph(si, di, num_src=nodes, num_dst=nodes)

            def hstep():
                x.grad = None
                w.grad = None
                touch()
                handle.gather_weight_scatter(w, x).backward(cot)
            for _ in range(3):
                hstep()
            out["with_graph_handle"] = {"forward_ms": device_ms(lambda: handle.gather_weight_scatter(w.detach(), x.detach()), max(2, iters // 2)),
                                        "forward_backward_ms": device_ms(hstep, max(2, iters // 2)), "handle": dict(handle.stats)}
            del handle
        finally:
            ops.set_option("slab_mode", old)
            ops.clear_caches()
        res[name] = out
        del di, si, x, w, cot
        torch.cuda.empty_cache()

    def mh_train_step(name):
        """SURVEY 8(f1) for the multi-head operator: a GAT-style step on configs[3]'s graph - mh_spmm forward,
        d/dsrc over the transposed list, d/dweight by the multi-head SDDMM - through autograd: as dispatched (plans found by the host
        layer), on the per-edge kernels, and through the static-graph handle; plus the attention form that keeps the coefficients in the
        plan's edge order from the score SDDMM to the SpMM."""
        import geot_amd as geot
        nodes, nnz, H, F = int(232_965 * scale), int(114_615_892 * scale), 4, 64
        di = powerlaw_index(nnz, nodes, 11, dev)
        g = torch.Generator(device=dev)
        g.manual_seed(12)
        si = torch.randint(0, nodes, (nnz,), device=dev, generator=g)
        x = torch.rand(nodes, H, F, device=dev, generator=g, requires_grad=True)
        w = torch.rand(nnz, H, device=dev, generator=g, requires_grad=True)
        cot = torch.rand(nodes, H, F, device=dev, generator=g)

        def touch():                                   # (the weights change every step, as under an optimiser)
            with torch.no_grad():
                w.mul_(1.0)

        def step():
            x.grad = None
            w.grad = None
            touch()
            geot.mh_spmm(si, di, w, x).backward(cot)
        out = {"workload": f"mh_spmm forward + backward (d/dsrc, d/dweight) through autograd, {nodes} nodes, {nnz} edges (configs[3]'s graph, "
                           f"uniform-random sources), heads={H} feat={F}, float32; the weight is updated in place before every step"}
        old = ops.set_option("slab_mode", "auto")
        try:
            for mode, key in (("auto", "as_dispatched"), ("never", "per_edge_kernels")):
                ops.set_option("slab_mode", mode)
                ops.clear_caches()
                for _ in range(3):
                    step()
                out[key] = {"forward_ms": device_ms(lambda: geot.mh_spmm(si, di, w.detach(), x.detach()), max(2, iters // 2)),
                            "forward_backward_ms": device_ms(step, max(2, iters // 2))}
            out["speedup_forward_backward"] = out["per_edge_kernels"]["forward_backward_ms"] / out["as_dispatched"]["forward_backward_ms"]
            ops.set_option("slab_mode", "auto")
            ops.clear_caches()
            handle = geot.Graph(si, di, num_src=nodes, num_dst=nodes)

            def hstep():
                x.grad = None
                w.grad = None
                touch()
                handle.mh_spmm(w, x).backward(cot)
            for _ in range(3):
                hstep()
            out["with_graph_handle"] = {"forward_ms": device_ms(lambda: handle.mh_spmm(w.detach(), x.detach()), max(2, iters // 2)),
                                        "forward_backward_ms": device_ms(hstep, max(2, iters // 2))}
            # the same step in bf16 storage (fp32 accumulation in every kernel; d/dweight on the matrix cores)
            xb = x.detach().bfloat16().requires_grad_()
            wb = w.detach().bfloat16().requires_grad_()
            cotb = cot.bfloat16()

            def hstep_bf16():
                xb.grad = None
                wb.grad = None
                with torch.no_grad():
                    wb.mul_(1.0)
                handle.mh_spmm(wb, xb).backward(cotb)
            for _ in range(3):
                hstep_bf16()
            out["with_graph_handle"]["bf16_forward_backward_ms"] = device_ms(hstep_bf16, max(2, iters // 2))
            del xb, wb, cotb
            # attention: scores by the SDDMM, exp, the SpMM - per-edge tensors in edge order (operators) and in plan order (handle)
            q = torch.rand(nodes, H, F, device=dev, generator=g) / 8
            k = torch.rand(nodes, H, F, device=dev, generator=g) / 8

            def att_ops():
                s = torch.ops.geot.mh_sddmm(si, di, q, k, False)
                return geot.mh_spmm(si, di, torch.exp(s), x.detach())

            def att_handle():
                s = handle.mh_sddmm(q, k, plan_order=True)
                return handle.mh_spmm(s.with_values(torch.exp(s.values)), x.detach())
            for _ in range(3):
                att_ops()
                att_handle()
            out["attention_forward_ms"] = {"operators_edge_order": device_ms(att_ops, max(2, iters // 2)),
                                           "graph_handle_plan_order": device_ms(att_handle, max(2, iters // 2))}
            # the same in bf16 storage through the handle: the scores come from the matrix cores (seg_slab_sddmm_mfma_kernel: 16 edges x the
            # group's rows x 32 features per v_mfma_f32_16x16x32_bf16), fp32 accumulation as everywhere
            qb, kb, xb = q.bfloat16(), k.bfloat16(), x.detach().bfloat16()

            def att_handle_bf16():
                s = handle.mh_sddmm(qb, kb, plan_order=True)
                return handle.mh_spmm(s.with_values(torch.exp(s.values.float()).bfloat16()), xb)
            for _ in range(3):
                att_handle_bf16()
            t_sddmm = device_ms(lambda: handle.mh_sddmm(qb, kb, plan_order=True), max(2, iters // 2))
            sddmm_kernel = hip.last_kernel()
            out["attention_forward_ms"]["graph_handle_plan_order_bf16"] = device_ms(att_handle_bf16, max(2, iters // 2))
            out["attention_forward_ms"]["bf16_scores_alone_ms"] = t_sddmm
            out["attention_forward_ms"]["bf16_scores_kernel"] = sddmm_kernel
            out["handle"] = dict(handle.stats)
            del handle, q, k, qb, kb, xb
        finally:
            ops.set_option("slab_mode", old)
            ops.clear_caches()
        res[name] = out
        del di, si, x, w, cot
        torch.cuda.empty_cache()

    # one entry failing (out of memory on a smaller device, a comparator that does not load) must not take the others with it
    plan = [("cfg1", lambda n: cfg1(n)),
            ("gws_cfg3", lambda n: gws(n, "uniform", torch.float32)),
            ("gws_cfg3_local", lambda n: gws(n, "local", torch.float32)),
            ("gws_cfg3_powerlaw_src", lambda n: gws(n, "powerlaw", torch.float32)),
            ("gws_cfg3_blockmodel", lambda n: blockmodel(n)),
            ("gws_cfg3_blockmodel_asis", lambda n: blockmodel(n, "asis")),        # (profiling passes only: not in SECONDARY)
            ("gws_cfg3_blockmodel_renum", lambda n: blockmodel(n, "renum")),
            ("mh_spmm_cfg4", lambda n: mh(n, torch.float32)),
            ("mh_spmm_cfg4_powerlaw_src", lambda n: mh(n, torch.float32, "powerlaw")),
            ("mh_spmm_cfg4_coalesced", lambda n: mh(n, torch.float32, "uniform", coalesced=True)),
            ("gws_cfg3_bf16", lambda n: gws(n, "uniform", torch.bfloat16)),
            ("mh_spmm_cfg4_bf16", lambda n: mh(n, torch.bfloat16)),
            ("gws_train_step_cfg4_graph", lambda n: train_step(n)),
            ("mh_train_step_cfg4_graph", lambda n: mh_train_step(n))]
    for name, run in plan:
        if name not in want:
            continue
        try:
            run(name)
        except Exception as e:  # noqa: BLE001
            res[name] = {"error": repr(e)}
            ops.clear_caches()
            torch.cuda.empty_cache()
    return res


def cfg5_leg(world, rank, dev, scale, steps, warmup, cuts="equal", collective="all_gather"):
    """BASELINE.json configs[4] as a bounded extra leg of the default line (every N: the SCALE run then carries the gather_scatter
    numbers at 1 / 2 / 4 / 8 GPUs without a flag): weak scaling - every GPU holds 1/8 of the configuration's edges and dst rows
    (8 ranks = the full 1.6 B edges), src replicated (56.9 GB).  All ranks call this; the result is meaningful on rank 0."""
    import torch
    import geot_amd as geot
    from geot_amd import hip, sharding
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
    nodes_all, feat = int(CFG5_NODES * scale), CFG5_FEAT
    rows_global = max(world, nodes_all * world // 8)
    nnz_target = int(CFG5_EDGES * scale) * world // 8
    setup_error = None
    try:
        index, src_index, first_key, rows, nnz_global, _ = global_list_shard(rows_global, nnz_target, nodes_all, world, rank, cuts, seed=13, device=dev)
        gen = torch.Generator(device=dev)
        gen.manual_seed(15)
        src = torch.rand(nodes_all, feat, device=dev, generator=gen)
    except BaseException as e:  # noqa: BLE001  (SystemExit of an empty shard included)
        setup_error = repr(e)
    if distributed:             # every rank steps or none does: a rank that failed to set up must not leave the others in a collective
        ok = torch.tensor([0 if setup_error else 1], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            raise RuntimeError(f"cfg5 leg skipped on every rank: set-up failed on some rank (this rank: {setup_error})")
    elif setup_error:
        raise RuntimeError(setup_error)
    timing = {}
    if distributed:
        def step():
            return sharding.sharded_gather_scatter(src_index, index, src, key_offset=first_key, timing=timing, collective=collective)[0]
    else:
        def step():
            return geot.gather_scatter(src_index, index, src)

    def sync_all():
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
            torch.cuda.synchronize(dev)
    for _ in range(warmup):
        step()
    sync_all()
    timing.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    ev = timing.get("exchange_events")
    exchange_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev) if ev else None
    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel = hip.last_kernel()
    ms = device_ms(lambda: hip.gather_scatter_out(src_index, index, src, torch.empty(rows, feat, device=dev)), 3, warmup=1)
    box = hip.profile_box(src[: 20_000_000])                  # this box's own streamed-read ceiling, now
    # ... and its RANDOM-ROW rate on this very table (uniform-random 512-byte rows, 16 reads in flight per lane, nothing else): the
    # yardstick a per-edge gather kernel on a 57 GB table is to be read against (the streamed ceiling is not: round 5's two boxes
    # were 1 % apart on streams and 13 % apart on this kernel)
    # ... and that rate with the operator's WRITE MIX: one output row written per `run` rows read (run = this shard's edges per dst row),
    # nothing else - on this part a few per cent of writes among random reads cost more than their bytes (tools/kexp4.hip: 6.4 TB/s
    # pure, 5.4 with one row written per 16 read, whatever the store policy), and that is what a gather_scatter kernel can reach
    run = max(1, round(index.numel() / max(rows, 1)))
    try:
        rows_box = hip.profile_box_rows(src, out=torch.empty(rows, feat, device=dev), run=run)
    except Exception as e:  # noqa: BLE001
        rows_box = {"random_row_gbps": None, "random_row_gbps_nt": None, "best_gbps": None, "mix_row_gbps": None, "mix_run": None, "error": repr(e)}
    uniq = int(torch.unique(src_index).numel())
    comp = index.numel() * 16 + uniq * 4 * feat + rows * 4 * feat
    # what the per-edge gather MOVES: one row read per edge (a 57 GB table is re-read from HBM every time: 256 MiB of Infinity Cache
    # hold 0.4 % of it; PMC: fabric traffic = 1.97 x compulsory, L2 hit 6 %, profiles/r05/cfg5_study/) + the indices + the output
    moved = index.numel() * (16 + 4 * feat) + rows * 4 * feat
    out = {"workload": f"gather_scatter, papers100M-scale synthetic (BASELINE.json configs[4]), feat={feat}: ONE global dst-sorted list of "
                       f"{nnz_global} edges -> {rows_global} dst rows cut into {world} edge ranges ({cuts} cuts); this rank {index.numel()} edges -> "
                       f"{rows} rows; src {nodes_all} x {feat} fp32 replicated ({nodes_all * feat * 4 / 1e9:.1f} GB per GPU); 8 ranks = the full 1.6 B edges",
           "scaling": "weak", "n_gpus": world, "steps": steps, "value": nnz_global * steps / elapsed, "unit": "edges/s",
           "ms_per_step": elapsed / steps * 1e3, "boundary_exchange_ms": exchange_ms, "collective": collective if distributed else None,
           "kernel": kernel, "kernel_ms_rank0": ms, "compulsory_bytes_rank0": comp, "distinct_source_rows_rank0": uniq,
           "roofline": {"bound": "hbm", "achieved": comp / ms / 1e6, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": comp / ms / 1e6 / HBM_PEAK_GBPS,
                        **profiled("gather_scatter_cfg5"),
                        "moved_bytes_rank0": moved, "moved_gbps": moved / ms / 1e6,
                        "row_gather_gbps": index.numel() * 4 * feat / ms / 1e6,
                        "box_read_ceiling_gbps": box["read_ceiling_gbps"], "box_sclk_mhz": box["sclk_mhz"],
                        "moved_frac_of_box_read_ceiling": moved / ms / 1e6 / box["read_ceiling_gbps"],
                        "box_random_row_gbps": rows_box["best_gbps"], "box_random_row_gbps_default_policy": rows_box["random_row_gbps"],
                        "box_random_row_gbps_nt": rows_box["random_row_gbps_nt"],
                        "row_gather_frac_of_box_random_row": (index.numel() * 4 * feat / ms / 1e6 / rows_box["best_gbps"]) if rows_box["best_gbps"] else None,
                        "box_random_row_gbps_with_write_mix": rows_box["mix_row_gbps"], "write_mix_run": rows_box["mix_run"],
                        "row_gather_frac_of_box_row_mix": (index.numel() * 4 * feat / ms / 1e6 / rows_box["mix_row_gbps"]) if rows_box["mix_row_gbps"] else None,
                        "note": "every edge's 512-byte row comes from HBM (table 57 GB >> 256 MiB Infinity Cache): the kernel runs at the part's random-"
                                "row rate - 4.7 TB/s of row reads from any table >= 4 GB whatever the tile shape, loads in flight, in-tile source "
                                "order or page locality (profiles/r05/cfg5_study/exp_gather_table*.txt); compulsory bytes credit a row once, the "
                                "graph reads it 2.2 times"}}
    del index, src_index, src
    torch.cuda.empty_cache()
    return out


COMPACT_LIMIT = 3000             # bytes: the driver keeps a tail of stdout (round 5: 8 081 bytes of a 23.8 KB line -> parsed: null)
HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data")
DIST_KEYS = ("ranks_seen", "boundary_exchange_ms", "collective", "boundary_exchange_ms_by_collective", "key_exchange_ms", "cuts",
             "dist_backend", "src_sharding")


def _num(x, digits=6):
    """A finite number rounded to `digits` significant figures, or None: the compact line carries numbers, never prose."""
    if isinstance(x, bool) or not isinstance(x, (int, float)):
        return None
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return x if isinstance(x, int) else float(f"{x:.{digits}g}")


def compact_record(res):
    """The graded stdout line: exactly the contract keys, `roofline` and `cpu_baseline` reduced to their contract fields (+ the
    kernel's name and time), and a flat `extras` of at most ten numbers.  No notes, no nested workloads: everything else is in the
    file `detail` names."""
    out = {k: res[k] for k in HEADLINE_KEYS if k in res}
    for k in ("value", "ms_per_step"):
        out[k] = _num(out.get(k), 9)
    cfg = res.get("config", {})
    out["config"] = {"workload": cfg.get("workload_short", cfg.get("workload", ""))[:200],
                     **{k: cfg[k] for k in ("nnz_per_gpu", "rows_per_gpu", "feat", "index_dtype") if k in cfg}}
    rf = res.get("roofline", {})
    out["roofline"] = {"bound": rf.get("bound"), "achieved": _num(rf.get("achieved")), "peak": rf.get("peak"), "unit": rf.get("unit"),
                       "frac": _num(rf.get("frac")), "traffic": rf.get("traffic"), "kernel": (rf.get("kernel") or "")[:96],
                       "kernel_ms": _num(rf.get("kernel_ms")), "box_read_ceiling_gbps": _num(rf.get("box_read_ceiling_gbps"))}
    cb = res.get("cpu_baseline")
    if cb is not None:
        out["cpu_baseline"] = {"value": _num(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                               "sample": (cb.get("sample") or "")[:120]}
        if cb.get("host_cores") is not None:
            out["cpu_baseline"]["host_cores"] = cb["host_cores"]
    for k in DIST_KEYS:
        if k in res:
            v = res[k]
            out[k] = {a: _num(b) for a, b in v.items()} if isinstance(v, dict) else (_num(v) if isinstance(v, float) else v)
    sec = res.get("secondary") or {}
    ex = {}

    def put(name, value):
        value = _num(value)
        if value is not None and len(ex) < 10:
            ex[name] = value

    def get(d, *path):
        for k in path:
            d = d.get(k) if isinstance(d, dict) else None
        return d
    put("gws_cfg3_ms", get(sec, "gws_cfg3", "kernel_ms"))
    put("rocsparse_best_ms", get(sec, "gws_cfg3", "rocsparse_best_ms"))
    put("gws_speedup_vs_rocsparse", get(sec, "gws_cfg3", "speedup_vs_rocsparse_best"))
    put("mh_spmm_cfg4_ms", get(sec, "mh_spmm_cfg4", "kernel_ms"))
    put("mh_spmm_cfg4_bf16_ms", get(sec, "mh_spmm_cfg4_bf16", "kernel_ms"))
    put("gather_scatter_cfg5_ms", get(sec, "gather_scatter_cfg5", "ms_per_step"))
    put("gather_scatter_cfg5_edges_per_s", get(sec, "gather_scatter_cfg5", "value"))
    put("cfg5_kernel_frac_of_box_random_row", get(sec, "gather_scatter_cfg5", "roofline", "row_gather_frac_of_box_random_row"))
    put("cfg5_kernel_frac_of_box_row_write_mix", get(sec, "gather_scatter_cfg5", "roofline", "row_gather_frac_of_box_row_mix"))
    put("cfg1_us_per_call", get(sec, "cfg1", "us_per_call_as_dispatched"))
    put("cfg5_boundary_exchange_ms", get(sec, "gather_scatter_cfg5", "boundary_exchange_ms"))     # (N > 1, where the entries above are absent)
    out["extras"] = ex
    errs = sorted(k for k, v in sec.items() if isinstance(v, dict) and "error" in v)
    if errs or "error" in sec:
        out["secondary_errors"] = errs[:8] or ["secondary"]
    return out


def emit(res, args):
    """Rank 0: the full record to --detail-out, ONE compact JSON line (<= COMPACT_LIMIT bytes) as the last line of stdout."""
    detail = None
    for path in (args.detail_out, os.path.join("/tmp", os.path.basename(args.detail_out))):
        try:
            with open(path, "w") as f:
                json.dump(res, f, indent=1)
            detail = path
            break
        except OSError:
            continue
    if args.full:
        print(json.dumps(res), flush=True)
        return
    line = compact_record(res)
    line["detail"] = os.path.relpath(detail, ROOT) if detail and detail.startswith(ROOT) else detail
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > COMPACT_LIMIT:              # never again an unparseable headline: shed the optional parts, then fail loudly
        for k in ("extras", "boundary_exchange_ms_by_collective", "secondary_errors"):
            line.pop(k, None)
        text = json.dumps(line, separators=(",", ":"))
        if len(text) > COMPACT_LIMIT:
            raise SystemExit(f"bench.py: compact record is {len(text)} bytes (> {COMPACT_LIMIT})")
    print(text, flush=True)


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)                                     # never returns
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if args.gpus != world and distributed:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path in this package)")
    # GEOT_DIST_BACKEND=gloo lets two test ranks share one GPU (RCCL refuses duplicate devices)
    backend = os.environ.get("GEOT_DIST_BACKEND", "nccl")       # "nccl" is RCCL on ROCm
    local_dev = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import geot_amd as geot
    from geot_amd import hip, sharding

    timing = {}
    if args.workload == "cfg2":
        nnz, rows, feat = NNZ, KEYS, FEAT
        if args.strong and distributed:
            nnz, rows = NNZ // world, KEYS // world
        index = powerlaw_index(nnz, rows, seed=rank, device=dev)          # rank-local keys 0..rows-1
        gen = torch.Generator(device=dev)
        gen.manual_seed(1000 + rank)
        src = torch.rand(nnz, feat, device=dev, generator=gen)
        alg = algorithmic_bytes(nnz, feat, rows)
        key_offset = rank * (rows - 1) if args.cuts == "equal" else rank * rows   # equal: neighbours share one key
        if distributed:
            def step(collective=args.collective):
                return sharding.sharded_index_scatter(index, src, key_offset=key_offset, timing=timing, collective=collective)[0]
        else:
            def step():
                return geot.index_scatter(0, src, index, "sum", True)
        coll = "RCCL" if backend == "nccl" else backend
        workload = ("index_scatter sorted sum, power-law 10M edges -> 1M nodes, feat=64 (BASELINE.json configs[1])"
                    + (f" per GPU, neighbouring shards share their boundary key, boundary rows exchanged by a {coll} {args.collective}" if distributed else ""))
        workload_short = "index_scatter sorted sum, power-law 10M edges -> 1M nodes, feat=64 (BASELINE.json configs[1])" + (" per GPU" if distributed else "")
        step_desc = ("sharding.sharded_index_scatter: 16-byte-per-rank key all_gather under the local kernels (tile + fix-up), "
                     f"{args.collective} of the first-row partials, owner add" if distributed else
                     "geot.index_scatter(0, src, index, 'sum', True): read-back of index[-1] + alloc + tile kernel + fix-up kernel")
        metric = METRIC
    else:
        # BASELINE.json configs[4]: ONE global dst-sorted edge list cut into `world` contiguous edge ranges (global_list_shard).
        #   weak (default): every GPU holds 1/8 of the configuration's edges and dst rows - the list covers world/8 of the dst
        #                   rows, 8 ranks = the full 1.6 B edges; src (all nodes) replicated;
        #   --strong:       the full configuration (x --scale) at every N, cut into N.
        nodes_all, feat = int(CFG5_NODES * args.scale), CFG5_FEAT
        share = 8 if args.strong else world
        rows_global = max(world, nodes_all * share // 8)
        nnz_target = int(CFG5_EDGES * args.scale) * share // 8
        index, src_index, first_key, rows, nnz_global, cut_edges = global_list_shard(
            rows_global, nnz_target, nodes_all, world, rank, args.cuts, seed=13, device=dev)
        nnz = index.numel()
        node_sharded = distributed and args.src_sharding == "node"
        gen = torch.Generator(device=dev)
        gen.manual_seed(15)                                                # src is REPLICATED: same seed on every rank
        if node_sharded:                                                   # ... or this rank's contiguous range of nodes only (its own seed)
            node_offsets = [nodes_all * r // world for r in range(world + 1)]
            gen.manual_seed(15 + rank)
            src = torch.rand(node_offsets[rank + 1] - node_offsets[rank], feat, device=dev, generator=gen)
            halo = sharding.HaloPlan.build(src_index, node_offsets)       # (a collective; once per edge list)
        else:
            src = torch.rand(nodes_all, feat, device=dev, generator=gen)
        uniq = int(torch.unique(src_index).numel())
        alg = nnz * 16 + uniq * 4 * feat + rows * 4 * feat                 # compulsory bytes
        key_offset = first_key
        if node_sharded:
            def step(collective=args.collective):
                return sharding.sharded_gather_scatter_node(src_index, index, src, node_offsets, key_offset=key_offset, timing=timing,
                                                            collective=collective, halo=halo)[0]
        elif distributed:
            def step(collective=args.collective):
                return sharding.sharded_gather_scatter(src_index, index, src, key_offset=key_offset, timing=timing, collective=collective)[0]
        else:
            def step():
                return geot.gather_scatter(src_index, index, src)
        workload = (f"gather_scatter, papers100M-scale synthetic, feat={feat}: one global dst-sorted list of {nnz_global} edges -> "
                    f"{rows_global} dst rows cut into {world} edge ranges ({args.cuts} cuts), rank 0 holds {nnz} edges / {rows} rows; "
                    f"src {nodes_all} x {feat} fp32 " + (f"sharded by node ({src.shape[0]} rows on this rank; halo plan: {halo.mode}, {halo.rows_fetched} rows = "
                                                       f"{halo.bytes_fetched(4 * feat) / 1e9:.2f} GB fetched per step)" if node_sharded else "replicated") +
                    " (BASELINE.json configs[4]; 8 ranks weak = the full 1.6 B edges)")
        workload_short = (f"gather_scatter papers100M-scale synthetic feat={feat}, {nnz_global} edges cut into {world} edge ranges, src "
                          + ("sharded by node" if node_sharded else "replicated") + " (BASELINE.json configs[4])")
        step_desc = "geot.gather_scatter(src_index, dst_index, src)" + (" via sharding.sharded_gather_scatter" if distributed else "")
        metric = "aggregated edges/sec, gather_scatter feat=128, edge-sharded, src " + ("sharded by node" if node_sharded else "replicated")

    def sync_all():
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
            torch.cuda.synchronize(dev)

    out = None
    for _ in range(args.warmup):
        out = step()
    sync_all()
    timing.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    def mean_ms(events):
        return sum(a.elapsed_time(b) for a, b in events) / len(events) if events else None

    exchange_ms = mean_ms(timing.get("exchange_events"))
    fetch_ms = mean_ms(timing.get("fetch_events"))     # (--src-sharding node: the per-step fetch of the source rows)
    key_ms = mean_ms(timing.get("key_events"))
    if key_ms is None and timing.get("key_wall_ms"):
        key_ms = sum(timing["key_wall_ms"]) / len(timing["key_wall_ms"])
    by_collective = {args.collective: exchange_ms}
    if distributed:                                   # the other form of the boundary exchange, a few extra steps (untimed in `value`)
        other = "reduce_scatter" if args.collective == "all_gather" else "all_gather"
        timing.clear()
        for _ in range(max(2, min(args.steps, 20))):
            out = step(other)
        sync_all()
        by_collective[other] = mean_ms(timing.get("exchange_events"))
        timing.clear()
    timing = None
    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    total_edges = world * nnz
    if args.workload == "cfg5":
        total_edges = nnz_global                      # (the shards of one list differ by an edge or by a row's tail)
    edges_per_s = total_edges * args.steps / elapsed

    # ---- roofline of the dominant kernel: HIP events around the tile kernel, K more steps -------
    hip.profile_enable(True)
    hip.profile_reset()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize(dev)
    prof = hip.profile_read()
    hip.profile_enable(False)
    kernel = hip.last_kernel()                        # what the launcher picked for this call (geot_last_kernel), not a literal
    box = hip.profile_box(src if args.workload == "cfg2" else src[: 20_000_000])   # this box's own read ceiling, now
    main_ms = prof["main_ms"] / max(prof["calls"], 1)
    fix_ms = prof["fixup_ms"] / max(prof["calls"], 1)
    achieved = alg / (main_ms * 1e-3) / 1e9
    traffic = traffic_source = None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if args.workload == "cfg2" and os.path.exists(tf):
        try:
            traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
            traffic_source = "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel on this workload, recorded session; not measured in this run)"
        except Exception:
            traffic = None

    if rank == 0:
        res = {
            "metric": metric, "value": edges_per_s, "unit": "edges/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if (args.strong and distributed) else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "workload_short": workload_short, "nnz_per_gpu": nnz, "rows_per_gpu": rows, "feat": feat,
                       "index_dtype": "int64", "step": step_desc},
            "hbm_gbps_whole_call": world * alg * args.steps / elapsed / 1e9,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel, "kernel_ms": main_ms,
                         "fixup_kernel_ms": fix_ms, "algorithmic_bytes_per_launch": alg,
                         "frac_tile_plus_fixup": alg / ((main_ms + fix_ms) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "frac_of_measured_copy_ceiling": achieved / HBM_COPY_GBPS,
                         "box_read_ceiling_gbps": box["read_ceiling_gbps"], "box_sclk_mhz": box["sclk_mhz"],
                         "frac_of_box_read_ceiling": achieved / box["read_ceiling_gbps"]},
        }
        if distributed:
            res["boundary_exchange_ms"] = exchange_ms      # rank 0, the timed steps' collective + owner adds (None: no key shared)
            res["collective"] = args.collective
            res["boundary_exchange_ms_by_collective"] = by_collective   # all_gather + owner add | reduce_scatter (north star's wording)
            res["key_exchange_ms"] = key_ms                # 16-byte-per-rank all_gather of (first, last) keys + copy to the host, side stream
            res["ranks_seen"] = dist.get_world_size()      # what the process group reports (after init_process_group)
            res["cuts"] = args.cuts
            res["dist_backend"] = backend
            if args.workload == "cfg5":
                res["src_sharding"] = args.src_sharding
                if args.src_sharding == "node":
                    res["source_rows_fetch"] = {"mode": halo.mode, "rows_fetched_per_step_rank0": halo.rows_fetched, "table_rows_rank0": halo.table_rows,
                                                "bytes_fetched_per_step_rank0": halo.bytes_fetched(4 * feat),
                                                "fetch_ms_rank0": fetch_ms,
                                                "src_rows_on_this_rank": int(src.shape[0]), "src_rows_total": nodes_all}
        if not distributed and args.workload == "cfg2" and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(index, src)
            except Exception as e:  # the baseline must never take the GPU number down with it
                res["cpu_baseline"] = {"value": None, "unit": "edges/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed", "error": repr(e)}
        if not distributed and args.workload == "cfg2" and not args.no_secondary:
            del index, src, out
            torch.cuda.empty_cache()
            try:
                only = [x for x in args.only_secondary.split(",") if x] or (list(SECONDARY) if args.secondary == "all" else None)
                res["secondary"] = secondary(dev, scale=args.scale, only=only)
            except Exception as e:
                res["secondary"] = {"error": repr(e)}
    # configs[4] rides along at every N (bounded: a few steps), so the driver's 1 / 2 / 4 / 8 runs report it without a flag
    want_cfg5 = args.workload == "cfg2" and not args.no_secondary and (not args.only_secondary or "cfg5" in args.only_secondary.split(","))
    if want_cfg5:
        index = src = out = None
        torch.cuda.empty_cache()
        # (test ranks that SHARE one GPU over a gloo rendezvous cannot each hold the 56.9 GB source table: a hundredth of it there)
        leg_scale = args.scale if (not distributed or backend == "nccl") else min(args.scale, 0.01)
        try:
            leg = cfg5_leg(world, rank, dev, leg_scale, steps=max(2, min(args.steps, 8)), warmup=2, cuts=args.cuts, collective=args.collective)
        except Exception as e:
            leg = {"error": repr(e)}
        if rank == 0:
            res.setdefault("secondary", {})["gather_scatter_cfg5"] = leg
    if rank == 0:
        emit(res, args)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
