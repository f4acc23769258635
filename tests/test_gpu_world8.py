"""BASELINE.json configs[4] rehearsed at world 8 on ONE GPU (gloo rendezvous, GEOT_DIST_BACKEND=gloo), with the HIP operator as
every rank's local reduction - so the first 8-GPU run of `bench.py --gpus 8` needs no code change.

  * correctness: one global dst-sorted edge list with everything the boundary logic has to survive - a leading gap (first key
    > 0), a hub that spans three whole ranks and parts of two more, a rank that lies wholly inside that run, empty keys between
    two ranks' ranges - cut with `equal_edge_cuts` into 8; `sharded_gather_scatter` (plain and weighted, both collectives) per
    rank with the DEFAULT local_op (the HIP kernels); the ranks' rows, concatenated, are the unsharded `geot.gather_scatter` /
    `gather_weight_scatter` of the same list - within 1e-5, and BIT-equal on exactly summable data;
  * bench.py --gpus 8 --workload cfg5 for both collectives, both kinds of cut, weak and --strong.

The reference has no multi-GPU path (new work); the semantics checked are those of its single-GPU
gather_scatter (csrc/cuda/gather_scatter_kernel.cuh:118-186: dst[dst_index[e]] += src[src_index[e]])."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
WORLD = 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _global_list(exact):
    """~400 k edges over 3 000 dst rows, F = 128.  Keys: nothing below 5 (leading gap); key 700 holds 45 % of the edges (with
    equal cuts into 8: it starts inside rank 1, covers ranks 2-4 whole and ends inside rank 5); keys 1500..1899 are empty."""
    rng = np.random.default_rng(77)
    nnz, K, nodes, F = 400_000, 3_000, 2_500, 128
    n_low, n_hub, n_mid = int(nnz * 0.20), int(nnz * 0.45), int(nnz * 0.10)   # (the mid block ends exactly on the cut between ranks 5 and 6)
    low = rng.integers(5, 700, n_low)
    high = np.concatenate([rng.integers(701, 1500, n_mid), rng.integers(1900, K, nnz - n_low - n_hub - n_mid)])
    di = np.sort(np.concatenate([low, np.full(n_hub, 700), high])).astype(np.int64)
    di[-1] = K - 1
    si = rng.integers(0, nodes, nnz).astype(np.int64)
    if exact:   # small integers x powers of two: every partial sum is exact in fp32, any order of addition gives the same bits
        x = rng.integers(0, 8, (nodes, F)).astype(np.float32)
        w = (2.0 ** rng.integers(-2, 2, nnz)).astype(np.float32)
    else:
        x = rng.random((nodes, F), dtype=np.float32)
        w = rng.random(nnz, dtype=np.float32)
    return si, di, w, x


def _rank(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import geot_amd
        from geot_amd import sharding
        res = {}
        for exact in (True, False):
            si, di, w, x = (torch.from_numpy(a).cuda() for a in _global_list(exact))
            cuts = sharding.equal_edge_cuts(di.numel(), world)
            e0, e1 = cuts[rank], cuts[rank + 1]
            si_s, di_s, w_s = si[e0:e1].clone(), di[e0:e1].clone(), w[e0:e1].clone()
            full = {False: geot_amd.gather_scatter(si, di, x), True: geot_amd.gather_weight_scatter(si, di, w, x)}
            for weighted in (False, True):
                for coll in ("all_gather", "reduce_scatter"):
                    for it in range(2):                                # the second call speculates on the remembered keys
                        out, first = sharding.sharded_gather_scatter(si_s, di_s, x, weight_shard=w_s if weighted else None, collective=coll)
                    want = full[weighted][first:first + out.shape[0]]
                    if exact or out.shape[0] == 0:
                        ok = bool(torch.equal(out, want))
                    else:
                        scale = float(full[weighted].abs().max())
                        ok = bool(((out - want).abs().max() <= 1e-5 * scale).item())
                    res[(exact, weighted, coll)] = (first, out.shape[0], ok)
        # key_offset form (what bench.py uses): rank-local keys, global key = local + offset
        si, di, w, x = (torch.from_numpy(a).cuda() for a in _global_list(True))
        cuts = sharding.equal_edge_cuts(di.numel(), world)
        e0, e1 = cuts[rank], cuts[rank + 1]
        off = int(di[e0])
        out, first = sharding.sharded_gather_scatter(si[e0:e1].clone(), (di[e0:e1] - off).contiguous(), x, key_offset=off)
        want = geot_amd.gather_scatter(si, di, x)[first:first + out.shape[0]]
        res["key_offset"] = (first, out.shape[0], bool(torch.equal(out, want)))
        # reductions other than sum across the ranks (PyG's aggr; csrc/cpu/index_scatter_cpu.cpp:124-134 is the semantics): mean ships
        # (partial sum, count) of the shared rows, max / min the partial row; the hub row is combined from five ranks' partials
        si, di, w, x = (torch.from_numpy(a).cuda() for a in _global_list(False))
        x = x - 0.5
        for red in ("mean", "max", "min"):
            for weighted in (False, True):
                for coll in ("all_gather", "reduce_scatter"):
                    out, first = sharding.sharded_gather_scatter(si[e0:e1].clone(), di[e0:e1].clone(), x, weight_shard=w[e0:e1].clone() if weighted else None,
                                                                 collective=coll, reduce=red)
                    whole = geot_amd.gather_weight_scatter(si, di, w, x, red) if weighted else geot_amd.gather_scatter(si, di, x, red)
                    want = whole[first:first + out.shape[0]]
                    if out.shape[0] == 0 or red != "mean":
                        ok = bool(torch.equal(out, want))            # selections are exact whatever the grouping
                    else:
                        ok = bool(((out - want).abs().max() <= 1e-5 * float(whole.abs().max())).item())
                    res[(red, weighted, coll)] = (first, out.shape[0], ok)
        # ... and index_scatter with a per-edge operand sharded with the edges
        g = torch.Generator(device="cuda")
        g.manual_seed(5)
        src_e = torch.rand(di.numel(), 64, device="cuda", generator=g) - 0.5
        for red in ("sum", "mean", "max", "prod"):
            out, first = sharding.sharded_index_scatter(di[e0:e1].clone(), src_e[e0:e1].clone(), reduce=red, collective="reduce_scatter")
            whole = geot_amd.index_scatter(0, src_e, di, red, True)
            want = whole[first:first + out.shape[0]]
            if red == "prod":        # the hub's product of 180 k values underflows to +-0 / denormals either way: compare where it is normal
                ok = bool(torch.allclose(out, want, rtol=1e-3, atol=1e-30))
            elif red == "max" or out.shape[0] == 0:
                ok = bool(torch.equal(out, want))
            else:
                ok = bool(((out - want).abs().max() <= 1e-5 * float(whole.abs().max())).item())
            res[("index_scatter", red)] = (first, out.shape[0], ok)
        q.put((rank, res, int(di[e0]), int(di[e1 - 1]), full[False].shape[0]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_world8_sharded_gather_ops_on_the_hip_kernels_equal_the_unsharded_call():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=600) for _ in range(WORLD)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    firsts, lasts = [g[2] for g in got], [g[3] for g in got]
    K = got[0][4]
    # the list is what the docstring says it is: a leading gap, the hub over >= 3 whole ranks, a rank wholly inside it
    assert firsts[0] >= 5
    inside = [r for r in range(WORLD) if firsts[r] == 700 and lasts[r] == 700]
    assert len(inside) >= 3, (firsts, lasts)
    assert any(firsts[r] > lasts[r - 1] + 1 for r in range(1, WORLD)), "no empty keys between two ranks' ranges"
    for key in got[0][1]:
        row = 0
        for rank, res, *_ in got:
            first, n, ok = res[key]
            assert ok, (key, rank)
            assert first == row, (key, rank, first, row)
            row += n
        assert row == K, (key, row, K)
    # a rank wholly inside the hub owns no row at all
    assert all(got[r][1][(True, False, "all_gather")][1] == 0 for r in inside)


def _bench(extra, timeout=900):
    env = dict(os.environ, GEOT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(WORLD),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(WORLD), "--workload", "cfg5", "--scale", "0.01", "--steps", "3", "--warmup", "1"] + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("collective", ["all_gather", "reduce_scatter"])
def test_bench_cfg5_world8_equal_cuts(collective):
    r = _bench(["--cuts", "equal", "--collective", collective])
    assert r["n_gpus"] == WORLD and r["ranks_seen"] == WORLD and r["scaling"] == "weak" and r["cuts"] == "equal"
    assert r["collective"] == collective and r["unit"] == "edges/s" and r["value"] > 1e8
    by = r["boundary_exchange_ms_by_collective"]
    assert set(by) == {"all_gather", "reduce_scatter"} and all(v is not None and v > 0 for v in by.values()), by
    assert r["boundary_exchange_ms"] == by[collective] and r["key_exchange_ms"] is not None
    assert "gather" in r["roofline"]["kernel"] or r["roofline"]["kernel"].startswith("seg_tile_kernel<float, 4, true"), r["roofline"]["kernel"]
    assert 0 < r["roofline"]["frac"] < 1


def test_bench_cfg5_world8_aligned_cuts_need_no_collective():
    r = _bench(["--cuts", "aligned"])
    assert r["ranks_seen"] == WORLD and r["cuts"] == "aligned"
    assert all(v is None for v in r["boundary_exchange_ms_by_collective"].values()), r["boundary_exchange_ms_by_collective"]
    assert r["key_exchange_ms"] is not None and r["value"] > 1e8


def test_bench_cfg5_world8_strong():
    r = _bench(["--strong"])
    assert r["ranks_seen"] == WORLD and r["scaling"] == "strong" and r["value"] > 1e8
    # the full (scaled) configuration, cut into 8: ~2 M edges per rank at --scale 0.01
    assert abs(r["config"]["nnz_per_gpu"] * WORLD - int(1_615_685_872 * 0.01)) < 0.02 * 1_615_685_872 * 0.01


# ---- the RCCL branch of the protocol at world > 1 (one GPU: the transport is shimmed, the code path is the real one) ------------------
def _shim_rccl():
    """Make torch.distributed look like an RCCL group to geot_amd.sharding while the bytes travel over gloo: `get_backend` says
    "nccl", and the collectives sharding.py uses accept DEVICE tensors (staged through the host here; RCCL moves them over xGMI).
    With this, `_sharded_reduce`'s device branch - key all_gather on a side stream, pinned copy, event, collectives queued on the
    stream - runs with world > 1 on one GPU (it had only ever run at world 1)."""
    import torch.distributed as dist
    real = {n: getattr(dist, n) for n in ("all_gather_into_tensor", "reduce_scatter_tensor", "all_reduce", "all_to_all_single")}

    def staged(name, out_pos, in_pos):
        def call(*args, **kw):
            args = list(args)
            devs = [a for a in args if torch.is_tensor(a) and a.is_cuda]
            if not devs:
                return real[name](*args, **kw)
            torch.cuda.current_stream().synchronize()            # (gloo reads host memory: the producer kernels must be done)
            outs = {}
            for i, a in enumerate(args):
                if torch.is_tensor(a) and a.is_cuda:
                    outs[i] = a
                    args[i] = a.cpu()
            r = real[name](*args, **kw)
            for i in out_pos:
                if i in outs:
                    outs[i].copy_(args[i])
            return r
        return call
    dist.all_gather_into_tensor = staged("all_gather_into_tensor", (0,), (1,))
    dist.reduce_scatter_tensor = staged("reduce_scatter_tensor", (0,), (1,))
    dist.all_reduce = staged("all_reduce", (0,), (0,))
    dist.all_to_all_single = staged("all_to_all_single", (0,), (1,))
    dist.get_backend = lambda group=None: "nccl"


def _rccl_rank(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import geot_amd
        from geot_amd import sharding
        _shim_rccl()
        res = {}
        si, di, w, x = (torch.from_numpy(a).cuda() for a in _global_list(True))
        cuts = sharding.equal_edge_cuts(di.numel(), world)
        e0, e1 = cuts[rank], cuts[rank + 1]
        si_s, di_s, w_s = si[e0:e1].clone(), di[e0:e1].clone(), w[e0:e1].clone()
        full = geot_amd.gather_weight_scatter(si, di, w, x)
        for coll in ("all_gather", "reduce_scatter"):
            timing = {}
            for it in range(3):                                   # (the second call launches on the remembered keys, underneath the key exchange)
                out, first = sharding.sharded_gather_scatter(si_s, di_s, x, weight_shard=w_s, collective=coll, timing=timing)
            res[coll] = (first, out.shape[0], bool(torch.equal(out, full[first:first + out.shape[0]])),
                         len(timing.get("key_events", [])), len(timing.get("exchange_events", [])), "key_wall_ms" in timing)
        # node-sharded sources on the HIP kernels: the halo by all_to_all_single and the all_gather form, against the replicated rows
        nodes = x.shape[0]
        offs = [nodes * r // world for r in range(world + 1)]
        rep, first = sharding.sharded_gather_scatter(si_s, di_s, x, weight_shard=w_s)
        for name, above in (("halo", 2.0), ("all_gather", 0.0)):
            timing = {}
            out, f2 = sharding.sharded_gather_scatter_node(si_s, di_s, x[offs[rank]:offs[rank + 1]].contiguous(), offs, weight_shard=w_s,
                                                           all_gather_above=above, timing=timing,
                                                           halo=sharding.HaloPlan.build(si_s, offs, None, above))
            res["node_" + name] = (f2, out.shape[0], bool(f2 == first and torch.equal(out, rep)), timing["halo_plan"].mode,
                                   len(timing.get("fetch_events", [])))
        q.put((rank, res, full.shape[0]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_rccl_branch_of_the_protocol_runs_at_world_4_on_one_gpu():
    import torch.multiprocessing as mp
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    K = got[0][2]
    for coll in ("all_gather", "reduce_scatter"):
        row = 0
        for rank, res, _ in got:
            first, n, ok, key_events, exchange_events, via_host = res[coll]
            assert ok and first == row, (coll, rank)
            assert key_events == 3 and exchange_events == 3 and not via_host, (coll, rank, key_events, exchange_events, via_host)   # the DEVICE branch ran
            row += n
        assert row == K
    for name in ("node_halo", "node_all_gather"):
        for rank, res, _ in got:
            f2, n, ok, mode, fetches = res[name]
            assert ok and mode == name[5:] and fetches == 1, (name, rank, mode)


def test_bench_cfg5_world8_node_sharded_sources():
    """`bench.py --workload cfg5 --src-sharding node`: every rank holds 1/8 of the source rows and fetches what its edge range references
    per step; the line reports the plan's mode and the bytes exchanged.  Uniform-random sources reference (almost) every row: the plan
    takes the all_gather form; the timed step includes the fetch."""
    r = _bench(["--src-sharding", "node", "--full"])                  # (--full: the whole record on stdout; the compact line keeps numbers only)
    assert r["ranks_seen"] == WORLD and r["src_sharding"] == "node" and r["value"] > 1e7
    f = r["source_rows_fetch"]
    assert f["mode"] in ("halo", "all_gather") and f["bytes_fetched_per_step_rank0"] > 0 and f["fetch_ms_rank0"] > 0
    assert f["src_rows_on_this_rank"] * WORLD <= f["src_rows_total"] + WORLD
    assert "sharded by node" in r["metric"] and "sharded by node" in r["config"]["workload"]
