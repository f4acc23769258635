"""Multi-GPU path on CPU: world_size-2 (and 3) gloo process groups exercise the edge sharding, the
boundary-row exchange (the one data-path collective) and the row ownership rules.  The per-rank
reduction is the CPU oracle injected as `local_op` (tests may use the oracle as the checker; the
product default is the HIP operator)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, powerlaw_index, sorted_index


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_local_op(index_local, src_local, rows):
    from oracle import api
    return torch.from_numpy(api.index_scatter(index_local.numpy(), src_local.numpy(), rows=rows))


def _worker(rank, world, port, case, aligned, q, collective="all_gather"):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from geot_amd import sharding
        index = torch.from_numpy(case["index"])
        src = torch.from_numpy(case["src"])
        ish, ssh = sharding.shard_edges(index, src, world, rank, aligned=aligned)
        timing = {}
        out, first_row = sharding.sharded_index_scatter(ish, ssh, local_op=_oracle_local_op, exchange=not aligned, collective=collective,
                                                        timing=timing)
        assert len(timing.get("key_wall_ms", [])) == (1 if world > 1 else 0)          # the key exchange times itself
        q.put((rank, first_row, out.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(case, world, aligned, collective="all_gather"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, aligned, q, collective)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _cases():
    rng = np.random.default_rng(5)
    hub = np.sort(np.concatenate([np.full(3000, 11), rng.integers(0, 30, 400)])).astype(np.int64)
    gaps = np.sort(rng.integers(0, 40, 500)).astype(np.int64) * 7 + 3
    return {
        "uniform": dict(index=sorted_index(rng, 2000, 150), src=rng.random((2000, 8), dtype=np.float32)),
        "powerlaw": dict(index=powerlaw_index(5000, 300, 1), src=rng.random((5000, 4), dtype=np.float32)),
        "hub_spans_ranks": dict(index=hub, src=rng.random((len(hub), 4), dtype=np.float32)),
        "gaps_and_offset": dict(index=gaps, src=rng.random((500, 3), dtype=np.float32)),
    }


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", ["uniform", "powerlaw", "hub_spans_ranks", "gaps_and_offset"])
def test_equal_edge_cuts_with_boundary_exchange(name, world):
    from oracle import api
    case = _cases()[name]
    res = _run(case, world, aligned=False)
    full = api.index_scatter(case["index"], case["src"], acc64=True)
    row = 0
    for rank, first_row, out in res:
        assert first_row == row, (rank, first_row, row)
        row += out.shape[0]
    assert row == full.shape[0]
    got = np.concatenate([o for _, _, o in res])
    np.testing.assert_allclose(got, full, rtol=1e-5, atol=1e-6)


def _both_worker(rank, world, port, cases, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from geot_amd import sharding
        res = {}
        for name, case in cases.items():
            ish, ssh = sharding.shard_edges(torch.from_numpy(case["index"]), torch.from_numpy(case["src"]), world, rank)
            for coll in ("all_gather", "reduce_scatter"):
                out, first_row = sharding.sharded_index_scatter(ish, ssh, local_op=_oracle_local_op, collective=coll)
                res[(name, coll)] = (first_row, out.numpy().copy())
        q.put((rank, res))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_reduce_scatter_form_gives_the_rows_of_the_all_gather_form(world):
    """BASELINE.json's north star names an "RCCL reduce-scatter over xGMI for boundary segments": every rank contributes
    a [W, F] buffer whose row `owner(my first key)` is its first-row partial.  Same rows as the all_gather + owner-add form
    for every cut pattern - one shared key per cut, a hub covering several whole ranks, tiny shards, gaps.  With
    integer-valued data (every partial sum exact) the two forms agree bit for bit; with random data within rounding."""
    from oracle import api
    rng = np.random.default_rng(100 + world)
    hub = np.sort(np.concatenate([rng.integers(0, 5, 700), np.full(5000, 5), rng.integers(6, 40, 1500)])).astype(np.int64)
    cases = dict(_cases())
    cases["hub_over_many_ranks"] = dict(index=hub, src=rng.random((len(hub), 4), dtype=np.float32))
    cases["tiny"] = dict(index=np.array([0, 0, 0, 1, 1, 1, 1, 1, 4, 4, 4, 9, 9, 9, 9, 9], dtype=np.int64), src=rng.random((16, 3), dtype=np.float32))
    for name in list(cases):                                 # + integer-valued twins: sums of small integers are exact
        cases[name + "/exact"] = dict(index=cases[name]["index"], src=np.floor(cases[name]["src"] * 64).astype(np.float32))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_both_worker, args=(r, world, port, cases, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for name, case in cases.items():
        full = api.index_scatter(case["index"], case["src"], acc64=True)
        for coll in ("all_gather", "reduce_scatter"):
            got = np.concatenate([res[r][(name, coll)][1] for r in range(world)])
            assert got.shape == full.shape, (name, coll)
            if name.endswith("/exact"):
                np.testing.assert_array_equal(got, full)
            else:
                np.testing.assert_allclose(got, full, rtol=1e-5, atol=1e-5)
        for r in range(world):
            a, b = res[r][(name, "all_gather")], res[r][(name, "reduce_scatter")]
            assert a[0] == b[0] and a[1].shape == b[1].shape, (name, r)
            if name.endswith("/exact"):
                np.testing.assert_array_equal(a[1], b[1])


def test_eight_ranks_hub_over_many_ranks_and_tiny_shards():
    """The world size of BASELINE.json configs[4]: a hub that covers ranks 2..5 completely (their whole shard lies inside
    one run owned by rank 1), then 8 ranks with TWO edges each."""
    from oracle import api
    rng = np.random.default_rng(8)
    hub = np.sort(np.concatenate([rng.integers(0, 5, 700), np.full(5000, 5), rng.integers(6, 40, 1500)])).astype(np.int64)
    case = dict(index=hub, src=rng.random((len(hub), 4), dtype=np.float32))
    res = _run(case, 8, aligned=False)
    full = api.index_scatter(case["index"], case["src"], acc64=True)
    got = np.concatenate([o for _, _, o in res])
    assert got.shape == full.shape and sum(o.shape[0] for _, _, o in res) == full.shape[0]
    np.testing.assert_allclose(got, full, rtol=1e-5, atol=1e-5)
    assert any(o.shape[0] == 0 for _, _, o in res)                      # ranks inside the hub own no row
    row = 0
    for _, first_row, out in res:
        assert first_row == row or out.shape[0] == 0
        row += out.shape[0]
    tiny = dict(index=np.array([0, 0, 0, 1, 1, 1, 1, 1, 4, 4, 4, 9, 9, 9, 9, 9], dtype=np.int64),
                src=rng.random((16, 3), dtype=np.float32))
    res = _run(tiny, 8, aligned=False)
    np.testing.assert_allclose(np.concatenate([o for _, _, o in res]), api.index_scatter(tiny["index"], tiny["src"], acc64=True),
                               rtol=1e-6, atol=1e-6)


def test_boundary_plan_table():
    """The ownership rules on their own (every rank computes the same table from the gathered keys)."""
    from geot_amd.sharding import boundary_plan
    firsts, lasts = [0, 5, 5, 5, 9], [5, 5, 5, 7, 12]                    # key 5 runs from rank 0 into rank 3
    p0, p1, p2, p3, p4 = (boundary_plan(firsts, lasts, r) for r in range(5))
    assert p0["owns_first"] and p0["joins"] == [1, 2, 3] and p0["first_row"] == 0 and p0["any_shared"]
    assert not p1["owns_first"] and p1["joins"] == [] and p1["first_row"] == 6
    assert not p2["owns_first"] and p2["joins"] == []
    assert not p3["owns_first"] and p3["joins"] == [] and p3["first_row"] == 6
    assert p4["owns_first"] and p4["gap"] == 1 and p4["first_row"] == 8 and p4["joins"] == []
    assert [p["owner"] for p in (p0, p1, p2, p3, p4)] == [0, 0, 0, 0, 4]    # who receives each rank's first-row partial
    q = [boundary_plan([0, 4, 9], [3, 8, 9], r) for r in range(3)]       # segment-aligned cuts: nothing shared
    assert not any(x["any_shared"] for x in q) and all(x["owns_first"] and x["joins"] == [] for x in q)
    assert [x["gap"] for x in q] == [0, 0, 0] and [x["first_row"] for x in q] == [0, 4, 9]


@pytest.mark.parametrize("name", ["uniform", "gaps_and_offset"])
def test_segment_aligned_cuts_need_no_exchange(name):
    from oracle import api
    case = _cases()[name]
    res = _run(case, 2, aligned=True)
    got = np.concatenate([o for _, _, o in res])
    np.testing.assert_array_equal(got, api.index_scatter(case["index"], case["src"]))   # bit-exact: no re-association


def _gather_worker(rank, world, port, case, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from geot_amd import sharding
        from oracle import api

        def local_op(si, di, w, x, rows):
            return torch.from_numpy(api.gather_weight_scatter(si.numpy(), di.numpy(), None if w is None else w.numpy(),
                                                              x.numpy(), rows=rows))
        cuts = sharding.equal_edge_cuts(len(case["dst"]), world)
        sl = slice(cuts[rank], cuts[rank + 1])
        t = torch.from_numpy
        out, first = sharding.sharded_gather_scatter(t(case["si"][sl]), t(case["dst"][sl]), t(case["x"]),
                                                     weight_shard=t(case["w"][sl]), local_op=local_op)
        q.put((rank, first, out.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_gather_weight_scatter_replicated_src():
    from oracle import api
    rng = np.random.default_rng(9)
    nodes, nnz, F = 200, 6000, 8
    case = dict(dst=powerlaw_index(nnz, nodes, 2), si=rng.integers(0, nodes, nnz).astype(np.int64),
                w=rng.random(nnz, dtype=np.float32), x=rng.random((nodes, F), dtype=np.float32))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 3, port, case, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(3)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = np.concatenate([o for _, _, o in res])
    full = api.gather_weight_scatter(case["si"], case["dst"], case["w"], case["x"], acc64=True)
    assert got.shape == full.shape and res[0][1] == 0
    np.testing.assert_allclose(got, full, rtol=1e-5, atol=1e-6)


def _node_worker(rank, world, port, case, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from geot_amd import sharding
        from oracle import api

        def local_op(si, di, w, x, rows, reduce="sum"):
            if reduce != "sum":
                raise NotImplementedError
            return torch.from_numpy(api.gather_weight_scatter(si.numpy(), di.numpy(), None if w is None else w.numpy(),
                                                              x.numpy(), rows=rows))
        t = torch.from_numpy
        cuts = sharding.equal_edge_cuts(len(case["dst"]), world)
        sl = slice(cuts[rank], cuts[rank + 1])
        offs = case["node_offsets"]
        x_mine = t(case["x"][offs[rank]:offs[rank + 1]])
        res = {}
        rep, first = sharding.sharded_gather_scatter(t(case["si"][sl]), t(case["dst"][sl]), t(case["x"]), weight_shard=t(case["w"][sl]),
                                                     local_op=local_op)
        res["replicated"] = (first, rep.numpy().copy())
        for name, above in (("halo", 2.0), ("all_gather", 0.0), ("rule", 0.5)):
            timing = {}
            out, first = sharding.sharded_gather_scatter_node(t(case["si"][sl]), t(case["dst"][sl]), x_mine, offs, weight_shard=t(case["w"][sl]),
                                                              local_op=local_op, all_gather_above=above, timing=timing,
                                                              halo=sharding.HaloPlan.build(t(case["si"][sl]), offs, None, above))
            plan = timing["halo_plan"]
            res[name] = (first, out.numpy().copy(), plan.mode, plan.rows_fetched, plan.table_rows)
        # the remembered plan (no `halo=`): built on the first call, found on the second - for the SAME tensor object only
        builds = []
        real_build = sharding.HaloPlan.build
        sharding.HaloPlan.build = staticmethod(lambda *a, **k: (builds.append(1), real_build(*a, **k))[1])
        si_np = case["si"][sl].copy()
        si_t = t(si_np)
        a, _ = sharding.sharded_gather_scatter_node(si_t, t(case["dst"][sl]), x_mine, offs, weight_shard=t(case["w"][sl]), local_op=local_op)
        b, _ = sharding.sharded_gather_scatter_node(si_t, t(case["dst"][sl]), x_mine, offs, weight_shard=t(case["w"][sl]), local_op=local_op)
        assert torch.equal(a, b) and torch.equal(a, rep) and len(builds) == 1
        # another edge shard of the same length at the SAME address with version 0 (what the caching allocator hands a fixed-fanout
        # sampler): a different object - the plan is rebuilt (on every rank alike), never silently the old graph's sources
        si_np[:] = np.roll(case["si"], 17)[sl]                            # (a numpy write: no version counter sees it)
        si_new = t(si_np)
        assert si_new.data_ptr() == si_t.data_ptr() and si_new._version == si_t._version
        c, _ = sharding.sharded_gather_scatter_node(si_new, t(case["dst"][sl]), x_mine, offs, weight_shard=t(case["w"][sl]), local_op=local_op)
        want, _ = sharding.sharded_gather_scatter(si_new, t(case["dst"][sl]), t(case["x"]), weight_shard=t(case["w"][sl]), local_op=local_op)
        assert len(builds) == 2 and torch.equal(c, want)
        sharding.HaloPlan.build = real_build
        q.put((rank, res))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_node_sharded_sources_give_the_replicated_rows_bit_for_bit(world):
    """SURVEY.md section 8(e), second option: every rank owns a contiguous range of source rows and fetches the rows its edge range
    references with one all_to_all_single ("halo") - or all shards with one all_gather when the halo is most of the table.  The local
    kernel sees the same values in the same edge order: the rows equal the replicated form bit for bit.  Uneven node ranges (one rank
    owns no node at world 8), sources with locality (small halos) and without."""
    rng = np.random.default_rng(60 + world)
    nodes, nnz, F = 400, 9000, 6
    dst = powerlaw_index(nnz, nodes, 4)
    near = np.clip(dst + rng.integers(-25, 26, nnz), 0, nodes - 1)                # sources near their destination: small halos
    si = np.where(rng.random(nnz) < 0.8, near, rng.integers(0, nodes, nnz)).astype(np.int64)
    cuts = sorted(rng.choice(np.arange(1, nodes), size=world - 1, replace=False).tolist())
    if world == 8:
        cuts[3] = cuts[2]                                                          # a rank that owns no node
    case = dict(dst=dst, si=si, w=rng.random(nnz, dtype=np.float32), x=rng.random((nodes, F), dtype=np.float32),
                node_offsets=[0] + cuts + [nodes])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_node_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        first, rep = res[r]["replicated"]
        for name in ("halo", "all_gather", "rule"):
            f2, out, mode, fetched, table_rows = res[r][name]
            assert f2 == first and np.array_equal(out, rep), (r, name)
            if name != "rule":
                assert mode == name
        own = case["node_offsets"][r + 1] - case["node_offsets"][r]
        assert res[r]["all_gather"][3] == nodes - own                                  # every other rank's rows
        assert res[r]["halo"][3] < nodes - own or world == 2                           # the halo: only what the edge range references
        assert res[r]["halo"][4] <= nodes


def _repeat_worker(rank, world, port, case, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from geot_amd import sharding
        index = torch.from_numpy(case["index"].copy())
        src = torch.from_numpy(case["src"])
        cuts = sharding.equal_edge_cuts(index.numel(), world)
        ish, ssh = index[cuts[rank]:cuts[rank + 1]], src[cuts[rank]:cuts[rank + 1]]
        outs = []
        for it in range(4):
            if it == 2 and rank == world - 1:
                ish.data += 5                       # same tensor identity and version, different keys:
            out, first = sharding.sharded_index_scatter(ish, ssh, local_op=_oracle_local_op)
            outs.append((first, out.numpy().copy()))
        q.put((rank, outs))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_repeated_calls_use_and_verify_remembered_keys():
    """Calls 2+ take the speculative path (remembered end keys / ownership); call 3 changes the last rank's
    keys under the same tensor identity: every rank must notice and still produce the exact result, with
    exactly one collective per call on every rank (no hang)."""
    from oracle import api
    rng = np.random.default_rng(17)
    case = dict(index=powerlaw_index(6000, 400, 5), src=rng.random((6000, 4), dtype=np.float32))
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_repeat_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cuts = [(6000 * r) // world for r in range(world + 1)]
    for it in range(4):
        index = case["index"].copy()
        if it >= 2:
            index[cuts[world - 1]:] += 5
        full = api.index_scatter(index, case["src"], acc64=True)
        got = np.concatenate([res[r][it][1] for r in range(world)])
        assert got.shape == full.shape, (it, got.shape, full.shape)
        np.testing.assert_allclose(got, full, rtol=1e-5, atol=1e-6)
        row = 0
        for r in range(world):
            assert res[r][it][0] == row
            row += res[r][it][1].shape[0]


def test_cut_helpers():
    from geot_amd import sharding
    assert sharding.equal_edge_cuts(10, 4) == [0, 2, 5, 7, 10]
    idx = torch.tensor([0, 0, 0, 1, 1, 2, 2, 2, 2, 5])
    cuts = sharding.segment_aligned_cuts(idx, 2)
    assert cuts[0] == 0 and cuts[-1] == 10 and idx[cuts[1] - 1] != idx[cuts[1]]
    cuts = sharding.segment_aligned_cuts(torch.zeros(8, dtype=torch.int64), 4)          # one hub: later shards empty
    assert cuts == [0, 8, 8, 8, 8]


@pytest.mark.parametrize("cuts", ["equal", "aligned"])
def test_bench_global_list_shards_concatenate_to_one_list_whatever_the_world(cuts):
    """bench.py --workload cfg5 (BASELINE.json configs[4]) cuts ONE global dst-sorted list into the ranks' edge ranges without
    any rank materialising it: the shards of world 1 / 2 / 3 / 8 concatenate to the same list (same keys, same sources), the
    cuts are equal_edge_cuts' (or snapped to row starts: no key shared by two ranks), local keys start at 0."""
    import bench
    from geot_amd import sharding
    ref = None
    for world in (1, 2, 3, 8):
        parts = [bench.global_list_shard(5000, 60000, 7000, world, r, cuts, 13, "cpu") for r in range(world)]
        nnz, edges = parts[0][4], parts[0][5]
        full = torch.cat([p[0] + p[2] for p in parts])
        si = torch.cat([p[1] for p in parts])
        assert full.numel() == nnz and bool((full[1:] >= full[:-1]).all()) and int(full[-1]) == 4999
        assert 0 <= int(si.min()) and int(si.max()) < 7000
        if ref is None:
            ref = (full, si)
        assert torch.equal(full, ref[0]) and torch.equal(si, ref[1])
        for r, p in enumerate(parts):
            assert p[0].numel() == edges[r + 1] - edges[r] and int(p[0][0]) == 0 and int(p[0][-1]) == p[3] - 1
        if cuts == "equal":
            assert edges == sharding.equal_edge_cuts(nnz, world)
        else:
            assert edges == sharding.segment_aligned_cuts(full, world)
            assert all(int(a[0][-1]) + a[2] != b[2] for a, b in zip(parts, parts[1:]))


# ---- reductions other than sum across ranks (csrc/cpu/index_scatter_cpu.cpp:124-134 is the semantics) -----------------------------
def _oracle_reduce_local_op(index_local, src_local, rows, reduce="sum"):
    from oracle import api
    return torch.from_numpy(api.index_scatter_3pass(index_local.numpy(), src_local.numpy(), reduce, rows=rows))


def _reduce_worker(rank, world, port, cases, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from geot_amd import sharding
        res = {}
        for name, case in cases.items():
            ish, ssh = sharding.shard_edges(torch.from_numpy(case["index"]), torch.from_numpy(case["src"]), world, rank)
            for red in ("mean", "max", "min", "prod", "amax"):
                for coll in ("all_gather", "reduce_scatter"):
                    out, first_row = sharding.sharded_index_scatter(ish, ssh, local_op=_oracle_reduce_local_op, collective=coll, reduce=red)
                    res[(name, red, coll)] = (first_row, out.numpy().copy())
        q.put((rank, res))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_mean_max_min_prod_equal_the_unsharded_reduction(world):
    """mean ships (partial sum, edge count) of the shared rows and the owner divides once; max / min / prod ship the partial row
    (identity elsewhere in the reduce_scatter form, ReduceOp.MAX / MIN / PRODUCT).  Hub over several whole ranks, gaps, tiny
    shards; rows without edges stay 0 for every reduction (the reference's zero-initialised output)."""
    from oracle import api
    rng = np.random.default_rng(200 + world)
    hub = np.sort(np.concatenate([rng.integers(0, 5, 700), np.full(5000, 5), rng.integers(6, 40, 1500)])).astype(np.int64)
    cases = {k: v for k, v in _cases().items() if k in ("powerlaw", "gaps_and_offset")}
    cases["hub_over_many_ranks"] = dict(index=hub, src=rng.random((len(hub), 4), dtype=np.float32))
    cases["tiny"] = dict(index=np.array([0, 0, 0, 1, 1, 1, 1, 1, 4, 4, 4, 9, 9, 9, 9, 9], dtype=np.int64), src=rng.random((16, 3), dtype=np.float32))
    for c in cases.values():                                 # values around 1 with both signs: prod stays finite, max / min are not trivial
        c["src"] = (0.9 + 0.2 * c["src"]) * np.where(rng.random(c["src"].shape) < 0.3, -1, 1).astype(np.float32)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reduce_worker, args=(r, world, port, cases, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for name, case in cases.items():
        for red in ("mean", "max", "min", "prod", "amax"):
            full = api.index_scatter_3pass(case["index"], case["src"], "max" if red == "amax" else red)
            for coll in ("all_gather", "reduce_scatter"):
                row = 0
                for r in range(world):
                    first_row, out = res[r][(name, red, coll)]
                    assert first_row == row or out.shape[0] == 0, (name, red, coll, r)
                    row += out.shape[0]
                got = np.concatenate([res[r][(name, red, coll)][1] for r in range(world)])
                assert got.shape == full.shape, (name, red, coll)
                if red in ("max", "min", "amax"):
                    np.testing.assert_array_equal(got, full)                      # selections: exact whatever the grouping
                else:
                    np.testing.assert_allclose(got, full, rtol=2e-5, atol=1e-6)


def _oracle_reduce_local_op_16(index_local, src_local, rows, reduce="sum"):
    """16-bit storage the way the reference's CPU path treats it (csrc/cpu/index_scatter_cpu.cpp:78-86): fp32 accumulation, one
    rounding to the storage type at the end."""
    from oracle import api
    out = api.index_scatter_3pass(index_local.numpy(), src_local.float().numpy(), reduce, rows=rows)
    return torch.from_numpy(out).to(src_local.dtype)


def _mean16_worker(rank, world, port, case, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from geot_amd import sharding
        res = {}
        index = torch.from_numpy(case["index"])
        for dt in (torch.float16, torch.bfloat16):
            src = torch.from_numpy(case["src"]).to(dt)
            ish, ssh = sharding.shard_edges(index, src, world, rank)
            for coll in ("all_gather", "reduce_scatter"):
                out, first_row = sharding.sharded_index_scatter(ish, ssh, local_op=_oracle_reduce_local_op_16, collective=coll, reduce="mean")
                res[(str(dt), coll)] = (first_row, out.float().numpy().copy())
        q.put((rank, res))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_mean_of_16bit_rows_never_forms_a_sum_in_the_storage_type(world):
    """the shared rows of a sharded mean used to travel as a partial SUM written in the storage type - a hub of
    200 000 edges of values around 1 shared by the ranks is +inf in fp16 (max 65 504) and loses everything below 2^-8 of the
    running sum in bf16, while the unsharded operator accumulates in fp32 and rounds once.  Now: mean x count in float64."""
    from oracle import api
    rng = np.random.default_rng(300 + world)
    hub = np.sort(np.concatenate([rng.integers(0, 3, 500), np.full(200_000, 3), rng.integers(4, 20, 900)])).astype(np.int64)
    case = dict(index=hub, src=(0.75 + 0.5 * rng.random((len(hub), 3), dtype=np.float32)))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mean16_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for dt, ulp in ((torch.float16, 2.0 ** -10), (torch.bfloat16, 2.0 ** -7)):
        src16 = torch.from_numpy(case["src"]).to(dt)
        full = torch.from_numpy(api.index_scatter_3pass(case["index"], src16.float().numpy(), "mean")).to(dt).float().numpy()
        for coll in ("all_gather", "reduce_scatter"):
            got = np.concatenate([res[r][(str(dt), coll)][1] for r in range(world)])
            assert got.shape == full.shape
            assert np.isfinite(got).all(), (dt, coll)
            # one ulp of the storage type around the unsharded result (values around 1: ulp = 2^-10 / 2^-7)
            np.testing.assert_allclose(got, full, rtol=0, atol=1.01 * ulp)


def test_sharded_reduce_rejects_unknown_names():
    from geot_amd import sharding
    with pytest.raises(ValueError, match="reduce argument must be either sum, prod, mean, amax or amin"):
        sharding._sharded_reduce(torch.zeros(1, dtype=torch.int64), (1,), torch.float32, torch.device("cpu"), None, None, True, None,
                                 None, "all_gather", "median")
