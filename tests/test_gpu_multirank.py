"""Two bench.py ranks on ONE GPU (gloo rendezvous, GEOT_DIST_BACKEND=gloo): exercises the distributed
code path of bench.py and geot_amd.sharding with the HIP operator as the per-rank reduction."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, powerlaw_index

pytestmark = pytest.mark.gpu


def test_bench_two_ranks_one_gpu():
    env = dict(os.environ, GEOT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29611", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "5", "--warmup", "2"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["value"] > 1e9 and r["unit"] == "edges/s"
    assert r["roofline"]["bound"] == "hbm" and 0 < r["roofline"]["frac"] < 1
    leg = r["secondary"]["gather_scatter_cfg5"]                       # configs[4] rides along at every N (a hundredth of it where ranks share a GPU)
    assert "error" not in leg and leg["n_gpus"] == 2 and leg["value"] > 1e8 and leg["scaling"] == "weak"


def _rank(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import geot_amd
        from geot_amd import sharding
        index = torch.from_numpy(powerlaw_index(300_000, 20_000, 4)).cuda()
        torch.manual_seed(0)
        src = torch.rand(300_000, 64, device="cuda")
        ish, ssh = sharding.shard_edges(index, src, world, rank)
        ish, ssh = ish.clone(), ssh.contiguous()
        ok = True
        for it in range(4):                                    # calls 2+ speculate on the remembered keys
            if it == 2:                                        # every rank shifts its copy of the LAST shard's keys
                cuts = sharding.equal_edge_cuts(index.numel(), world)
                index.data[cuts[world - 1]:] += 7
                if rank == world - 1:
                    ish.data += 7
            out, first = sharding.sharded_index_scatter(ish, ssh)
            full = geot_amd.index_scatter(0, src, index)
            ok = ok and out.shape[0] > 0 and first + out.shape[0] <= full.shape[0] and bool(
                torch.allclose(out, full[first:first + out.shape[0]], rtol=1e-5, atol=1e-6))
        q.put((rank, first, out.shape[0], bool(ok)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_hip_operator_matches_unsharded():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, 3, 29613, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
    assert all(ok for *_, ok in res), res
    assert res[0][1] == 0 and sum(n for _, _, n, _ in res) == 20_007
    for (r0, f0, n0, _), (r1, f1, n1, _) in zip(res, res[1:]):
        assert f0 + n0 == f1


def test_rccl_calls_used_by_the_multi_gpu_path_work_here():
    """Single-rank "nccl" (= RCCL) group: the exact collective calls bench.py / sharding.py make at N > 1."""
    script = r'''
import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
# the two collectives of geot_amd/sharding.py: every rank's (first_key, last_key) as int64, first-row partials as fp32
keys = torch.tensor([3, 2 ** 40 + 7], dtype=torch.int64, device=dev)
allkeys = torch.empty(2, dtype=torch.int64, device=dev)
dist.all_gather_into_tensor(allkeys, keys)
host = torch.empty(2, dtype=torch.int64).pin_memory()
host.copy_(allkeys, non_blocking=True)
send = torch.arange(64, dtype=torch.float32, device=dev)
recv = torch.empty(64, dtype=torch.float32, device=dev)
dist.all_gather_into_tensor(recv, send)
# ... and the reduce_scatter form of the boundary exchange: a [W, F] buffer in, this rank's row of the sum out
rs_in = torch.arange(64, dtype=torch.float32, device=dev) + 0.5
rs_out = torch.empty(64, dtype=torch.float32, device=dev)
dist.reduce_scatter_tensor(rs_out, rs_in)
assert torch.equal(rs_out, rs_in)
# bench.py: max over ranks of the elapsed time, barrier
t = torch.tensor([1.5], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert host.tolist() == [3, 2 ** 40 + 7] and torch.equal(recv, send) and t.item() == 1.5 and dist.get_backend() == "nccl"
# and the whole sharded call on a 1-rank RCCL group (no collective is needed at world 1; the code path is the GPU one)
import sys
sys.path.insert(0, os.environ["GEOT_ROOT"])
import geot_amd
from geot_amd import sharding
idx = torch.sort(torch.randint(0, 5000, (300000,), device=dev)).values
src = torch.rand(300000, 64, device=dev)
ref = geot_amd.index_scatter(0, src, idx)
for coll in ("all_gather", "reduce_scatter"):
    for _ in range(3):
        timing = {}
        out, first = sharding.sharded_index_scatter(idx, src, collective=coll, timing=timing)
    assert first == 0 and torch.equal(out, ref) and len(timing["key_events"]) == 1
dist.destroy_process_group()
print("RCCL OK")
'''
    p = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GEOT_ROOT=ROOT))
    assert p.returncode == 0 and "RCCL OK" in p.stdout, p.stderr[-2000:]


def test_bench_json_contract_single_gpu():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line"
    r = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in r, key
    assert r["metric"] == "aggregated edges/sec + HBM GB/s, index_scatter feat=64 sorted sum" and r["unit"] == "edges/s"
    assert r["n_gpus"] == 1 and r["steps"] == 5 and r["warmup"] == 2 and r["higher_is_better"] is True
    assert r["dtype"] == "f32" and r["data"] == "synthetic" and r["vs_baseline"] is None and "workload" in r["config"]
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.3 < rf["frac"] < 1.0
    assert rf["traffic"] is None or rf["traffic"] > 2.8e9
    cb = r["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["unit"] == "edges/s" and cb["cores"] >= 1 and cb["value"] > 1e6
    assert r["value"] > 1e9 and abs(r["value"] - 10_000_000 / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-6
    # BASELINE.json's other configs ride along as `secondary`: configs[0], configs[2] (four stand-ins), configs[3] (two), configs[4]'s shard
    sec = r["secondary"]
    for name in ("cfg1", "gws_cfg3", "gws_cfg3_local", "gws_cfg3_powerlaw_src", "gws_cfg3_blockmodel", "mh_spmm_cfg4", "mh_spmm_cfg4_powerlaw_src",
                 "gather_scatter_cfg5"):
        assert name in sec and "error" not in sec[name], (name, sec.get(name))
        assert 0 < sec[name]["roofline"]["frac"] < 1 and "traffic_source" in sec[name]["roofline"]
    assert sec["cfg1"]["us_per_call_as_dispatched"] < 100 and sec["cfg1"]["cpu_baseline"]["value"] > 0
    assert sec["gather_scatter_cfg5"]["n_gpus"] == 1 and sec["gather_scatter_cfg5"]["value"] > 1e9
    assert sec["gws_cfg3_blockmodel"]["renumbered"]["speedup_vs_as_shipped"] > 1.2
    assert r["roofline"]["kernel"] == "seg_tile_kernel<float, 4, false, 0, false, 3, 3, 16>"       # from the library (geot_last_kernel)
