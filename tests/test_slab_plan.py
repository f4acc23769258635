"""Phase A of the source-blocked kernels (geot_amd/slab.py) on the CPU: the plan's arrays are run through a plain numpy
emulation of seg_slab_kernel (groups per round and unit, serpentine order, rows of a group in a local accumulator,
pieces of split hubs through carry slots) and the result is compared with the oracle.  The HIP kernel itself is
checked on the GPU (tests/test_gpu_slab.py)."""
import numpy as np
import pytest
import torch

from conftest import powerlaw_index


def emulate(plan, w, wmode, x, K, H):
    T = {k: v.numpy() for k, v in plan.tensors.items()}
    xs = x.reshape(x.shape[0], -1)
    F = xs.shape[1]
    out = np.zeros((K, F), np.float64)
    carry = np.zeros((max(plan.struct.n_carry, 1), F), np.float64)
    G, units, R = plan.struct.n_groups, plan.struct.units, plan.struct.rows_per_group
    seen = set()
    for r in range(-(-G // units)):
        for u in range(units):
            pos = r * units + ((units - 1 - u) if r & 1 else u)
            if pos >= G:
                continue
            assert pos not in seen
            seen.add(pos)
            e0, e1 = T["g_begin"][pos], T["g_begin"][pos + 1]
            dl = T["e_dl"][e0:e1].astype(np.int64)
            assert (dl < T["g_nv"][pos]).all() and T["g_nv"][pos] <= R
            rows = xs[T["e_src"][e0:e1]].astype(np.float64)
            pe = T["e_perm"][e0:e1]
            if wmode == 1:
                rows = rows * w[pe][:, None]
            elif wmode == 2:
                rows = rows * np.repeat(w[pe], F // H, axis=1)
            elif wmode == 3:
                rows = rows * np.repeat(w[:, pe].T, F // H, axis=1)
            acc = np.zeros((R, F), np.float64)
            np.add.at(acc, dl, rows)
            for l in range(T["g_nv"][pos]):
                t = T["v_out"][T["g_vrow0"][pos] + l]
                if t >= 0:
                    out[t] = acc[l]
                else:
                    carry[-t - 1] = acc[l]
    assert len(seen) == G
    for s in range(plan.struct.n_split):
        out[T["c_row"][s]] = carry[T["c_first"][s]: T["c_first"][s] + T["c_count"][s]].sum(0)
    return out


@pytest.mark.parametrize("nodes,nnz,F,units,R", [(300, 20000, 64, 8, 4), (50, 30000, 64, 16, 3), (2000, 40000, 128, 32, 5),
                                                 (10, 5000, 256, 4, 2), (400, 3000, 64, 64, 15)])
def test_plan_covers_every_edge_once_and_reproduces_the_oracle(oracle, nodes, nnz, F, units, R):
    from geot_amd import slab
    rng = np.random.default_rng(nodes)
    di = powerlaw_index(nnz, nodes, nodes)
    di[: nnz // 3] = di[nnz // 3]                       # a hub that must be split into virtual rows
    di = np.sort(di)
    di[di == 5] = 6                                     # an empty key
    si = rng.integers(0, nodes, nnz).astype(np.int64)
    w = rng.random(nnz).astype(np.float32)
    x = rng.random((nodes, F)).astype(np.float32)
    t = torch.from_numpy
    plan = slab.build_plan(t(si), t(di), nodes, nodes, F * 4, 1, 1, slab_bytes=F * 4 * 37, rows_per_group=R, units=units)
    assert sorted(plan.tensors["e_perm"].tolist()) == list(range(nnz))                  # a permutation of the edges
    assert plan.struct.n_split >= 1 and plan.meta["rounds"] >= 1
    # inside a group the edges are ordered by (slab, row in group)
    gb = plan.tensors["g_begin"].numpy()
    key = (plan.tensors["e_src"].numpy() // plan.meta["slab_rows"]) * 256 + plan.tensors["e_dl"].numpy()
    for p in range(plan.struct.n_groups):
        assert np.all(np.diff(key[gb[p]: gb[p + 1]]) >= 0)
    assert np.all(np.diff(np.diff(gb)) <= 0)                                            # groups by size, descending
    got = emulate(plan, w, 1, x, nodes, 1)
    ref = oracle.gather_weight_scatter(si, di, w, x, rows=nodes, acc64=True)
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-4)
    assert np.all(got[np.bincount(di, minlength=nodes) == 0] == 0)
    H = 4
    wh = rng.random((nnz, H)).astype(np.float32)
    x3 = x.reshape(nodes, H, F // H)
    ref = oracle.mh_spmm(si, di, wh, x3, rows=nodes, acc64=True).reshape(nodes, -1)
    plan = slab.build_plan(t(si), t(di), nodes, nodes, F * 4, 2, H, slab_bytes=F * 4 * 11, rows_per_group=R, units=units)
    np.testing.assert_allclose(emulate(plan, wh, 2, x3, nodes, H), ref, rtol=1e-5, atol=1e-4)
    plan = slab.build_plan(t(si), t(di), nodes, nodes, F * 4, 3, H, rows_per_group=R, units=units)
    np.testing.assert_allclose(emulate(plan, np.ascontiguousarray(wh.T), 3, x3, nodes, H), ref, rtol=1e-5, atol=1e-4)


def test_density_rule():
    from geot_amd import slab
    assert slab.worthwhile(114_615_892, 232_965, 232_965, 1024)          # BASELINE.json configs[3]: Reddit scale
    assert not slab.worthwhile(123_718_280, 2_449_029, 2_449_029, 512)   # configs[2]: ogbn-products scale
    assert not slab.worthwhile(201_960_734, 13_882_494, 111_059_956, 512)  # configs[4] shard
    assert slab.worthwhile(120_000_000, 450_000, 450_000, 1024)          # measured 1.62x
    assert slab.worthwhile(23_213_838, 232_965, 232_965, 1024)           # Reddit2 (the reference's benchmark/utils.py:41-43), H=4 F=64: 1.43x
    assert not slab.worthwhile(100_000_000, 1_000_000, 1_000_000, 1024)  # measured 0.68x
    assert slab.worthwhile(120_000_000, 600_000, 600_000, 512)           # measured 1.22x
    assert not slab.worthwhile(120_000_000, 800_000, 800_000, 512)       # measured 1.09x: not worth a 1 GB plan
    assert not slab.worthwhile(100_000_000, 1_000_000, 1_000_000, 512)   # measured 0.86x
    assert not slab.worthwhile(114_615_892, 232_965, 232_965, 1000)      # rows must be 256 / 512 / 1024 bytes
    assert not slab.worthwhile(1_000_000, 2000, 2000, 256)               # small problems stay on the tile kernel
