#!/usr/bin/env python3
"""Soak fuzz of the HIP path against the oracle at sizes the unit tests do not reach (up to a few million
edges: hundreds to thousands of tiles, so hub chains cross many 64-tile windows) with the library's internal
switches drawn at random as well ("hub" window sums on/off/auto, the three narrow-row implementations, forced
tile shapes).  Developer tool: `python tools/soak_fuzz.py [--iters 200] [--seed 0]`; exits non-zero on the
first mismatch and prints the reproducer."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import geot_amd as geot  # noqa: E402
from geot_amd import hip  # noqa: E402
from oracle import api as oracle  # noqa: E402
from test_gpu_fuzz import random_index  # noqa: E402


def big_index(rng, nnz):
    kind = rng.integers(0, 5)
    if kind == 0:     # a few keys, long runs (global pooling)
        keys = int(rng.integers(1, 12))
        return np.sort(rng.integers(0, keys, nnz)).astype(np.int64)
    if kind == 1:     # hubs of very different sizes between short runs
        parts, k = [], 0
        left = nnz
        while left > 0:
            n = int(min(left, rng.choice([1, 3, 17, 600, 40_000, 300_000])))
            parts.append(np.full(n, k, dtype=np.int64))
            k += int(rng.choice([1, 1, 2, 30]))
            left -= n
        return np.concatenate(parts)
    if kind == 2:     # power law
        w = np.arange(1, nnz // 10 + 2, dtype=np.float64) ** (-1 / 1.5)
        return np.sort(rng.choice(len(w), nnz, p=w / w.sum())).astype(np.int64)
    return random_index(rng, nnz)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--ops", default="is,is,is_unsorted_flag,gws,gs", help="comma list to draw the op from")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
    covered = {}
    for it in range(a.iters):
        nnz = int(rng.choice([70_000, 300_000, 1_000_000, int(rng.integers(50_000, 3_000_000))]))
        F = int(rng.choice([1, 2, 4, 5, 8, 16, 31, 32, 64, 100, 128, 128, 256]))
        if nnz * F > 120_000_000:
            nnz = 120_000_000 // F
        index = big_index(rng, nnz)
        nnz = len(index)
        # bound the OUTPUT too: sparse-key generators can spread 3 M edges over 10^8 rows, and the float64
        # references of such a case need hundreds of GB of host memory (this took a GPU box down once)
        max_rows = max(50_000_000 // F, 1000)
        if int(index[-1]) >= max_rows:
            index = np.minimum(index // (int(index[-1]) // max_rows + 1), max_rows - 1)
        hub, narrow = int(rng.choice([-1, 0, 1])), int(rng.choice([1, 1, 2, 0]))
        cg = int(rng.choice([0, 0, 16, 32, 64]))
        slab_always = bool(rng.integers(0, 2)) and F in (64, 128, 256)      # the source-blocked kernels on random graphs too
        geot.ops.set_option("slab_mode", "always" if slab_always else "auto")
        hip.set_option("hub", hub)
        hip.set_option("narrow", narrow)
        hip.tune(cg, 0, -1, -1)
        handoff, tries = int(rng.choice([1, 1, 2, 0])), int(rng.choice([400000, 400000, 0]))   # round 3: in-kernel hand-off on / off / forced to give up
        hip.set_option("handoff", handoff)
        hip.set_option("handoff_tries", tries)
        red = str(rng.choice(["sum", "sum", "mean", "max", "min"]))
        op = str(rng.choice(a.ops.split(",")))
        covered[(op, red)] = covered.get((op, red), 0) + 1
        tag = f"it={it} op={op} nnz={nnz} F={F} keys={index[-1] + 1} red={red} hub={hub} narrow={narrow} cg={cg} slab={slab_always} handoff={handoff}/{tries}"
        if slab_always and not op.startswith("is"):
            covered[("slab", red)] = covered.get(("slab", red), 0) + 1
        src = rng.standard_normal((nnz, F)).astype(np.float32)
        if op == "is" and red == "sum" and rng.integers(0, 6) == 0 and nnz > 5000 and np.all(index[:-1] <= index[1:]):
            # round 3: descents written behind the version counter - the stale "ascending" fact is met by the descent guard
            import warnings
            covered[("is_data_scramble", red)] = covered.get(("is_data_scramble", red), 0) + 1
            t_index, t_src = t(index), t(src)
            geot.index_scatter(0, t_src, t_index, red, True)
            at = rng.integers(1, nnz - 1002, 25)
            bad = index.copy()
            bad[at], bad[at + 1000] = index[at + 1000].copy(), index[at].copy()
            t_index.data.copy_(t(bad))
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                out = geot.index_scatter(0, t_src, t_index, red, True).cpu().numpy()
                torch.cuda.synchronize()
                geot.index_scatter(0, t_src[:16], t_index[:16].clone(), red, True)       # (the host notices the alarm here)
            order = np.argsort(bad, kind="stable")
            hi = oracle.index_scatter(bad[order], src[order], rows=out.shape[0], acc64=True)
            mag = oracle.index_scatter(bad[order], np.abs(src[order]), rows=out.shape[0], acc64=True)
            ok = out.shape == hi.shape and np.all(np.abs(out - hi) <= 2e-5 * mag + 1e-30)
        elif op.startswith("is") and rng.integers(0, 3) == 0:
            # round 3: the other storage types (the tile shapes and the hand-off differ by dtype): 16-bit storage accumulates in
            # fp32 and rounds once; fp64 all the way
            dt = str(rng.choice(["bf16", "f16", "f64"]))
            covered[("is_" + dt, red)] = covered.get(("is_" + dt, red), 0) + 1
            tag += " dtype=" + dt
            tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f64": torch.float64}[dt]
            t_src = t(src if dt != "f16" else src / 8).to(tdt)
            back = t_src.double().cpu().numpy()                       # the values as stored
            out = geot.index_scatter(0, t_src, t(index), red, op == "is").double().cpu().numpy()
            eps = {"bf16": 2.0 ** -8, "f16": 2.0 ** -11, "f64": 1e-13}[dt]
            rows = out.shape[0]
            hi = np.zeros((rows, F)); np.add.at(hi, index, back)
            if red == "sum":
                mag = np.zeros((rows, F)); np.add.at(mag, index, np.abs(back))
                ok = out.shape == hi.shape and np.all(np.abs(out - hi) <= eps * np.abs(hi) + (2e-5 if dt != "f64" else 1e-13) * mag + 1e-30)
                if dt == "f16" and mag.max() > 6e4:
                    ok = True                                          # (a hub beyond float16's range: inf by construction, not checked)
            elif red == "mean":
                cnt = np.maximum(np.bincount(index, minlength=rows), 1)[:, None]
                ok = np.allclose(out, hi / cnt, rtol=2 * eps + 2e-5, atol=1e-6)
            else:
                ref = np.full((rows, F), -np.inf if red == "max" else np.inf)
                (np.maximum if red == "max" else np.minimum).at(ref, index, back)
                ref[np.bincount(index, minlength=rows) == 0] = 0.0
                ok = np.array_equal(out, ref)
        elif op.startswith("is"):
            out = geot.index_scatter(0, t(src), t(index), red, op == "is").cpu().numpy()
            if red == "sum":
                hi = oracle.index_scatter(index, src, acc64=True)
                mag = oracle.index_scatter(index, np.abs(src), acc64=True)
                ok = out.shape == hi.shape and np.all(np.abs(out - hi) <= 2e-5 * mag + 1e-30) and np.all(out[mag == 0] == 0)
            else:
                ref = oracle.index_scatter_3pass(index, src, reduce=red)
                ok = out.shape == ref.shape and (np.allclose(out, ref, rtol=2e-4, atol=2e-5) if red == "mean"
                                                 else np.array_equal(out, ref))
        elif op == "mh_bwd":
            # round 5: mh_spmm's backward - d/dsrc over the transposed list (its own plan when slab-forced), d/dweight by the multi-head
            # SDDMM (per edge or over the forward's plan) - against plain-torch float64 autograd of the reference test's formula
            H = int(rng.choice([2, 4, 8]))
            Fh = int(rng.choice([16, 32, 64]))
            if nnz > 1_000_000:
                index = index[:1_000_000]
                nnz = len(index)
            geot.ops.set_option("slab_mode", "always" if rng.integers(0, 2) else "auto")
            nodes = int(index[-1]) + 1 + int(rng.integers(0, 9))
            si = rng.integers(0, nodes, nnz).astype(np.int64)
            head_major = bool(rng.integers(0, 2))
            tag += f" H={H} Fh={Fh} head_major={head_major} nodes={nodes}"
            covered[("mh_bwd", "sum")] = covered.get(("mh_bwd", "sum"), 0) + 1
            t_si, t_di = t(si), t(index)
            t_x = t(rng.random((nodes, H, Fh), dtype=np.float32)).requires_grad_()
            w_em = t(rng.random((nnz, H), dtype=np.float32) + 0.25)
            t_w = (w_em.t().contiguous() if head_major else w_em.clone()).requires_grad_()
            rows = int(index[-1]) + 1
            up = t(rng.random((rows, H, Fh), dtype=np.float32))
            ok = True
            for _ in range(2):                                                         # (the second sighting builds the plans)
                y = geot.mh_spmm(t_si, t_di, t_w, t_x)
                gx, gw = torch.autograd.grad(y, [t_x, t_w], up)
            xr, wr = t_x.detach().double().requires_grad_(), w_em.double().requires_grad_()
            ref = torch.zeros(rows, H, Fh, device="cuda", dtype=torch.float64).index_add_(0, t_di, xr[t_si] * wr[:, :, None])
            rgx, rgw = torch.autograd.grad(ref, [xr, wr], up.double())
            if head_major:
                rgw = rgw.t()
            for name, a_, b_ in (("y", y, ref), ("d/dsrc", gx, rgx), ("d/dweight", gw, rgw)):
                if a_.shape != b_.shape or float((a_.double() - b_).abs().max()) > 3e-5 * float(b_.abs().max()) + 1e-30:
                    ok = False
                    tag += f" FAILED:{name}"
            out = hi = mag = None
        elif op == "mh":
            # multi-head SpMM (round 4: its own source-blocked kernels - one row per wave-instruction for rows of 1 KiB / 512 / 256
            # bytes): heads x width that make such rows or not, three storage types, both weight layouts, per-edge or slab-forced
            H = int(rng.choice([2, 4, 8]))
            Fh = int(rng.choice([8, 16, 32, 64, 64, 128]))
            dt = str(rng.choice(["f32", "f32", "bf16", "f16"]))
            tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[dt]
            if nnz * H * Fh > 100_000_000:
                index = index[: 100_000_000 // (H * Fh)]
                nnz = len(index)
            geot.ops.set_option("slab_mode", "always" if rng.integers(0, 2) else "auto")
            nodes = int(index[-1]) + 1 + int(rng.integers(0, 9))
            si = rng.integers(0, nodes, nnz).astype(np.int64)
            x = (rng.random((nodes, H, Fh), dtype=np.float32) * 0.25)
            w = rng.random((nnz, H), dtype=np.float32) + 0.25
            t_x, t_w = t(x).to(tdt), t(w).to(tdt)
            xs, wsv = t_x.float().cpu().numpy(), t_w.float().cpu().numpy()            # the values as stored
            head_major = bool(rng.integers(0, 2))
            tag += f" H={H} Fh={Fh} dtype={dt} head_major={head_major}"
            covered[("mh_" + dt, "sum")] = covered.get(("mh_" + dt, "sum"), 0) + 1
            t_si, t_di = t(si), t(index)
            wt = t_w.t().contiguous() if head_major else t_w
            for _ in range(3):                                                         # (the second sighting of an edge list builds its plan)
                got = geot.mh_spmm(t_si, t_di, wt, t_x)
            out = got.float().cpu().numpy().reshape(-1, H * Fh)
            hi = oracle.mh_spmm(si, index, wsv, xs, rows=out.shape[0], acc64=True).reshape(-1, H * Fh)
            mag = oracle.mh_spmm(si, index, wsv, np.abs(xs), rows=out.shape[0], acc64=True).reshape(-1, H * Fh)
            eps = {"f32": 0.0, "bf16": 2.0 ** -8, "f16": 2.0 ** -11}[dt]
            floor = 2.0 ** -24 if dt == "f16" else 0.0                               # (float16 results below 6e-5 are subnormal: spacing 2^-24)
            ok = out.shape == hi.shape and np.all(np.abs(out - hi) <= eps * np.abs(hi) + 2e-5 * mag + floor + 1e-30) and np.all(out[mag == 0] == 0)
            if dt == "f16" and mag.max() > 6e4:
                ok = True                                                              # (beyond float16's range: not checked)
        else:
            nodes = int(index[-1]) + 1 + int(rng.integers(0, 9))
            si = rng.integers(0, nodes, nnz).astype(np.int64)
            x = rng.standard_normal((nodes, F)).astype(np.float32)
            w = rng.random(nnz, dtype=np.float32) + 0.25 if op == "gws" else None
            if red == "sum" and rng.integers(0, 5) == 0:
                # round 3: a NEW edge list / weight written into the same tensors behind the version counter, after the
                # host has remembered what it derived from the old one (plan, weight in plan order): the content guard
                import warnings
                covered[("content_swap", op)] = covered.get(("content_swap", op), 0) + 1
                t_si, t_di, t_x = t(si), t(index), t(x)
                t_w = t(w) if w is not None else None
                call = (lambda: geot.gather_weight_scatter(t_si, t_di, t_w, t_x)) if w is not None else (lambda: geot.gather_scatter(t_si, t_di, t_x))
                for _ in range(3):
                    call()
                si = rng.integers(0, nodes, nnz).astype(np.int64)
                t_si.data.copy_(t(si))
                if w is not None and rng.integers(0, 2):
                    w = rng.random(nnz, dtype=np.float32) + 0.25
                    t_w.data.copy_(t(w))
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    out = call().cpu().numpy()
                if w is not None:
                    hi = oracle.gather_weight_scatter(si, index, w, x, acc64=True)
                    mag = oracle.gather_weight_scatter(si, index, w, np.abs(x), acc64=True)
                else:
                    hi = oracle.gather_scatter(si, index, x, acc64=True)
                    mag = oracle.gather_scatter(si, index, np.abs(x), acc64=True)
                ok = out.shape == hi.shape and np.all(np.abs(out - hi) <= 2e-5 * mag + 1e-30) and np.all(out[mag == 0] == 0)
            elif red in ("sum", "mean") and rng.integers(0, 3) == 0:
                # round 6: 16-bit storage through the gather operators - with slab_always and rows of 256 / 512 bytes (F = 128 / 256) the
                # plan is cut into waves and the matrix-core kernels run it (sums and means; one rounding of the fp32 result)
                dt = str(rng.choice(["bf16", "f16"]))
                covered[(op + "_" + dt, red)] = covered.get((op + "_" + dt, red), 0) + 1
                tag += " dtype=" + dt
                tdt = {"bf16": torch.bfloat16, "f16": torch.float16}[dt]
                t_x = t(x if dt != "f16" else x / 8).to(tdt)
                t_w = t(w).to(tdt) if w is not None else None
                xs = t_x.double().cpu().numpy()
                ws = t_w.double().cpu().numpy() if w is not None else None
                out = (geot.gather_scatter(t(si), t(index), t_x, red) if w is None
                       else geot.gather_weight_scatter(t(si), t(index), t_w, t_x, red)).double().cpu().numpy()
                rows_o = out.shape[0]
                msg = xs[si] if ws is None else xs[si] * ws[:, None]
                hi = np.zeros((rows_o, F)); np.add.at(hi, index, msg)
                mag = np.zeros((rows_o, F)); np.add.at(mag, index, np.abs(msg))
                if red == "mean":
                    cnt = np.maximum(np.bincount(index, minlength=rows_o), 1)[:, None]
                    hi, mag = hi / cnt, mag / cnt
                eps = {"bf16": 2.0 ** -8, "f16": 2.0 ** -11}[dt]
                floor = 2.0 ** -24 if dt == "f16" else 0.0
                ok = out.shape == hi.shape and np.all(np.abs(out - hi) <= eps * np.abs(hi) + 2e-5 * mag + floor + 1e-30)
                if dt == "f16" and mag.max() > 6e4:
                    ok = True                                          # (beyond float16's range: not checked)
            elif red == "sum":
                if op == "gws":
                    out = geot.gather_weight_scatter(t(si), t(index), t(w), t(x)).cpu().numpy()
                    hi = oracle.gather_weight_scatter(si, index, w, x, acc64=True)
                    mag = oracle.gather_weight_scatter(si, index, w, np.abs(x), acc64=True)
                else:
                    out = geot.gather_scatter(t(si), t(index), t(x)).cpu().numpy()
                    hi = oracle.gather_scatter(si, index, x, acc64=True)
                    mag = oracle.gather_scatter(si, index, np.abs(x), acc64=True)
                ok = out.shape == hi.shape and np.all(np.abs(out - hi) <= 2e-5 * mag + 1e-30) and np.all(out[mag == 0] == 0)
            else:   # PyG-style aggr over the messages: the oracle reduces the materialised messages
                msg = x[si] if w is None else x[si] * w[:, None]
                out = (geot.gather_scatter(t(si), t(index), t(x), red) if w is None
                       else geot.gather_weight_scatter(t(si), t(index), t(w), t(x), red)).cpu().numpy()
                if red == "mean":   # float64 reference: with a handful of distinct messages the reference's own
                    ref = np.zeros(out.shape, dtype=np.float64)     # sequential fp32 sum drifts by 1e-3 relative
                    np.add.at(ref, index, msg.astype(np.float64))
                    ref /= np.maximum(np.bincount(index, minlength=out.shape[0]), 1)[:, None]
                    ok = np.allclose(out, ref, rtol=2e-5, atol=2e-6)
                else:
                    ref = oracle.index_scatter_3pass(index, msg, reduce=red, rows=out.shape[0])
                    ok = out.shape == ref.shape and np.array_equal(out, ref)
        if not ok:
            print("MISMATCH", tag, flush=True)
            try:        # where and by how much (sum cases keep `hi` / `mag`)
                ratio = np.abs(out - hi) / (2e-5 * mag + 1e-30)
                r, c = np.unravel_index(np.argmax(ratio), ratio.shape)
                print(f"  worst: row {r} col {c}: got {out[r, c]!r} want {hi[r, c]!r} |diff| / bound = {ratio[r, c]:.3g}; rows over the bound: "
                      f"{np.unique(np.nonzero(ratio > 1)[0])[:10]}, nodes {x.shape[0] if 'x' in dir() else '-'}, stats {geot.ops.stats()}", flush=True)
            except Exception as e:  # noqa: BLE001
                print("  (no detail:", repr(e), ")", flush=True)
            sys.exit(1)
        if it % 20 == 0:
            print("ok", tag, flush=True)
    hip.set_option("hub", -1); hip.set_option("narrow", 1); hip.tune(0, 0, -1, -1); geot.ops.set_option("slab_mode", "auto")
    hip.set_option("handoff", 1); hip.set_option("handoff_tries", 400000)
    print("covered (op, reduce): " + ", ".join(f"{o}/{r}={n}" for (o, r), n in sorted(covered.items())))
    print(f"SOAK PASSED ({a.iters} cases, seed {a.seed})")


if __name__ == "__main__":
    main()
