"""Phase A of the source-blocked ("slab") gather kernels (geot_amd/csrc/seg_slab.hip, geot_slab_spmm).

For a DENSE graph - hundreds of edges per destination row, a source table of a few hundred MB: Reddit, BASELINE.json
configs[3] - the per-edge row gather of the dst-sorted kernels re-reads every source row hundreds of times from
the Infinity Cache.  The slab kernel serves those re-reads from the XCDs' L2 instead by making the whole chip walk
the source table in step; that needs the edge list arranged once per graph:

    groups   consecutive (virtual) dst rows, <= R rows and about B edges each; a row with more than CAP edges is
             split into virtual rows (their partial sums are combined afterwards, in order);
    order    groups by size (descending): the groups of one round are equally long;
             edges inside a group by (source slab, row in group, original position).

Everything here is a pure function of (src_index, dst_index, shapes): it is built once per edge list and kept (the
host layer keys it on the tensors' content identity, like the transposed edge list of the backward pass).  The
reference re-derives nothing comparable: its kernels gather per edge (csrc/cuda/mh_spmm_kernel.cuh:28-111).
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional

import numpy as np
import torch

from . import _lib, hip

SLAB_BYTES = 1 << 20          # measured on MI355X (profiles/r02/kexp2_slab_cfg4_table.txt): 0.5-2 MiB slabs serve
                              # row gathers at 24-29 TB/s, 3 MiB at 19 TB/s (L2 is 4 MiB per XCD)


class SlabPlan:
    """Device arrays of a geot_slab_plan + the ctypes struct that points at them."""

    def __init__(self, tensors: dict, scalars: dict, meta: dict):
        self.tensors, self.meta = tensors, meta
        s = _lib.SlabPlan()
        for name, t in tensors.items():
            setattr(s, name, t.data_ptr())
        for name, v in scalars.items():
            setattr(s, name, int(v))
        self.struct = s

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.tensors.values())


def worthwhile(nnz: int, out_rows: int, src_rows: int, rowbytes: int) -> bool:
    """Is the graph dense enough for L2 re-use?  Per round the chip holds R x units output rows; each XCD then reads
    (edges of the round / 8) rows out of a table of src_rows: below ~4 uses per row and round the sweep through L2
    costs more than it saves (configs[2], ogbn-products scale, has 0.1; configs[3], Reddit scale, 7)."""
    if rowbytes not in (256, 512, 1024) or nnz >= 2 ** 31 or nnz < 8_000_000 or out_rows < 1 or src_rows >= 2 ** 31:
        return False
    L = _lib.load()
    units = L.geot_slab_units() * (1024 // rowbytes)
    rounds = max(1, math.ceil(out_rows / (15 * units)))
    return nnz / rounds / 8 / max(src_rows, 1) >= 4.0


def build_plan(src_index: torch.Tensor, dst_index: torch.Tensor, out_rows: int, src_rows: int, rowbytes: int,
               weight_mode: int, heads: int = 1, slab_bytes: int = SLAB_BYTES,
               rows_per_group: Optional[int] = None, units: Optional[int] = None) -> SlabPlan:
    """dst_index ascending (the caller has checked), int64 COO on the GPU.  A few sorts / scans on the device plus
    one short host loop over the groups; ~0.1 s at 115 M edges.  Synchronises (Phase A is not on the step path)."""
    L = _lib.load()
    dev = dst_index.device
    nnz = int(dst_index.numel())
    lanes = rowbytes // 16
    units = int(units or L.geot_slab_units() * (64 // lanes))      # (`units` override: CPU emulation in the tests)
    R = int(rows_per_group or L.geot_slab_rows_per_group(weight_mode, heads))
    counts = torch.bincount(dst_index, minlength=out_rows)[:out_rows]
    rowptr = torch.cumsum(counts, 0) - counts
    nonempty = int((counts > 0).sum())
    rounds0 = max(1, math.ceil(nonempty / (R * units)))
    budget = max(256, math.ceil(nnz / (rounds0 * units)))      # edges per group
    cap = max(64, budget // 2)                                 # edges per virtual row (hub pieces)
    nv_row = (counts + (cap - 1)) // cap                       # virtual rows per dst row (0 for an empty row)
    vstart = torch.cumsum(nv_row, 0) - nv_row
    V = int(nv_row.sum())
    v_row = torch.repeat_interleave(torch.arange(out_rows, device=dev), nv_row, output_size=V)
    v_piece = torch.arange(V, device=dev) - vstart[v_row]
    v_cnt = torch.minimum(counts[v_row] - v_piece * cap, torch.full((), cap, device=dev, dtype=torch.int64))

    # ---- groups: greedy over consecutive virtual rows, <= R rows and <= budget edges (short host loop over groups)
    cum = np.concatenate([[0], np.cumsum(v_cnt.cpu().numpy())])
    idx = np.arange(V)
    nxt = np.minimum(idx + R, np.searchsorted(cum, cum[:-1] + budget, side="right") - 1)
    nxt = np.maximum(nxt, idx + 1).tolist()
    starts = []
    i = 0
    while i < V:
        starts.append(i)
        i = nxt[i]
    starts = np.asarray(starts, dtype=np.int64)
    G = len(starts)
    ends = np.append(starts[1:], V)
    g_edges = cum[ends] - cum[starts]
    order = np.argsort(-g_edges, kind="stable")                # position p -> group id
    pos_of_group = np.empty(G, dtype=np.int64)
    pos_of_group[order] = np.arange(G)
    g_begin = np.concatenate([[0], np.cumsum(g_edges[order])]).astype(np.int64)

    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)  # noqa: E731
    group_of_vrow = torch.repeat_interleave(torch.arange(G, device=dev), t(ends - starts, torch.int64), output_size=V)
    start_of_group = t(starts, torch.int64)
    pos_t = t(pos_of_group, torch.int64)

    # ---- per edge: virtual row, group position, slab, row in group -> one stable sort ------------------------------
    e = torch.arange(nnz, device=dev)
    vrow_e = vstart[dst_index] + (e - rowptr[dst_index]) // cap
    del e
    gid_e = group_of_vrow[vrow_e]
    dl_e = (vrow_e - start_of_group[gid_e])
    slab_rows = max(1, slab_bytes // rowbytes)
    n_slabs = (src_rows + slab_rows - 1) // slab_rows
    key = (pos_t[gid_e] * n_slabs + torch.div(src_index, slab_rows, rounding_mode="floor").clamp_(0, n_slabs - 1)) * R + dl_e
    del gid_e, vrow_e
    _, perm = torch.sort(key, stable=True)
    del key
    e_src = src_index[perm].to(torch.int32)
    e_dl = dl_e[perm].to(torch.uint8)
    e_perm = perm.to(torch.int32)
    del perm, dl_e

    # ---- outputs of the virtual rows: dst row, or a carry slot for the pieces of a split row ------------------------
    split_v = nv_row[v_row] > 1
    carry_slot = torch.cumsum(split_v.to(torch.int64), 0) - 1
    v_out = torch.where(split_v, -(carry_slot + 1), v_row)
    split_rows = torch.nonzero(nv_row > 1).flatten()
    c_count = nv_row[split_rows].to(torch.int32)
    c_first = carry_slot[vstart[split_rows]] if split_rows.numel() else split_rows
    n_carry = int(split_v.sum())

    tensors = {
        "e_src": e_src, "e_dl": e_dl, "e_perm": e_perm,
        "g_begin": t(g_begin, torch.int64), "g_vrow0": t(starts[order], torch.int32),
        "g_nv": t((ends - starts)[order], torch.int32), "v_out": v_out.contiguous(),
        "c_row": split_rows.contiguous(), "c_first": c_first.contiguous(), "c_count": c_count.contiguous(),
    }
    for k, v in tensors.items():                                # a NULL pointer for an empty array is fine for the kernels
        if v.numel() == 0:
            tensors[k] = torch.zeros(1, dtype=v.dtype, device=dev)
    scalars = {"n_groups": G, "n_vrows": V, "n_carry": n_carry, "n_split": int(split_rows.numel()), "nnz": nnz,
               "units": units, "rows_per_group": R}
    meta = {"rounds": math.ceil(G / units), "budget": budget, "cap": cap, "groups": G, "vrows": V, "split_rows": int(split_rows.numel()),
            "slabs": n_slabs, "slab_rows": slab_rows, "units": units, "rows_per_group": R, "weight_mode": weight_mode, "heads": heads,
            "rowbytes": rowbytes}
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    return SlabPlan(tensors, scalars, meta)


def slab_spmm_out(plan: SlabPlan, weight: Optional[torch.Tensor], weight_mode: int, src: torch.Tensor, out: torch.Tensor,
                  heads: int, feat: int) -> torch.Tensor:
    """out[d, h, :] = sum_e w(e, h) * src[s[e], h, :] over the plan's edges (pointer-level doorway)."""
    tensors = [src, out] + ([weight] if weight is not None else [])
    dev = hip._require_gpu(*tensors)
    if src.dtype != torch.float32:
        raise RuntimeError("slab_spmm: float32 only")
    L = _lib.load()
    with hip._on_device(dev):
        st = hip._stream_handle(dev)
        nbytes = int(L.geot_slab_workspace_bytes(ctypes.byref(plan.struct), heads * feat))
        ws = hip.workspace(dev, nbytes, st)
        rc = L.geot_slab_spmm(ctypes.byref(plan.struct), None if weight is None else weight.data_ptr(), weight_mode,
                              src.data_ptr(), out.data_ptr(), heads, feat, src.shape[0], out.shape[0], _lib.GEOT_F32,
                              ws.data_ptr(), ws.numel(), st)
    _lib.check(rc, "geot_slab_spmm")
    return out
