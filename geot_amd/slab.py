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
from typing import Optional

import torch

from . import _lib, hip



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

    def with_e_perm(self, e_perm: torch.Tensor) -> "SlabPlan":
        """The same plan reading its per-edge weights through ANOTHER index (int32 [nnz]: plan position -> position in the weight array):
        every other array is shared.  geot_amd.Graph uses it to hand the backward's kernels weights that live in the forward's order."""
        if e_perm.dtype != torch.int32 or e_perm.numel() != self.tensors["e_perm"].numel() or not e_perm.is_contiguous():
            raise ValueError("with_e_perm: a contiguous int32 tensor with one entry per edge")
        scalars = {name: getattr(self.struct, name) for name in _SCALARS + ("slab_shift", "n_slabs")}
        twin = SlabPlan({**self.tensors, "e_perm": e_perm}, scalars, dict(self.meta))
        twin.base = getattr(self, "base", self)
        return twin


def worthwhile(nnz: int, out_rows: int, src_rows: int, rowbytes: int) -> bool:
    """Is the graph dense enough for L2 re-use?  Per round the chip holds R x units output rows; each XCD then reads
    (edges of the round / 8) rows out of a table of src_rows: below ~4 uses per row and round the sweep through L2
    costs more than it saves (configs[2], ogbn-products scale, has 0.1; configs[3], Reddit scale, 7).  The rule lives in
    the host layer (csrc/host_plan.cpp slab_worthwhile)."""
    return bool(torch.ops.geot._slab_worthwhile(nnz, out_rows, src_rows, rowbytes))


_FIELDS = ("e_src", "e_dl", "e_perm", "g_begin", "g_vrow0", "g_nv", "v_out", "c_row", "c_first", "c_count", "v_row", "v_total", "c_total")
_SCALARS = ("n_groups", "n_vrows", "n_carry", "n_split", "nnz", "units", "rows_per_group")


def build_plan(src_index: torch.Tensor, dst_index: torch.Tensor, out_rows: int, src_rows: int, rowbytes: int,
               weight_mode: int, heads: int = 1, slab_bytes: int = 0,
               rows_per_group: Optional[int] = None, units: Optional[int] = None) -> SlabPlan:
    """dst_index ascending (the caller has checked), int64 COO.  Built by the host layer's planner (the same code the
    operators use on the second call with an edge list): a few scans and one stable sort on the tensors' device plus
    one host loop over the virtual rows; ~15 ms at 115 M edges in a warm process.  Works on CPU tensors too (the tests run a numpy
    emulation of the kernel over the arrays).  `units` / `rows_per_group` / `slab_bytes`: 0 = the library's values."""
    from . import ops  # noqa: F401  (loads the plugin)
    res = torch.ops.geot._slab_plan(src_index, dst_index, int(out_rows), int(src_rows), int(rowbytes), int(weight_mode),
                                    int(heads), int(slab_bytes), int(rows_per_group or 0), int(units or 0))
    sc = res[-1].tolist()
    tensors = dict(zip(_FIELDS, res[:-1]))
    scalars = dict(zip(_SCALARS, sc[:7]))
    scalars["slab_shift"], scalars["n_slabs"] = sc[12], sc[10]
    meta = {"rounds": sc[7], "budget": sc[8], "cap": sc[9], "slabs": sc[10], "slab_rows": sc[11], "groups": sc[0], "vrows": sc[1],
            "split_rows": sc[3], "units": sc[5], "rows_per_group": sc[6], "weight_mode": weight_mode, "heads": heads,
            "rowbytes": rowbytes}
    return SlabPlan(tensors, scalars, meta)


def slab_spmm_out(plan: SlabPlan, weight: Optional[torch.Tensor], weight_mode: int, src: torch.Tensor, out: torch.Tensor,
                  heads: int, feat: int, reduce: str = "sum", stage_weights: bool = True) -> torch.Tensor:
    """out[d, h, :] = reduce_e w(e, h) * src[s[e], h, :] over the plan's edges (pointer-level doorway);
    reduce: 'sum' | 'mean' | 'max' | 'min' (weight modes 0 / 1)."""
    tensors = [src, out] + ([weight] if weight is not None else [])
    dev = hip._require_gpu(*tensors)
    if src.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        raise RuntimeError("slab_spmm: float32, float16 or bfloat16")
    if weight is not None and weight.dtype != src.dtype:
        raise RuntimeError("slab_spmm: weight must have the dtype of src")
    L = _lib.load()
    with hip._on_device(dev):
        st = hip._stream_handle(dev)
        # (with room for the weights in plan order: edge-order weights are then staged inside the kernel, `stage_weights=False`: read through e_perm)
        nbytes = int(L.geot_slab_workspace_bytes_staged(ctypes.byref(plan.struct), heads * feat, weight_mode, heads, hip._dtype_code(src, "slab_spmm"))
                     if stage_weights else L.geot_slab_workspace_bytes(ctypes.byref(plan.struct), heads * feat))
        ws = hip.workspace(dev, nbytes, st)
        if not stage_weights:
            ws = ws[:nbytes]
        rc = L.geot_slab_spmm(ctypes.byref(plan.struct), None if weight is None else weight.data_ptr(), weight_mode,
                              src.data_ptr(), out.data_ptr(), heads, feat, src.shape[0], out.shape[0], hip._dtype_code(src, "slab_spmm"),
                              hip._REDUCE_CODES[reduce], ws.data_ptr(), ws.numel(), st)
    _lib.check(rc, "geot_slab_spmm")
    return out


def rows_per_group(weight_mode: int, heads: int, dtype: torch.dtype, rowbytes: int = 0) -> int:
    """R of a plan for this storage type (16-bit storage keeps fp32 accumulators in LDS: half the rows per group) and - given
    `rowbytes` - for the kernel that will run it (multi-head plans over rows of 512 / 256 bytes: one row per wave-instruction, more
    rows per group; include/geot_hip.h geot_slab_rows_per_group_shape)."""
    code = {torch.float32: _lib.GEOT_F32, torch.float16: _lib.GEOT_F16, torch.bfloat16: _lib.GEOT_BF16}[dtype]
    if rowbytes:
        return int(_lib.load().geot_slab_rows_per_group_shape(weight_mode, heads, code, rowbytes))
    return int(_lib.load().geot_slab_rows_per_group_dtype(weight_mode, heads, code))


def slab_sddmm_out(plan: SlabPlan, mat_1: torch.Tensor, mat_2: torch.Tensor, out: torch.Tensor, staged: bool = True) -> torch.Tensor:
    """out[e] = <mat_1[dst(e)], mat_2[src(e)]> in original edge order over the plan's edges (pointer-level doorway).
    staged (default): results leave the persistent kernel in plan order and a second kernel reorders them group by group through LDS
    (geot_slab_sddmm_staged);
    False: written straight to out[original edge id] (every result a partial write of its own: ~1.5x slower at F=128)."""
    dev = hip._require_gpu(mat_1, mat_2, out)
    L = _lib.load()
    with hip._on_device(dev):
        st = hip._stream_handle(dev)
        ws = hip.workspace(dev, int(L.geot_slab_workspace_bytes(ctypes.byref(plan.struct), mat_1.shape[1])), st)
        if staged:
            staging = torch.empty(out.numel(), dtype=out.dtype, device=dev)
            rc = L.geot_slab_sddmm_staged(ctypes.byref(plan.struct), mat_1.data_ptr(), mat_2.data_ptr(), out.data_ptr(), staging.data_ptr(),
                                          mat_1.shape[1], mat_1.shape[0], mat_2.shape[0], hip._dtype_code(mat_1, "slab_sddmm"),
                                          ws.data_ptr(), ws.numel(), st)
        else:
            rc = L.geot_slab_sddmm(ctypes.byref(plan.struct), mat_1.data_ptr(), mat_2.data_ptr(), out.data_ptr(), mat_1.shape[1],
                                   mat_1.shape[0], mat_2.shape[0], hip._dtype_code(mat_1, "slab_sddmm"), ws.data_ptr(), ws.numel(), st)
    _lib.check(rc, "geot_slab_sddmm")
    return out


def slab_mh_sddmm_out(plan: SlabPlan, mat_1: torch.Tensor, mat_2: torch.Tensor, out: Optional[torch.Tensor],
                      staging: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[e, h] = <mat_1[dst(e), h], mat_2[src(e), h]> over the plan's edges, mat_* [rows, H, F] (geot_slab_mh_sddmm).
    out given: [nnz, H] in ORIGINAL edge order (through `staging` and the unstage kernel).  out None: the results stay in plan
    order - returned as the [nnz, H] staging tensor, what slab_spmm_out reads back under weight_mode 5 (H = 1: mode 4)."""
    dev = hip._require_gpu(mat_1, mat_2)
    L = _lib.load()
    H, F = mat_1.shape[1], mat_1.shape[2]
    nnz = int(plan.struct.nnz)
    if staging is None:
        staging = torch.empty((nnz, H), dtype=mat_1.dtype, device=dev)
    with hip._on_device(dev):
        st = hip._stream_handle(dev)
        ws = hip.workspace(dev, int(L.geot_slab_workspace_bytes(ctypes.byref(plan.struct), H * F)), st)
        rc = L.geot_slab_mh_sddmm(ctypes.byref(plan.struct), mat_1.data_ptr(), mat_2.data_ptr(), None if out is None else out.data_ptr(),
                                  staging.data_ptr(), H, F, mat_1.shape[0], mat_2.shape[0], hip._dtype_code(mat_1, "slab_mh_sddmm"),
                                  ws.data_ptr(), ws.numel(), st)
    _lib.check(rc, "geot_slab_mh_sddmm")
    return staging if out is None else out
