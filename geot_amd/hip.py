"""Tensor-level doorway to the C ABI (include/geot_hip.h): raw pointers, sizes, current stream.

These are the `*_cuda(...)` entry points of the reference (csrc/cuda/header_cuda.h:4-30) restated
for the HIP library: the caller supplies the output tensor and its row count, nothing here
synchronises with the host, allocates device memory per call (beyond the cached workspace) or
touches the CPU.  The operator layer (geot_amd/csrc/torch_ops.cpp + geot_amd/ops.py) adds the reference's shape rule, checks
and error texts on top.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Tuple

import torch

from . import _lib

_DT = {torch.float32: _lib.GEOT_F32, torch.float64: _lib.GEOT_F64,
       torch.float16: _lib.GEOT_F16, torch.bfloat16: _lib.GEOT_BF16}  # 16-bit: fp32 accumulation

_REDUCE_CODES = {"max": 0, "mean": 1, "min": 2, "sum": 3, "prod": 4}  # csrc/reducetype.h:3

# one workspace per (device, stream): the C ABI allows reuse by one stream at a time
_workspaces: Dict[Tuple[int, int], torch.Tensor] = {}


def _require_gpu(*tensors: torch.Tensor) -> torch.device:
    dev = tensors[0].device
    for t in tensors:
        if t.device.type != "cuda":
            raise RuntimeError(
                "geot_amd runs on MI355X only: got a tensor on device "
                f"'{t.device}'.  There is no CPU fallback in this package.")
        if t.device != dev:
            raise RuntimeError("all tensors must be on the same device")
    return dev


_SCALAR_NAMES = {torch.int32: "Int", torch.int16: "Short", torch.int8: "Char", torch.uint8: "Byte",
                 torch.int64: "Long", torch.float32: "Float", torch.float64: "Double",
                 torch.float16: "Half", torch.bfloat16: "BFloat16", torch.bool: "Bool"}


def _dtype_code(t: torch.Tensor, name: str) -> int:
    if t.dtype not in _DT:
        # the reference dispatches AT_DISPATCH_FLOATING_TYPES (csrc/cuda/index_scatter_cuda.cu:68)
        raise RuntimeError(f'"{name}" not implemented for \'{_SCALAR_NAMES.get(t.dtype, str(t.dtype))}\'')
    return _DT[t.dtype]


def _index_ptr(t: torch.Tensor, name: str) -> int:
    if t.dtype != torch.int64:
        # reference: index.data_ptr<int64_t>() (csrc/cuda/index_scatter_cuda.cu:11) -> c10 error text
        raise RuntimeError(f"expected scalar type Long but found {_SCALAR_NAMES.get(t.dtype, str(t.dtype))} ({name})")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return t.data_ptr()


def _stream_handle(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


class _on_device:
    """Device guard that costs nothing when `dev` is already current (the common case)."""
    __slots__ = ("dev", "prev")

    def __init__(self, dev: torch.device):
        self.dev = dev.index if dev.index is not None else torch.cuda.current_device()
        self.prev = None

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.dev:
            self.prev = cur
            torch.cuda.set_device(self.dev)
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)
        return False


def workspace(dev: torch.device, nbytes: int, stream: Optional[int] = None) -> torch.Tensor:
    """Cached scratch for the current stream; zero-initialised once (control words), reused after.  Under graph capture a scratch that
    is not there yet is made for this call only and not kept (memory of a graph's private pool must not outlive the graph in a cache)."""
    stream = _stream_handle(dev) if stream is None else stream
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.zeros(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        if not torch.cuda.is_current_stream_capturing():
            _workspaces[key] = ws
    return ws


def release_workspaces() -> None:
    _workspaces.clear()
    _ws_bytes.clear()


_ws_bytes: Dict[tuple, int] = {}


def _ws_args(L, dev, nnz, feat, rows, dt, stream: Optional[int] = None):
    key = (nnz, feat, rows, dt)
    nbytes = _ws_bytes.get(key)
    if nbytes is None:
        if len(_ws_bytes) > 256:
            _ws_bytes.clear()
        nbytes = _ws_bytes[key] = int(L.geot_workspace_bytes(nnz, feat, rows, dt))
    ws = workspace(dev, nbytes, stream)
    return ws, ws.data_ptr(), ws.numel()


def index_scatter_out(index: torch.Tensor, src: torch.Tensor, out: torch.Tensor, sorted: bool = True,
                      reduce: str = "sum") -> torch.Tensor:
    """out[index[e], :] (+)= src[e, :] over a contiguous [nnz, F] view; `out` is [rows, F], written in full.
    reduce: 'sum' | 'mean' | 'min' | 'max' | 'prod' (non-sum: sorted index only)."""
    dev = _require_gpu(index, src, out)
    if reduce != "sum":
        if not sorted:
            raise NotImplementedError("index_scatter: unsorted index supports reduce='sum' only")
        L = _lib.load()
        dt = _dtype_code(src, "index_scatter_sorted")
        nnz = index.numel()
        feat = src.numel() // nnz if nnz else (out.numel() // max(out.shape[0], 1))
        rows = out.shape[0]
        with _on_device(dev):
            ws, wsp, wsn = _ws_args(L, dev, nnz, feat, rows, dt)
            rc = L.geot_index_scatter_reduce(_index_ptr(index, "index"), src.data_ptr(), out.data_ptr(), nnz, feat,
                                             rows, dt, _REDUCE_CODES[reduce], wsp, wsn, _stream_handle(dev))
        _lib.check(rc, "geot_index_scatter_reduce")
        return out
    L = _lib.load()
    dt = _dtype_code(src, "index_scatter_sorted" if sorted else "index_scatter_unsorted")
    nnz = index.numel()
    feat = src.numel() // nnz if nnz else (out.numel() // max(out.shape[0], 1))
    rows = out.shape[0]
    with _on_device(dev):
        st = _stream_handle(dev)
        ws, wsp, wsn = _ws_args(L, dev, nnz, feat, rows, dt, st)
        rc = L.geot_index_scatter(_index_ptr(index, "index"), src.data_ptr(), out.data_ptr(), nnz, feat, rows,
                                  dt, 1 if sorted else 0, wsp, wsn, st)
    _lib.check(rc, "geot_index_scatter")
    return out


def gather_scatter_out(src_index, dst_index, src, out) -> torch.Tensor:
    dev = _require_gpu(src_index, dst_index, src, out)
    L = _lib.load()
    dt = _dtype_code(src, "gather_scatter_sorted")
    nnz, feat, rows = dst_index.numel(), src.shape[1], out.shape[0]
    with _on_device(dev):
        ws, wsp, wsn = _ws_args(L, dev, nnz, feat, rows, dt)
        rc = L.geot_gather_scatter(_index_ptr(src_index, "src_index"), _index_ptr(dst_index, "dst_index"),
                                   src.data_ptr(), out.data_ptr(), nnz, feat, src.shape[0], rows, dt,
                                   wsp, wsn, _stream_handle(dev))
    _lib.check(rc, "geot_gather_scatter")
    return out


def gather_reduce_out(src_index, dst_index, weight, src, out, reduce: str) -> torch.Tensor:
    """out[d] = reduce over the messages (weight[e] *) src[src_index[e]] of row d; weight may be None."""
    tensors = [src_index, dst_index, src, out] + ([weight] if weight is not None else [])
    dev = _require_gpu(*tensors)
    L = _lib.load()
    dt = _dtype_code(src, "gather_scatter_sorted")
    nnz, feat, rows = dst_index.numel(), src.shape[1], out.shape[0]
    with _on_device(dev):
        ws, wsp, wsn = _ws_args(L, dev, nnz, feat, rows, dt)
        rc = L.geot_gather_reduce(_index_ptr(src_index, "src_index"), _index_ptr(dst_index, "dst_index"),
                                  None if weight is None else weight.data_ptr(), src.data_ptr(), out.data_ptr(),
                                  nnz, feat, src.shape[0], rows, dt, _REDUCE_CODES[reduce], wsp, wsn, _stream_handle(dev))
    _lib.check(rc, "geot_gather_reduce")
    return out


def gather_weight_scatter_out(src_index, dst_index, weight, src, out) -> torch.Tensor:
    dev = _require_gpu(src_index, dst_index, weight, src, out)
    L = _lib.load()
    dt = _dtype_code(src, "gather_weight_scatter_sorted")
    if weight.dtype != src.dtype:
        raise RuntimeError(f"expected weight of dtype {src.dtype} but found {weight.dtype}")
    nnz, feat, rows = dst_index.numel(), src.shape[1], out.shape[0]
    with _on_device(dev):
        ws, wsp, wsn = _ws_args(L, dev, nnz, feat, rows, dt)
        rc = L.geot_gather_weight_scatter(_index_ptr(src_index, "src_index"), _index_ptr(dst_index, "dst_index"),
                                          weight.data_ptr(), src.data_ptr(), out.data_ptr(), nnz, feat,
                                          src.shape[0], rows, dt, wsp, wsn, _stream_handle(dev))
    _lib.check(rc, "geot_gather_weight_scatter")
    return out


def mh_spmm_out(src_index, dst_index, weight, src, out, head_major: bool) -> torch.Tensor:
    dev = _require_gpu(src_index, dst_index, weight, src, out)
    L = _lib.load()
    dt = _dtype_code(src, "mh_spmm_sorted")
    if weight.dtype != src.dtype:
        raise RuntimeError(f"expected weight of dtype {src.dtype} but found {weight.dtype}")
    nnz, heads, feat, rows = dst_index.numel(), src.shape[1], src.shape[2], out.shape[0]
    with _on_device(dev):
        ws = workspace(dev, int(L.geot_mh_workspace_bytes(nnz, heads, feat, rows, dt)))
        wsp, wsn = ws.data_ptr(), ws.numel()
        rc = L.geot_mh_spmm(_index_ptr(src_index, "src_index"), _index_ptr(dst_index, "dst_index"),
                            weight.data_ptr(), src.data_ptr(), out.data_ptr(), nnz, heads, feat, src.shape[0],
                            rows, _lib.GEOT_W_HEAD_MAJOR if head_major else _lib.GEOT_W_EDGE_MAJOR, dt,
                            wsp, wsn, _stream_handle(dev))
    _lib.check(rc, "geot_mh_spmm")
    return out


def sddmm_coo_out(src_index, dst_index, mat_1, mat_2, out) -> torch.Tensor:
    dev = _require_gpu(src_index, dst_index, mat_1, mat_2, out)
    L = _lib.load()
    dt = _dtype_code(mat_1, "sddmm_coo")
    with _on_device(dev):
        rc = L.geot_sddmm_coo(_index_ptr(src_index, "src_index"), _index_ptr(dst_index, "dst_index"),
                              mat_1.data_ptr(), mat_2.data_ptr(), out.data_ptr(), dst_index.numel(),
                              mat_1.shape[1], mat_1.shape[0], mat_2.shape[0], dt, _stream_handle(dev))
    _lib.check(rc, "geot_sddmm_coo")
    return out


def mh_sddmm_coo_out(src_index, dst_index, mat_1, mat_2, out, head_major: bool) -> torch.Tensor:
    """out(e, h) = <mat_1[dst_index[e], h], mat_2[src_index[e], h]>; mat_* [rows, H, F]; out [nnz, H] or (head_major) [H, nnz]."""
    dev = _require_gpu(src_index, dst_index, mat_1, mat_2, out)
    L = _lib.load()
    dt = _dtype_code(mat_1, "mh_sddmm_coo")
    with _on_device(dev):
        rc = L.geot_mh_sddmm_coo(_index_ptr(src_index, "src_index"), _index_ptr(dst_index, "dst_index"),
                                 mat_1.data_ptr(), mat_2.data_ptr(), out.data_ptr(), dst_index.numel(), mat_1.shape[1], mat_1.shape[2],
                                 mat_1.shape[0], mat_2.shape[0], _lib.GEOT_W_HEAD_MAJOR if head_major else _lib.GEOT_W_EDGE_MAJOR, dt,
                                 _stream_handle(dev))
    _lib.check(rc, "geot_mh_sddmm_coo")
    return out


def gather_rows_out(index, src, out) -> torch.Tensor:
    dev = _require_gpu(index, src, out)
    L = _lib.load()
    dt = _dtype_code(src, "gather_rows")
    nnz = index.numel()
    feat = src.numel() // max(src.shape[0], 1)
    with _on_device(dev):
        rc = L.geot_gather_rows(_index_ptr(index, "index"), src.data_ptr(), out.data_ptr(), nnz, feat,
                                src.shape[0], dt, _stream_handle(dev))
    _lib.check(rc, "geot_gather_rows")
    return out


def index_probe_out(index: torch.Tensor, out2: torch.Tensor) -> torch.Tensor:
    """out2 (device int64[2]) <- [index[-1], number of descents]; asynchronous on the current stream."""
    dev = _require_gpu(index, out2)
    if out2.dtype != torch.int64 or out2.numel() < 2 or not out2.is_contiguous():
        raise TypeError("index_probe: out2 must be a contiguous int64 tensor of 2 elements")
    with _on_device(dev):
        rc = _lib.load().geot_index_probe(_index_ptr(index, "index"), index.numel(), out2.data_ptr(), _stream_handle(dev))
    _lib.check(rc, "geot_index_probe")
    return out2


def index_probe_range_out(index: torch.Tensor, out4: torch.Tensor) -> torch.Tensor:
    """out4 (device int64[4]) <- [index[-1], number of descents, min, max]; asynchronous on the current stream."""
    dev = _require_gpu(index, out4)
    if out4.dtype != torch.int64 or out4.numel() < 4 or not out4.is_contiguous():
        raise TypeError("index_probe_range: out4 must be a contiguous int64 tensor of 4 elements")
    with _on_device(dev):
        rc = _lib.load().geot_index_probe_range(_index_ptr(index, "index"), index.numel(), out4.data_ptr(), _stream_handle(dev))
    _lib.check(rc, "geot_index_probe_range")
    return out4


def sort_index(index: torch.Tensor, key_max: int):
    """(keys ascending, perm) - the stable sort of `index` (keys in [0, key_max], key_max < 2^32) by geot_sort_index."""
    dev = _require_gpu(index)
    L = _lib.load()
    nnz = index.numel()
    keys, perm = torch.empty_like(index), torch.empty_like(index)
    with _on_device(dev):
        nbytes = L.geot_sort_workspace_bytes(nnz)
        if nbytes == 0:
            raise RuntimeError("geot_sort_workspace_bytes failed")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=index.device)
        rc = L.geot_sort_index(_index_ptr(index, "index"), nnz, int(key_max), keys.data_ptr(), perm.data_ptr(), ws.data_ptr(),
                               ws.numel(), _stream_handle(dev))
    _lib.check(rc, "geot_sort_index")
    return keys, perm


def csr_gws_out(indptr, indices, weight, src, out) -> torch.Tensor:
    """out[r] = sum_{e in [indptr[r], indptr[r+1])} weight[e] * src[indices[e]]; weight None = ones."""
    tensors = [indptr, indices, src, out] + ([weight] if weight is not None else [])
    dev = _require_gpu(*tensors)
    L = _lib.load()
    dt = _dtype_code(src, "csr_gws")
    nrow, nnz, feat, rows = indptr.numel() - 1, indices.numel(), src.shape[1], out.shape[0]
    with _on_device(dev):
        nbytes = int(L.geot_csr_workspace_bytes(nnz, feat, rows, dt))
        ws = workspace(dev, nbytes)
        rc = L.geot_csr_gws(_index_ptr(indptr, "indptr"), _index_ptr(indices, "indices"),
                            None if weight is None else weight.data_ptr(), src.data_ptr(), out.data_ptr(),
                            nrow, nnz, feat, src.shape[0], rows, dt, ws.data_ptr(), ws.numel(), _stream_handle(dev))
    _lib.check(rc, "geot_csr_gws")
    return out


def coo_to_csr_out(coo_row, rowptr, assume_sorted: bool = False) -> torch.Tensor:
    """rowptr int32 [nrow+1] from COO row ids (int64).  Unsorted input: finishes with a prefix sum."""
    dev = _require_gpu(coo_row, rowptr)
    L = _lib.load()
    if rowptr.dtype != torch.int32:
        raise RuntimeError("rowptr must be int32")
    with _on_device(dev):
        rc = L.geot_coo_to_csr(_index_ptr(coo_row, "coo_row"), coo_row.numel(), rowptr.numel() - 1,
                               rowptr.data_ptr(), 1 if assume_sorted else 0, _stream_handle(dev))
    _lib.check(rc, "geot_coo_to_csr")
    if not assume_sorted:
        torch.cumsum(rowptr, 0, dtype=torch.int32, out=rowptr)
    return rowptr


# ---- measurement hooks ---------------------------------------------------------------------------
def profile_enable(on: bool) -> None:
    _lib.load().geot_profile_enable(1 if on else 0)


def profile_reset() -> None:
    _lib.load().geot_profile_reset()


def profile_read() -> dict:
    a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    n = ctypes.c_int64()
    _lib.load().geot_profile_read(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(n))
    return {"main_ms": a.value, "fixup_ms": b.value, "aux_ms": c.value, "calls": n.value}


def last_kernel() -> str:
    """Name of the dominant kernel this thread's last call launched, as rocprofv3 prints it (geot_last_kernel)."""
    L = _lib.load()
    L.geot_last_kernel.restype = ctypes.c_char_p
    L.geot_last_kernel.argtypes = []
    return (L.geot_last_kernel() or b"").decode()


def profile_box(buf: torch.Tensor, iters: int = 5) -> dict:
    """Read ceiling (pure nt 16-B-per-lane read of `buf`) and shader clock of THIS box, now."""
    dev = _require_gpu(buf)
    g, c = ctypes.c_double(), ctypes.c_double()
    with _on_device(dev):
        rc = _lib.load().geot_profile_box(buf.data_ptr(), buf.numel() * buf.element_size(), iters, ctypes.byref(g),
                                          ctypes.byref(c), _stream_handle(dev))
    _lib.check(rc, "geot_profile_box")
    return {"read_ceiling_gbps": g.value, "sclk_mhz": c.value}


def profile_box_rows(table: torch.Tensor, iters: int = 3, out: Optional[torch.Tensor] = None, run: int = 0) -> dict:
    """Random-row ceiling of THIS box on the caller's own [rows, F] table (geot_profile_box_rows): uniform-random rows, 16 reads in
    flight per lane, nothing else - default-policy and non-temporal loads; with `out` (a scratch [rows', F] tensor, overwritten) and
    `run`: also the read / write MIX - one row written per `run` rows read.  All rates are GB/s of rows READ."""
    dev = _require_gpu(table)
    if table.dim() != 2 or not table.is_contiguous():
        raise ValueError("profile_box_rows: a contiguous [rows, F] table")
    a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    mix = out is not None and run >= 1
    if mix and (out.dim() != 2 or not out.is_contiguous() or out.shape[1] * out.element_size() != table.shape[1] * table.element_size()):
        raise ValueError("profile_box_rows: `out` must be a contiguous tensor of rows as wide as the table's")
    with _on_device(dev):
        rc = _lib.load().geot_profile_box_rows(table.data_ptr(), table.shape[0], table.shape[1] * table.element_size(), iters,
                                               ctypes.byref(a), ctypes.byref(b), out.data_ptr() if mix else None, out.shape[0] if mix else 0,
                                               int(run) if mix else 0, ctypes.byref(c) if mix else None, _stream_handle(dev))
    _lib.check(rc, "geot_profile_box_rows")
    return {"random_row_gbps": a.value, "random_row_gbps_nt": b.value, "best_gbps": max(a.value, b.value),
            "mix_row_gbps": c.value if mix else None, "mix_run": int(run) if mix else None}


def tune(edges_per_group: int = 0, vec: int = 0, nontemporal: int = -1, lpr_log2: int = -1) -> None:
    _lib.load().geot_tune(edges_per_group, vec, nontemporal, lpr_log2)
    _ws_bytes.clear()  # the tile shape, hence the workspace need, follows the plan


def set_option(name: str, value: int) -> None:
    """Named switches of the library (include/geot_hip_dev.h).  A name the loaded build does not know raises: the switches of
    measured-and-rejected variants ("slab_probe", "slab_pair", ...) exist in the development build only (GEOT_HIP_LIB=dev)."""
    rc = _lib.load().geot_set_option(name.encode(), int(value))
    _lib.check(rc, "geot_set_option")
    _ws_bytes.clear()


def build_info() -> str:
    return _lib.load().geot_build_info().decode()
