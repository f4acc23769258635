"""numpy/ctypes front end of oracle/libgeot_oracle.so (the C restatement).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Parity status: pinned for
index_scatter against the compiled reference (oracle/_ref) and the committed golden
vectors; gather ops pinned against the reference tests' torch comparators.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgeot_oracle.so")
_lib = None

REDUCE_CODES = {"max": 0, "amax": 0, "mean": 1, "min": 2, "amin": 2, "sum": 3, "prod": 4}

_i64p = ctypes.POINTER(ctypes.c_int64)
_vp = ctypes.c_void_p
_i64 = ctypes.c_int64


def build(force: bool = False) -> str:
    """Compile geot_oracle.c with gcc (seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "geot_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.geot_oracle_out_rows.restype = _i64
        _lib.geot_oracle_segment_table.restype = _i64
    return _lib


def _idx(a) -> np.ndarray:
    a = np.ascontiguousarray(a)
    if a.dtype != np.int64:
        raise TypeError(f"index must be int64, got {a.dtype}")
    if a.ndim != 1:
        raise ValueError("index must be 1 dimensional")
    return a


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_vp)


def _check(rc: int, what: str):
    if rc == 0:
        return
    msg = {-1: "invalid argument", -2: "index out of range", -3: "out of memory"}.get(rc, str(rc))
    raise ValueError(f"oracle {what}: {msg}")


def max_threads() -> int:
    return int(lib().geot_oracle_max_threads())


def out_rows(index) -> int:
    """rows = index[-1] + 1  (csrc/index_scatter.cpp:15-20)."""
    index = _idx(index)
    r = int(lib().geot_oracle_out_rows(_ptr(index), _i64(index.shape[0])))
    if r < 0:
        raise IndexError("index is empty: the reference evaluates index[-1]")
    return r


def index_scatter(index, src, rows: int | None = None, acc64: bool = False, refquirk: bool = False):
    """dst[index[i]] += src[i], sequential in edge order.  src is [nnz, ...] (dim 0)."""
    index = _idx(index)
    src = np.ascontiguousarray(src)
    nnz = index.shape[0]
    if src.shape[0] != nnz:
        raise ValueError("index length must be equal to src dimension size")
    K = out_rows(index) if rows is None else int(rows)
    F = int(np.prod(src.shape[1:], dtype=np.int64)) if src.ndim > 1 else 1
    out = np.empty((K,) + src.shape[1:], dtype=src.dtype)
    if src.dtype == np.float32:
        name = "geot_oracle_index_scatter_f32"
        if acc64:
            name += "_acc64"
        if refquirk:
            name = "geot_oracle_index_scatter_refquirk_f32"
    elif src.dtype == np.float64:
        name = "geot_oracle_index_scatter_f64"
    else:
        raise TypeError(f"unsupported dtype {src.dtype}")
    rc = getattr(lib(), name)(_ptr(index), _ptr(src), _ptr(out), _i64(nnz), _i64(F), _i64(K))
    _check(rc, name)
    return out


def index_scatter_3pass(index, src, reduce: str = "sum", threads: int = 0, rows: int | None = None):
    """The reference CPU algorithm (csrc/cpu/index_scatter_cpu.cpp:25-122), intended operand."""
    index = _idx(index)
    src = np.ascontiguousarray(src, dtype=np.float32)
    nnz = index.shape[0]
    K = out_rows(index) if rows is None else int(rows)
    F = int(np.prod(src.shape[1:], dtype=np.int64)) if src.ndim > 1 else 1
    out = np.empty((K,) + src.shape[1:], dtype=np.float32)
    if reduce not in REDUCE_CODES:
        raise ValueError(
            f"reduce argument must be either sum, prod, mean, amax or amin, got {reduce}")
    rc = lib().geot_oracle_index_scatter_3pass_f32(
        _ptr(index), _ptr(src), _ptr(out), _i64(nnz), _i64(F), _i64(K),
        ctypes.c_int(REDUCE_CODES[reduce]), ctypes.c_int(threads))
    _check(rc, "index_scatter_3pass")
    return out


def segment_table(index):
    """(rows, offsets) of the non-empty segments (csrc/cpu/index_scatter_cpu.cpp:38-75)."""
    index = _idx(index)
    nnz = index.shape[0]
    rows = np.empty(nnz, dtype=np.int64)
    offs = np.empty(nnz + 1, dtype=np.int64)
    n = int(lib().geot_oracle_segment_table(_ptr(index), _i64(nnz), _ptr(rows), _ptr(offs)))
    if n < 0:
        raise IndexError("index is empty")
    return rows[:n].copy(), offs[: n + 1].copy()


def gather_weight_scatter(src_index, dst_index, weight, src, rows: int | None = None,
                          acc64: bool = False):
    """dst[d[e]] += src[s[e]] * w[e]; weight None -> gather_scatter."""
    src_index, dst_index = _idx(src_index), _idx(dst_index)
    src = np.ascontiguousarray(src)
    if src.ndim != 2:
        raise ValueError("src must be 2 dimensional")
    nnz = dst_index.shape[0]
    K = out_rows(dst_index) if rows is None else int(rows)
    F = src.shape[1]
    out = np.empty((K, F), dtype=src.dtype)
    w = None if weight is None else np.ascontiguousarray(weight, dtype=src.dtype)
    name = {np.dtype(np.float32): "geot_oracle_gather_weight_scatter_f32",
            np.dtype(np.float64): "geot_oracle_gather_weight_scatter_f64"}[src.dtype]
    if acc64 and src.dtype == np.float32:
        name += "_acc64"
    rc = getattr(lib(), name)(_ptr(src_index), _ptr(dst_index), None if w is None else _ptr(w),
                              _ptr(src), _ptr(out), _i64(nnz), _i64(F), _i64(src.shape[0]), _i64(K))
    _check(rc, name)
    return out


def gather_scatter(src_index, dst_index, src, rows: int | None = None, acc64: bool = False):
    return gather_weight_scatter(src_index, dst_index, None, src, rows=rows, acc64=acc64)


def mh_spmm(src_index, dst_index, weight, src, transposed: bool = False,
            rows: int | None = None, acc64: bool = False):
    """dst[d[e],h,:] += w[e,h] * src[s[e],h,:]; transposed -> weight is [H, nnz]."""
    src_index, dst_index = _idx(src_index), _idx(dst_index)
    src = np.ascontiguousarray(src)
    if src.ndim != 3:
        raise ValueError("src must be 3 dimensional")
    nnz = dst_index.shape[0]
    K = out_rows(dst_index) if rows is None else int(rows)
    _, H, F = src.shape
    w = np.ascontiguousarray(weight, dtype=src.dtype)
    if w.shape != ((H, nnz) if transposed else (nnz, H)):
        raise ValueError("Invalid weight size")
    out = np.empty((K, H, F), dtype=src.dtype)
    name = {np.dtype(np.float32): "geot_oracle_mh_spmm_f32",
            np.dtype(np.float64): "geot_oracle_mh_spmm_f64"}[src.dtype]
    if acc64 and src.dtype == np.float32:
        name += "_acc64"
    rc = getattr(lib(), name)(_ptr(src_index), _ptr(dst_index), _ptr(w), _ptr(src), _ptr(out),
                              _i64(nnz), _i64(H), _i64(F), _i64(src.shape[0]), _i64(K),
                              ctypes.c_int(1 if transposed else 0))
    _check(rc, name)
    return out


def sddmm_coo(src_index, dst_index, mat_1, mat_2, acc64: bool = False):
    """out[e] = <mat_1[dst_index[e]], mat_2[src_index[e]]>."""
    src_index, dst_index = _idx(src_index), _idx(dst_index)
    m1 = np.ascontiguousarray(mat_1, dtype=np.float32)
    m2 = np.ascontiguousarray(mat_2, dtype=np.float32)
    nnz = dst_index.shape[0]
    out = np.empty(nnz, dtype=np.float32)
    name = "geot_oracle_sddmm_coo_f32" + ("_acc64" if acc64 else "")
    rc = getattr(lib(), name)(_ptr(src_index), _ptr(dst_index), _ptr(m1), _ptr(m2), _ptr(out),
                              _i64(nnz), _i64(m1.shape[1]), _i64(m1.shape[0]), _i64(m2.shape[0]))
    _check(rc, name)
    return out


def gather_rows(index, src):
    """dst[e] = src[index[e]] (backward of index_scatter)."""
    index = _idx(index)
    src = np.ascontiguousarray(src, dtype=np.float32)
    F = int(np.prod(src.shape[1:], dtype=np.int64)) if src.ndim > 1 else 1
    out = np.empty((index.shape[0],) + src.shape[1:], dtype=np.float32)
    rc = lib().geot_oracle_gather_rows_f32(_ptr(index), _ptr(src), _ptr(out),
                                           _i64(index.shape[0]), _i64(F), _i64(src.shape[0]))
    _check(rc, "gather_rows")
    return out


def csr_gws(indptr, indices, weight, src, out_rows: int | None = None, acc64: bool = False):
    """out[r] = sum_e weight[e] * src[indices[e]] over row r's nonzeros; out_rows defaults to
    len(indptr) (the reference shim's row count, csrc/csr_gws.cpp:29-31)."""
    indptr, indices = _idx(indptr), _idx(indices)
    src = np.ascontiguousarray(src, dtype=np.float32)
    w = None if weight is None else np.ascontiguousarray(weight, dtype=np.float32)
    nrow = indptr.shape[0] - 1
    rows = indptr.shape[0] if out_rows is None else int(out_rows)
    out = np.empty((rows, src.shape[1]), dtype=np.float32)
    name = "geot_oracle_csr_gws_f32" + ("_acc64" if acc64 else "")
    rc = getattr(lib(), name)(_ptr(indptr), _ptr(indices), None if w is None else _ptr(w), _ptr(src), _ptr(out),
                              _i64(nrow), _i64(src.shape[1]), _i64(src.shape[0]), _i64(rows))
    _check(rc, name)
    return out


def coo_to_csr(coo_row, nrow: int | None = None):
    coo_row = _idx(coo_row)
    nrow = int(coo_row.max()) + 1 if nrow is None else int(nrow)
    rowptr = np.empty(nrow + 1, dtype=np.int32)
    rc = lib().geot_oracle_coo_to_csr(_ptr(coo_row), _i64(coo_row.shape[0]), _i64(nrow), _ptr(rowptr))
    _check(rc, "coo_to_csr")
    return rowptr
