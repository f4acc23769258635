"""ctypes front end of oracle/_ref/libgeot_ref*.so -- the REFERENCE's own CPU index_scatter.

TEST INFRASTRUCTURE ONLY.  The library is compiled in THIS container from
/root/reference/csrc/cpu/index_scatter_cpu.cpp + oracle/ref_harness.cpp (``make -C oracle ref``);
on the GPU box only the prebuilt files exist (oracle/_ref/ travels with the snapshot).
``available()`` is False where neither the files nor the reference tree exist.

Reminder: as shipped, the reference CPU kernel sums ``src[index[n]]``
rather than ``src[n]`` (csrc/cpu/index_scatter_cpu.cpp:110-112).  The identity used for
parity is  ref(index, src) == oracle(index, src[index])  -- bit-exact.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF_DIR = os.path.join(_HERE, "_ref")
REFERENCE_ROOT = "/root/reference"
_libs: dict = {}


def _path(omp: bool) -> str:
    return os.path.join(_REF_DIR, "libgeot_ref_omp.so" if omp else "libgeot_ref.so")


def can_build() -> bool:
    return os.path.exists(os.path.join(REFERENCE_ROOT, "csrc/cpu/index_scatter_cpu.cpp"))


def build() -> bool:
    """Compile the reference CPU path in place (about 90 s).  No-op without /root/reference."""
    if not can_build():
        return False
    subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)
    return True


def available(omp: bool = False) -> bool:
    return os.path.exists(_path(omp))


def lib(omp: bool = False) -> ctypes.CDLL:
    if omp not in _libs:
        if not available(omp):
            raise FileNotFoundError(
                f"{_path(omp)} missing: run `make -C oracle ref` where /root/reference exists")
        import torch  # noqa: F401  -- puts libtorch/libc10 into the process before dlopen
        L = ctypes.CDLL(_path(omp))
        L.geot_ref_last_error.restype = ctypes.c_char_p
        _libs[omp] = L
    return _libs[omp]


def index_scatter_cpu(index, src, reduce: str = "sum", sorted: bool = True, omp: bool = False,
                      threads: int = 0, rows: int | None = None):
    """The reference's index_scatter_cpu on (index, src) exactly as shipped."""
    index = np.ascontiguousarray(index, dtype=np.int64)
    src = np.ascontiguousarray(src)
    # bfloat16 travels as raw uint16 bit patterns (numpy has no bf16): pass dtype_code=3 explicitly
    code = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.float16): 2,
            np.dtype(np.uint16): 3}.get(src.dtype)
    if code is None:
        raise TypeError("float32 / float64 / float16 / uint16(=bfloat16 bits) only")
    nnz = index.shape[0]
    F = int(np.prod(src.shape[1:], dtype=np.int64)) if src.ndim > 1 else 1
    K = int(index[-1]) + 1 if rows is None else int(rows)
    out = np.empty((K,) + src.shape[1:], dtype=src.dtype)
    L = lib(omp)
    if threads:
        L.geot_ref_set_num_threads(ctypes.c_int(threads))
    rc = L.geot_ref_index_scatter_cpu(
        index.ctypes.data_as(ctypes.c_void_p), src.ctypes.data_as(ctypes.c_void_p),
        out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(nnz), ctypes.c_int64(F),
        ctypes.c_int64(K), ctypes.c_int(code),
        reduce.encode(), ctypes.c_int(1 if sorted else 0))
    if rc != 0:
        raise RuntimeError(L.geot_ref_last_error().decode())
    return out
