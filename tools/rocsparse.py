"""rocSPARSE CSR SpMM through its generic API (ctypes on librocsparse.so) - the COMPARATOR the north star names
("gather_weight_scatter SpMM beating rocSPARSE CSR SpMM on the same graph"; the reference measures the same
comparison against cuSPARSE, benchmark/bench_spmm.py:81-138).  Used by bench.py (`secondary.gws_cfg3`),
tests/test_gpu_round2.py and tools/rocsparse_spmm.py.  Not part of the product.

rocSPARSE's native index width is int32 (what torch.sparse hands it too); the geot op reads the int64 COO the
reference's API prescribes.  The preprocess stage (per-matrix analysis, like our cached structures) is excluded
from the timing; its buffer size is reported.
"""
import ctypes

import torch

OP_NONE, BASE0, I32, F32R, ROW = 111, 0, 2, 151, 0
STAGE_BUF, STAGE_PRE, STAGE_COMPUTE = 1, 2, 3
#: rocsparse_spmm_alg values that accept CSR (rocsparse-types.h)
ALGS = {"default": 0, "csr": 1, "csr_row_split": 4, "csr_nnz_split": 5, "csr_merge_path": 9}


class CsrSpMM:
    """y = A @ x for A = CSR(rowptr int32 [rows+1], col int32 [nnz], val fp32 [nnz]), x / y row-major fp32."""

    def __init__(self, rows, cols, rowptr, col, val, x, y):
        self.L = L = ctypes.CDLL("librocsparse.so")
        vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        self.keep = (rowptr, col, val, x, y)
        self.handle = vp()
        assert L.rocsparse_create_handle(ctypes.byref(self.handle)) == 0
        L.rocsparse_set_stream(self.handle, vp(torch.cuda.current_stream().cuda_stream))
        self.A, self.B, self.C = vp(), vp(), vp()
        L.rocsparse_create_csr_descr.argtypes = [ctypes.POINTER(vp), i64, i64, i64, vp, vp, vp, ci, ci, ci, ci]
        assert L.rocsparse_create_csr_descr(ctypes.byref(self.A), rows, cols, col.numel(), rowptr.data_ptr(),
                                            col.data_ptr(), val.data_ptr(), I32, I32, BASE0, F32R) == 0
        L.rocsparse_create_dnmat_descr.argtypes = [ctypes.POINTER(vp), i64, i64, i64, vp, ci, ci]
        F = x.shape[1]
        assert L.rocsparse_create_dnmat_descr(ctypes.byref(self.B), cols, F, F, x.data_ptr(), F32R, ROW) == 0
        assert L.rocsparse_create_dnmat_descr(ctypes.byref(self.C), rows, F, F, y.data_ptr(), F32R, ROW) == 0
        L.rocsparse_spmm.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, ci, ci, ci, ctypes.POINTER(ctypes.c_size_t), vp]
        self.alpha, self.beta = ctypes.c_float(1.0), ctypes.c_float(0.0)
        self.alg = None
        self.buf = None
        self.size = ctypes.c_size_t(0)

    def _call(self, stage, bufptr):
        return self.L.rocsparse_spmm(self.handle, OP_NONE, OP_NONE, ctypes.byref(self.alpha), self.A, self.B,
                                     ctypes.byref(self.beta), self.C, F32R, self.alg, stage, ctypes.byref(self.size), bufptr)

    def prepare(self, alg_name: str) -> bool:
        """buffer-size + preprocess stages of one algorithm; False if rocSPARSE rejects it for this layout."""
        self.alg = ALGS[alg_name]
        self.size = ctypes.c_size_t(0)
        if self._call(STAGE_BUF, None) != 0:
            return False
        self.buf = torch.empty(max(self.size.value, 16), dtype=torch.uint8, device=self.keep[3].device)
        return self._call(STAGE_PRE, self.buf.data_ptr()) == 0

    def run(self):
        rc = self._call(STAGE_COMPUTE, self.buf.data_ptr())
        assert rc == 0, f"rocsparse_spmm compute stage returned {rc}"

    @property
    def buffer_bytes(self) -> int:
        return int(self.size.value)


def csr_from_sorted_coo(dst_index: torch.Tensor, src_index: torch.Tensor, rows: int):
    """int32 CSR arrays of a dst-sorted COO edge list."""
    rowptr = torch.zeros(rows + 1, dtype=torch.int32, device=dst_index.device)
    rowptr[1:] = torch.cumsum(torch.bincount(dst_index, minlength=rows), 0).int()
    return rowptr, src_index.int()


def device_ms(fn, iters: int, warmup: int = 2) -> float:
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def best_csr_spmm(dst_index, src_index, weight, x, rows, iters: int = 3, algs=("csr_nnz_split", "csr_merge_path", "csr_row_split", "default")):
    """Times every accepted CSR algorithm; returns (best dict, all dicts, y of the best)."""
    rowptr, col = csr_from_sorted_coo(dst_index, src_index, rows)
    y = torch.empty(rows, x.shape[1], device=x.device)
    op = CsrSpMM(rows, x.shape[0], rowptr, col, weight, x, y)
    results = []
    for name in algs:
        if not op.prepare(name):
            results.append({"algorithm": name, "ms": None, "note": "rejected by rocSPARSE for this layout"})
            continue
        ms = device_ms(op.run, iters, warmup=1)
        results.append({"algorithm": name, "ms": ms, "preprocess_buffer_bytes": op.buffer_bytes})
    ok = [r for r in results if r["ms"] is not None]
    best = min(ok, key=lambda r: r["ms"]) if ok else None
    if best is not None:
        op.prepare(best["algorithm"])
        op.run()
        torch.cuda.synchronize()
    return best, results, y
