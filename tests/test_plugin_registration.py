"""What geot_amd/_C.so registers, and that a Python layer which DEFINES geot::gather_scatter / gather_weight_scatter /
csr_gws itself - as the reference's wrappers do with torch.library.custom_op (geot/gather_scatter.py:7,
geot/gather_weight_scatter.py:15, geot/csr_gws.py:25) - loads on top of it without a duplicate registration, in either
order (the reference's geot/__init__.py:4-19 imports its wrappers first and the library second).

CPU suite: registration only (fresh processes).  `-m gpu`: the same stand-in layer computing on the MI355X, its results
and gradients against torch.  The committed fixture tests/golden/ref_wrappers_on_plugin.json is what the reference's
UNMODIFIED files produced on this plugin in the build container (tests/golden/make_ref_on_plugin.py).
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, ROOT

PLUGIN = os.path.join(ROOT, "geot_amd", "_C.so")

# the operators the reference's csrc/*.cpp define (csrc/index_scatter.cpp:43-47, gather_scatter.cpp:16-17,
# gather_weight_scatter.cpp:12-16, csr_gws.cpp:12-13, mh_spmm.cpp:23)
REFERENCE_CSRC_OPS = {"geot::index_scatter", "geot::gather_scatter_impl", "geot::gather_weight_scatter_impl",
                      "geot::sddmm_coo_impl", "geot::csr_gws_impl", "geot::mh_spmm"}
# ... and the ones its Python defines
PYTHON_DEFINED = {"geot::gather_scatter", "geot::gather_weight_scatter", "geot::csr_gws", "geot::coo_to_csr"}

# A reference-STYLE Python layer in our own words: custom_op definitions of the three names whose bodies forward to the
# `*_impl` ops, a fake rule with a dynamic row count, and a backward that re-sorts the edges by source and calls the
# forward op on the swapped lists.  (Structure dictated by the API; nothing here is the reference's text.)
LAYER = r'''
import torch

def define_layer():
    ops = torch.ops.geot

    @torch.library.custom_op("geot::gather_scatter", mutates_args=())
    def gather_scatter(src_index: torch.Tensor, dst_index: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
        return ops.gather_scatter_impl(src_index, dst_index, src)

    @torch.library.custom_op("geot::gather_weight_scatter", mutates_args=())
    def gather_weight_scatter(src_index: torch.Tensor, dst_index: torch.Tensor, weight: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
        return ops.gather_weight_scatter_impl(src_index, dst_index, weight, src)

    @torch.library.custom_op("geot::csr_gws", mutates_args=())
    def csr_gws(csrptr: torch.Tensor, csrind: torch.Tensor, weight: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
        return ops.csr_gws_impl(csrptr, csrind, weight, src)

    def dyn_rows(*args):
        src = args[-1]
        return src.new_empty([torch.library.get_ctx().new_dynamic_size(), src.shape[1]])

    for name in ("gather_scatter", "gather_weight_scatter", "csr_gws"):
        torch.library.register_fake("geot::" + name)(dyn_rows)

    def gs_setup(ctx, inputs, output):
        ctx.save_for_backward(inputs[0], inputs[1])

    def gs_backward(ctx, grad):
        s, d = ctx.saved_tensors
        order = torch.sort(s).indices
        return None, None, ops.gather_scatter_impl(d[order], s[order], grad.contiguous())

    def gws_setup(ctx, inputs, output):
        ctx.save_for_backward(*inputs)

    def gws_backward(ctx, grad):
        s, d, w, x = ctx.saved_tensors
        order = torch.sort(s).indices
        g = grad.contiguous()
        dsrc = ops.gather_weight_scatter_impl(d[order], s[order], w[order], g)
        dw = ops.sddmm_coo_impl(s.to(torch.int32), d.to(torch.int32), g, x)     # <g[d[e]], x[s[e]]>, original edge order
        return None, None, dw, dsrc

    torch.library.register_autograd("geot::gather_scatter", gs_backward, setup_context=gs_setup)
    torch.library.register_autograd("geot::gather_weight_scatter", gws_backward, setup_context=gws_setup)
    return gather_scatter, gather_weight_scatter, csr_gws
'''

REGISTRATION = LAYER + r'''
import json, sys
plugin, order = sys.argv[1:3]

def has_schema(q):
    try:
        torch._C._dispatch_find_schema_or_throw(q, "")
        return True
    except RuntimeError:
        return False

res = {}
if order == "plugin_first":
    torch.ops.load_library(plugin)
    every = sorted(n for n in torch._C._dispatch_get_all_op_names() if n.startswith("geot::"))
    res["defines"] = [n for n in every if has_schema(n)]
    res["implements_only"] = [n for n in every if not has_schema(n)]
    fns = define_layer()
else:
    fns = define_layer()
    torch.ops.load_library(plugin)
res["schemas"] = [str(getattr(torch.ops.geot, n).default._schema) for n in ("gather_scatter", "gather_weight_scatter", "csr_gws")]
dump = torch._C._dispatch_dump("geot::gather_weight_scatter")
res["cuda_kernel_is_the_plugins"] = any(l.startswith("CUDA: ") and "torch_ops.cpp" in l for l in dump.splitlines())
try:
    fns[0](torch.tensor([0, 1]), torch.tensor([0, 1]), torch.rand(3, 4))
    res["cpu_call"] = "no error"
except RuntimeError as e:
    res["cpu_call"] = str(e).splitlines()[0]
print("RESULT " + json.dumps(res))
'''


def _run(script, *args, timeout=600):
    p = subprocess.run([sys.executable, "-c", script, *args], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
    assert p.returncode == 0 and lines, p.stdout[-1500:] + p.stderr[-3000:]
    return json.loads(lines[-1][7:])


@pytest.mark.parametrize("order", ["plugin_first", "wrappers_first"])
def test_python_layer_can_define_the_public_ops_on_top_of_the_plugin(order):
    res = _run(REGISTRATION, PLUGIN, order)
    assert res["schemas"] == ["geot::gather_scatter(Tensor src_index, Tensor dst_index, Tensor src) -> Tensor",
                              "geot::gather_weight_scatter(Tensor src_index, Tensor dst_index, Tensor weight, Tensor src) -> Tensor",
                              "geot::csr_gws(Tensor csrptr, Tensor csrind, Tensor weight, Tensor src) -> Tensor"]
    assert res["cuda_kernel_is_the_plugins"]                      # GPU tensors go dispatcher -> C++ host layer, no Python hop
    assert "geot::gather_scatter: CPU tensors are not supported" in res["cpu_call"]       # the call landed in the plugin
    if order == "plugin_first":
        defines = set(res["defines"])
        assert REFERENCE_CSRC_OPS <= defines                      # everything csrc/*.cpp defines ...
        assert not (PYTHON_DEFINED & defines)                     # ... and nothing the reference's Python defines
        assert set(res["implements_only"]) == PYTHON_DEFINED - {"geot::coo_to_csr"}


def test_fixture_of_the_reference_wrappers_on_this_plugin():
    """tests/golden/ref_wrappers_on_plugin.json: the reference's own three files imported on top of _C.so (build
    container).  Here: the record says both orders imported, the plugin serves the device keys, and the set of operators
    the plugin defined then is the set it defines now."""
    fx = json.load(open(os.path.join(GOLDEN, "ref_wrappers_on_plugin.json")))
    assert [r["order"] for r in fx["runs"]] == ["plugin_first", "wrappers_first"]
    for run in fx["runs"]:
        assert run["import_error"] is None
        for name, keys in run["dispatch"].items():
            assert keys["CUDA"] == "plugin" and keys["CPU"] == "plugin", (name, keys)
            assert keys["Autograd"] == "python" and keys["Meta"] == "python"          # the wrappers' own autograd / fake rules
        for name, err in run["cpu_call"].items():
            assert err and "CPU tensors are not supported by geot_amd" in err
        for name, f in run["fake"].items():
            assert f == {"ndim": 2, "cols": 3, "rows_is_symbolic": True, "dtype": "torch.float32"}
    live = _run(REGISTRATION, PLUGIN, "plugin_first")
    assert fx["runs"][0]["plugin_defines"] == live["defines"]
    assert fx["runs"][0]["plugin_implements_only"] == live["implements_only"]


def test_geot_amd_defines_the_public_ops_in_python():
    import torch

    import geot_amd  # noqa: F401
    from geot_amd import ops
    assert not ops._FOREIGN
    for name, schema in ops.PUBLIC_SCHEMAS.items():
        assert str(getattr(torch.ops.geot, name).default._schema) == "geot::" + schema
        dump = torch._C._dispatch_dump("geot::" + name)
        assert any(l.startswith("CUDA: ") and "torch_ops.cpp" in l for l in dump.splitlines())
    text = open(os.path.join(ROOT, "geot_amd", "csrc", "torch_ops.cpp")).read()
    for name in ops.PUBLIC_SCHEMAS:
        assert f'm.def("{name}(' not in text


COMPUTE = LAYER + r'''
import sys
torch.ops.load_library(sys.argv[1])
gather_scatter, gather_weight_scatter, csr_gws = define_layer()
dev = "cuda"
torch.manual_seed(0)
n, nnz, F = 700, 30000, 64
d = torch.sort(torch.randint(0, n, (nnz,), device=dev)).values
d[-1] = n - 1
s = torch.randint(0, n, (nnz,), device=dev)
s[0] = n - 1                                  # (the reference's backward sizes its result by max(src_index) + 1)
w = torch.rand(nnz, device=dev, requires_grad=True)
x = torch.rand(n, F, device=dev, requires_grad=True)
close = lambda a, b: torch.allclose(a, b, rtol=1e-5, atol=1e-4)
# forward, through the Python function and through the dispatcher name
ref_gs = torch.zeros(n, F, device=dev).index_add_(0, d, x.detach()[s])
assert close(gather_scatter(s, d, x.detach()), ref_gs) and close(torch.ops.geot.gather_scatter(s, d, x.detach()), ref_gs)
ref = torch.zeros(n, F, device=dev).index_add(0, d, x[s] * w[:, None])
y = gather_weight_scatter(s, d, w, x)
assert close(y, ref)
g = torch.rand_like(y)
dw_ref, dx_ref = torch.autograd.grad(ref, (w, x), g)
dw, dx = torch.autograd.grad(y, (w, x), g)
assert close(dx, dx_ref) and close(dw, dw_ref)
x2 = x.detach().clone().requires_grad_(True)
(dx2,) = torch.autograd.grad(gather_scatter(s, d, x2), (x2,), g)
assert close(dx2, torch.autograd.grad(torch.zeros(n, F, device=dev).index_add(0, d, x2[s]), (x2,), g)[0])
rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
rowptr[1:] = torch.cumsum(torch.bincount(d, minlength=n), 0)
out = csr_gws(rowptr.int(), s.int(), w.detach(), x.detach())
assert out.shape == (n + 1, F) and close(out[:n], ref.detach())
# under torch.compile the wrapper's fake rule gives the shape and the plugin's kernel runs
f = torch.compile(lambda a, b, c, e: gather_weight_scatter(a, b, c, e) * 2.0, fullgraph=True)
assert close(f(s, d, w.detach(), x.detach()), ref.detach() * 2.0)
print("RESULT {}")
'''


@pytest.mark.gpu
def test_reference_style_layer_computes_on_the_plugin():
    _run(COMPUTE, PLUGIN, timeout=900)
