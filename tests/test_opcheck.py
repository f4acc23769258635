"""torch.library.opcheck over every geot:: operator (`-m gpu`) - what the reference's test/opcheck.py sets out to do
(it loads `_C.so` and imports `opcheck`, nothing more): schema honesty (no hidden mutation or aliasing of inputs), the
fake-tensor rule against the real result (dtype, device, rank, dynamic row count), the autograd registration, and the
AOT-dispatch path `torch.compile` takes.  Operators whose row count is read from the data (index[-1] + 1, the reference's
rule, `ctx.new_dynamic_size()` in its fake impls) are traced with dynamic shapes only - a static trace cannot hold an
unbacked size, for them as for torch.nonzero; the `*_rows` forms, whose row count is an argument, pass the static one too."""
import pytest
import torch
from torch.library import opcheck

pytestmark = pytest.mark.gpu
UTILS = ("test_schema", "test_autograd_registration", "test_faketensor", "test_aot_dispatch_static", "test_aot_dispatch_dynamic")


@pytest.fixture(scope="module")
def graph():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import geot_amd  # noqa: F401  (registers fakes and autograd)
    torch.manual_seed(3)
    n, nnz, F, H = 200, 3000, 16, 4
    di = torch.sort(torch.randint(0, n, (nnz,), device="cuda")).values
    di[-1] = n - 1
    si = torch.randint(0, n, (nnz,), device="cuda")
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    rowptr[1:] = torch.bincount(di, minlength=n).cumsum(0)
    return dict(n=n, nnz=nnz, F=F, H=H, di=di, si=si, rowptr=rowptr,
                w=torch.rand(nnz, device="cuda"), wh=torch.rand(nnz, H, device="cuda"),
                x=torch.rand(n, F, device="cuda"), xh=torch.rand(n, H, F // H, device="cuda"),
                e=torch.rand(nnz, F, device="cuda"))


def cases(g):
    G = torch.ops.geot
    rg = lambda t: t.clone().requires_grad_(True)  # noqa: E731
    return {
        "index_scatter": (G.index_scatter.default, (0, g["di"], g["e"], "sum", True)),
        "index_scatter_grad": (G.index_scatter.default, (0, g["di"], rg(g["e"]), "sum", True)),
        "index_scatter_mean": (G.index_scatter.default, (0, g["di"], g["e"], "mean", False)),
        "gather_scatter": (G.gather_scatter.default, (g["si"], g["di"], rg(g["x"]))),
        "gather_weight_scatter": (G.gather_weight_scatter.default, (g["si"], g["di"], rg(g["w"]), rg(g["x"]))),
        "gather_scatter_impl": (G.gather_scatter_impl.default, (g["si"], g["di"], g["x"])),
        "gather_weight_scatter_impl": (G.gather_weight_scatter_impl.default, (g["si"], g["di"], g["w"], g["x"])),
        "gather_reduce": (G.gather_reduce.default, (g["si"], g["di"], g["w"], g["x"], "max")),
        "gather_reduce_mean_grad": (G.gather_reduce.default, (g["si"], g["di"], rg(g["w"]), rg(g["x"]), "mean")),
        "gather_reduce_max_grad": (G.gather_reduce.default, (g["si"], g["di"], rg(g["w"]), rg(g["x"]), "max")),
        "mh_spmm_grad": (G.mh_spmm.default, (g["si"], g["di"], rg(g["wh"]), rg(g["xh"]), "sum")),
        "mh_sddmm": (G.mh_sddmm.default, (g["si"], g["di"], g["xh"], g["xh"], False)),
        "gather_scatter_rows": (G.gather_scatter_rows.default, (g["si"], g["di"], rg(g["x"]), g["n"])),
        "gather_weight_scatter_rows": (G.gather_weight_scatter_rows.default, (g["si"], g["di"], rg(g["w"]), rg(g["x"]), g["n"])),
        "mh_spmm": (G.mh_spmm.default, (g["si"], g["di"], g["wh"], g["xh"], "sum")),
        "mh_spmm_rows": (G.mh_spmm_rows.default, (g["si"], g["di"], g["wh"], g["xh"], g["n"])),
        "sddmm_coo_impl": (G.sddmm_coo_impl.default, (g["si"].int(), g["di"].int(), g["x"], g["x"])),
        "csr_gws": (G.csr_gws.default, (g["rowptr"], g["si"], g["w"], g["x"])),
        "csr_gws_impl": (G.csr_gws_impl.default, (g["rowptr"].int(), g["si"].int(), g["w"], g["x"])),
        "gather_rows": (G.gather_rows.default, (g["di"], g["x"])),
        "transpose_edges": (G.transpose_edges.default, (g["si"], g["di"])),
        "transposed_weight": (G.transposed_weight.default, (g["si"], g["di"], g["w"])),
        "transpose_edges_weighted": (G.transpose_edges_weighted.default, (g["si"], g["di"], g["w"])),
        "coo_to_csr": (G.coo_to_csr.default, (g["di"],)),
    }


NAMES = ["index_scatter", "index_scatter_grad", "index_scatter_mean", "gather_scatter", "gather_weight_scatter", "gather_scatter_impl",
         "gather_weight_scatter_impl", "gather_reduce", "gather_reduce_mean_grad", "gather_reduce_max_grad", "mh_spmm_grad", "mh_sddmm", "gather_scatter_rows", "gather_weight_scatter_rows", "mh_spmm", "mh_spmm_rows",
         "sddmm_coo_impl", "csr_gws", "csr_gws_impl", "gather_rows", "transpose_edges", "transposed_weight", "transpose_edges_weighted",
         "coo_to_csr"]


STATIC_SHAPE = {"mh_sddmm", "gather_scatter_rows", "gather_weight_scatter_rows", "mh_spmm_rows", "sddmm_coo_impl", "gather_rows", "transpose_edges",
                "transposed_weight", "transpose_edges_weighted", "csr_gws_impl"}


@pytest.mark.parametrize("name", NAMES)
def test_opcheck(graph, name):
    op, args = cases(graph)[name]
    utils = UTILS if name in STATIC_SHAPE else tuple(u for u in UTILS if u != "test_aot_dispatch_static")
    opcheck(op, args, test_utils=utils)
