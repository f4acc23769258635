"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/geot_hip.h declares, the host layer reproduces the reference's schemas / checks / error
texts, and the product never touches the oracle.  No compute call is made (no GPU here).
"""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT

import geot_amd
from geot_amd import _lib, ops


def declared_functions():
    import glob
    text = "".join(open(h).read() for h in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))))
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(geot_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert "geot_index_scatter" in names and "geot_mh_spmm" in names and len(names) >= 15
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(_lib.SYMBOLS) == names            # the Python loader binds the same list
    assert L.geot_abi_version() == _lib.ABI_VERSION
    assert b"gfx950" in ctypes.cast(ctypes.CDLL(_lib.LIB_PATH).geot_build_info, ctypes.CFUNCTYPE(ctypes.c_char_p))()


def test_stable_header_has_no_experiment_switches():
    """a maintainer binding per INTEGRATION.md binds geot_hip.h; the tuning knobs and measurement hooks live
    in geot_hip_dev.h, and the ABI version moved with the plan contract."""
    stable = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "geot_hip.h")).read(), flags=re.S)
    for name in ("geot_tune", "geot_set_option", "geot_profile_enable", "geot_last_kernel"):
        assert name not in stable, name
    dev = open(os.path.join(ROOT, "include", "geot_hip_dev.h")).read()
    assert "geot_tune" in dev and "geot_set_option" in dev and "NOT part of the stable ABI" in dev
    assert "#define GEOT_ABI_VERSION 2" in open(os.path.join(ROOT, "include", "geot_hip.h")).read() and _lib.ABI_VERSION == 2


def test_product_library_carries_no_experiments():
    """the measured-and-rejected kernel variants (two rows per instruction, ...) and the timing probe that
    returns wrong results by design live in the DEVELOPMENT build only (libgeot_hip_dev.so, GEOT_HIP_LIB=dev).  The product exports
    none of their symbols and geot_set_option REFUSES their names (an unknown name is an error, never silently ignored)."""
    import subprocess
    from geot_amd import _lib, hip
    if _lib.DEV:
        pytest.skip("this process runs the development build")
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for needle in ("wpair", "slab_probe", "g_slab_pair", "g_slab_nt", "g_slab_wrow_all", "g_slab_tight", "g_slab_stage", "g_slab_unroll"):
        assert needle not in syms, needle
    assert "DEVELOPMENT" not in hip.build_info()
    for name in ("slab_probe", "slab_pair", "slab_wrow_all", "slab_nt", "slab_tight", "slab_stage", "slab_unroll", "no_such_switch"):
        with pytest.raises(RuntimeError, match="unknown option"):
            hip.set_option(name, 1)
    for name, value in (("slab_window", -2), ("slab_far", 12), ("slab_sddmm_mfma", 1), ("unroll", 0), ("xcd", 1)):
        hip.set_option(name, value)                                    # the product's own switches still answer


def test_library_is_gfx950_code_object():
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob
    # single offload target, no dual paths (rocPRIM's host-side tuning table names other archs as plain strings;
    # what counts is the set of code objects bundled in the library)
    import re
    assert set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob)) == {b"gfx950"} and b"sm_80" not in blob


def test_workspace_bytes_is_monotonic_upper_bound():
    L = _lib.load()
    a = L.geot_workspace_bytes(10_000_000, 64, 1_000_000, _lib.GEOT_F32)
    b = L.geot_workspace_bytes(20_000_000, 64, 1_000_000, _lib.GEOT_F32)
    c = L.geot_workspace_bytes(10_000_000, 64, 1_000_000, _lib.GEOT_F64)
    assert 0 < a < b and a < c
    assert L.geot_workspace_bytes(0, 1, 0, _lib.GEOT_F32) >= 256
    assert a < 32 << 20                                   # tight: the graded config needs ~11 MB of scratch
    assert L.geot_mh_workspace_bytes(1_000_000, 64, 32, 10_000, _lib.GEOT_F32) > L.geot_mh_workspace_bytes(1_000_000, 4, 32, 10_000, _lib.GEOT_F32) // 8


def test_schemas_match_reference_dispatcher():
    # csrc/index_scatter.cpp:44-46, csrc/gather_scatter.cpp:16-17, csrc/gather_weight_scatter.cpp:12-16,
    # csrc/mh_spmm.cpp:23 (inferred), geot/gather_scatter.py:7, geot/gather_weight_scatter.py:15
    expect = {
        "index_scatter": "geot::index_scatter(int dim, Tensor index, Tensor src, str reduce, bool sorted) -> Tensor",
        "gather_scatter_impl": "geot::gather_scatter_impl(Tensor src_index, Tensor dst_index, Tensor src) -> Tensor",
        "gather_weight_scatter_impl": "geot::gather_weight_scatter_impl(Tensor src_index, Tensor dst_index, Tensor weight, Tensor src) -> Tensor",
        "sddmm_coo_impl": "geot::sddmm_coo_impl(Tensor src_index, Tensor dst_index, Tensor mat_1, Tensor mat_2) -> Tensor",
        "mh_spmm": "geot::mh_spmm(Tensor src_index, Tensor dst_index, Tensor weight, Tensor src, str reduce) -> Tensor",
        "gather_scatter": "geot::gather_scatter(Tensor src_index, Tensor dst_index, Tensor src) -> Tensor",
        "gather_weight_scatter": "geot::gather_weight_scatter(Tensor src_index, Tensor dst_index, Tensor weight, Tensor src) -> Tensor",
    }
    for name, schema in expect.items():
        assert str(getattr(torch.ops.geot, name).default._schema) == schema


def test_public_surface_matches_reference_package():
    import inspect

    import geot
    for name in ("index_scatter", "gather_scatter", "gather_weight_scatter", "mh_spmm", "mh_spmm_transposed"):
        assert getattr(geot, name) is getattr(geot_amd, name)
    sig = inspect.signature(geot.index_scatter)
    assert list(sig.parameters) == ["dim", "src", "index", "reduce", "sorted"]          # geot/index_scatter.py:5-7
    assert sig.parameters["reduce"].default == "sum" and sig.parameters["sorted"].default is True
    assert list(inspect.signature(geot.mh_spmm).parameters) == ["src_index", "dst_index", "weight", "src", "reduce"]


def test_reduce_parsing_matches_reduceutils():
    for r, k in (("sum", "sum"), ("mean", "mean"), ("min", "min"), ("amin", "min"), ("max", "max"),
                 ("amax", "max"), ("prod", "prod")):
        assert ops.get_reduction_enum(r) == k
    with pytest.raises(RuntimeError, match="reduce argument must be either sum, prod, mean, amax or amin, got mul"):
        ops.get_reduction_enum("mul")


def test_argument_checks_carry_the_reference_texts():
    """The C++ host layer (geot_amd/csrc/torch_ops.cpp) runs the reference's argument checks before it looks at the
    device, so they can be exercised with CPU tensors here."""
    G = torch.ops.geot
    src = torch.rand(6, 4)
    idx = torch.tensor([0, 0, 1, 1, 2, 2])
    with pytest.raises(RuntimeError, match="dim must be non-negative and less than input dimensions"):
        G.index_scatter(2, idx, src, "sum", True)
    with pytest.raises(RuntimeError, match="dim must be non-negative and less than input dimensions"):
        G.index_scatter(-1, idx, src, "sum", True)
    with pytest.raises(RuntimeError, match="index must be 1 dimensional"):
        G.index_scatter(0, idx.view(2, 3), src, "sum", True)
    with pytest.raises(RuntimeError, match="index length must be equal to src dimension size"):
        G.index_scatter(0, idx[:5], src, "sum", True)
    with pytest.raises(RuntimeError, match="reduce argument must be either sum, prod, mean, amax or amin, got bogus"):
        G.index_scatter(0, idx, src, "bogus", True)
    with pytest.raises(NotImplementedError, match="only 'sum'"):
        G.mh_spmm(idx, idx, torch.rand(6, 2), src.view(6, 2, 2), "mean")
    assert ops._aggr_kind("add") == "sum" and ops._aggr_kind("amax") == "max"
    with pytest.raises(RuntimeError, match="reduce argument must be either"):
        G.gather_reduce(idx, idx, None, src, "median")
    with pytest.raises(RuntimeError, match="src_index and dst_index must be 1 dimensional"):
        G.gather_scatter_impl(idx.view(2, 3), idx, src)
    with pytest.raises(RuntimeError, match="src must be 2 dimensional"):
        G.gather_scatter_impl(idx, idx, src.view(6, 2, 2))
    with pytest.raises(RuntimeError, match="src must be 3 dimensional"):
        G.mh_spmm(idx, idx, torch.rand(6, 2), src, "sum")
    with pytest.raises(RuntimeError, match="Invalid weight size"):
        G.mh_spmm(idx, idx, torch.rand(5, 2), src.view(6, 2, 2), "sum")
    with pytest.raises(RuntimeError, match="weight must be 1 dimensional with one value per edge"):
        G.gather_weight_scatter_impl(idx, idx, torch.rand(5), src)
    with pytest.raises(IndexError):                       # index[-1] of an empty index, as in the reference
        G.index_scatter(0, idx[:0], src[:0], "sum", True)


def test_host_options_and_counters():
    """The host layer's switches (env at load, set_option at run time) and counters."""
    st = ops.stats()
    assert set(st) >= {"probes", "row_mismatches", "sorts", "transposes", "plans_built", "slab_calls", "facts"}
    old = ops.set_option("unsorted_mode", "atomic")
    assert ops.get_option("unsorted_mode") == 2 and ops.set_option("unsorted_mode", old) == 2
    old = ops.set_option("slab_mode", "always")
    assert ops.get_option("slab_mode") == 1
    ops.set_option("slab_mode", old)
    assert ops.get_option("speculate_rows") in (0, 1) and ops.get_option("trust_version") in (0, 1)
    with pytest.raises(RuntimeError, match="unknown host option"):
        ops.set_option("nope", 1)
    ops.clear_caches()
    assert ops.stats()["facts"] == 0
    # CPU tensors never reach the probe: the gather ops refuse them after the argument checks
    with pytest.raises(RuntimeError, match="CPU tensors are not supported"):
        torch.ops.geot.gather_scatter(torch.tensor([0, 1, 2, 3]), torch.tensor([0, 2, 1, 2]), torch.rand(4, 2))


def test_cpu_tensors_fail_loudly_no_fallback():
    """The gather ops have no CPU kernel in the reference either ([CUDA]-only registrations, csrc/gather_scatter.cpp:114-117):
    CPU tensors are refused, nothing falls back.  (index_scatter is the one op with a CPU key, see test_cpu_key.py.)"""
    src = torch.rand(6, 4)
    idx = torch.tensor([0, 0, 1, 1, 2, 2])
    for call in (lambda: geot_amd.gather_scatter(idx, idx, src),
                 lambda: geot_amd.gather_weight_scatter(idx, idx, torch.rand(6), src),
                 lambda: geot_amd.mh_spmm(idx, idx, torch.rand(6, 2), src.view(6, 2, 2))):
        with pytest.raises(RuntimeError, match="CPU tensors are not supported"):
            call()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        geot_amd.hip.index_scatter_out(idx, src, torch.empty(3, 4))


def test_fake_tensor_shape_rule():
    """register_fake returns [dynamic, F] like the reference (geot/gather_scatter.py:12-18)."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from torch.fx.experimental.symbolic_shapes import ShapeEnv
    with FakeTensorMode(shape_env=ShapeEnv(), allow_non_fake_inputs=False):
        si = torch.empty(50, dtype=torch.int64)
        src = torch.empty(20, 8)
        out = torch.ops.geot.gather_scatter(si, si, src)
        assert out.dim() == 2 and out.shape[1] == 8 and isinstance(out.shape[0], torch.SymInt)
        out = torch.ops.geot.gather_weight_scatter(si, si, torch.empty(50), src)
        assert out.shape[1] == 8 and out.dtype == torch.float32
        out = torch.ops.geot.index_scatter(0, si, torch.empty(50, 3, 2), "sum", True)
        assert tuple(out.shape[1:]) == (3, 2)


def test_product_never_touches_the_oracle():
    """geot_amd/ (and the alias package) must not import, link or mention oracle/ or a CPU fallback."""
    bad = []
    for pkg in ("geot_amd", "geot"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, pkg)):
            for f in files:
                if f.endswith((".py", ".hip", ".cpp", ".h")):
                    text = open(os.path.join(dirpath, f)).read()
                    if re.search(r"\boracle\b|libgeot_oracle|libgeot_ref", text):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad
    import sys
    assert not any(m == "oracle" or m.startswith("oracle.") for m in sys.modules
                   if getattr(sys.modules[m], "__file__", None) and "geot_amd" in (sys.modules[m].__file__ or ""))


def test_build_staleness_is_decided_by_content_not_by_file_times(tmp_path):
    """geot_amd/_lib.py: a snapshot copy of the tree (the GPU box gets one) keeps no file times; a binary with a stamp is stale
    exactly when the CONTENT of its sources differs from what it was built from, and the shipped binaries are not stale."""
    import time
    from geot_amd import _lib
    src, binary = tmp_path / "a.hip", tmp_path / "lib.so"
    src.write_text("v1")
    assert _lib._stale(str(binary), [str(src)])                       # no binary
    binary.write_text("built from v1")
    time.sleep(0.01)
    os.utime(src)                                                      # source looks newer
    assert _lib._stale(str(binary), [str(src)])                        # no stamp: file times decide
    (tmp_path / "lib.so.srchash").write_text(_lib._digest([str(src)]) + "\n")
    assert not _lib._stale(str(binary), [str(src)])                    # stamp: same content, whatever the times say
    src.write_text("v2")
    os.utime(binary)                                                   # binary looks newer
    assert _lib._stale(str(binary), [str(src)])                        # ... and other content is stale, whatever the times say
    assert not _lib.needs_build() and not _lib.plugin_needs_build()    # what travels with the tree is what the sources build
    assert set(_lib.LIB_INPUTS) >= set(_lib.SOURCES) and _lib.HEADER in _lib.LIB_INPUTS and _lib.HEADER in _lib.PLUGIN_INPUTS
