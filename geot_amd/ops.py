"""The geot.* operator surface -- same names, argument order, shape rule and error texts as the reference, so
PyG / torch_scatter style call sites are drop-in.

Layering, as in the reference (paths relative to its tree):

* dispatcher ops + host logic    geot_amd/csrc/torch_ops.cpp -> geot_amd/_C.so, loaded here with
                                 torch.ops.load_library like geot/__init__.py:12-19 loads the reference's `_C`
                                 (csrc/index_scatter.cpp:11-56, csrc/gather_scatter.cpp:13-34,
                                 csrc/gather_weight_scatter.cpp:11-49, csrc/mh_spmm.cpp:10-23, csrc/csr_gws.cpp:11-60);
* this file                      what the reference keeps in Python: the public functions (geot/index_scatter.py:5-8,
                                 geot/gather_scatter.py:3-39, geot/gather_weight_scatter.py:4-51, geot/mh_spmm.py:4-12,
                                 geot/csr_gws.py:3-37), the fake-tensor rules and the autograd formulas.

Ops in the dispatcher namespace ``geot`` (device key CUDA -- which is what a ROCm build of PyTorch calls the GPU):
    geot::index_scatter(int dim, Tensor index, Tensor src, str reduce, bool sorted) -> Tensor
    geot::gather_scatter_impl(Tensor src_index, Tensor dst_index, Tensor src) -> Tensor
    geot::gather_weight_scatter_impl(Tensor src_index, Tensor dst_index, Tensor weight, Tensor src) -> Tensor
    geot::sddmm_coo_impl(Tensor src_index, Tensor dst_index, Tensor mat_1, Tensor mat_2) -> Tensor
    geot::mh_spmm(Tensor src_index, Tensor dst_index, Tensor weight, Tensor src, str reduce) -> Tensor
    geot::csr_gws_impl(Tensor indptr, Tensor indices, Tensor weight, Tensor src) -> Tensor
    geot::gather_scatter / geot::gather_weight_scatter / geot::csr_gws      (the differentiable ops: defined HERE, in
                                                                             Python, as in the reference; kernels in _C.so)

Deliberate differences (all documented in DESIGN.md):
* the output row count still comes from ``index[-1] + 1``, read back from the device and checked on every call
  (part of the reference contract), but the read-back overlaps the kernels and the output is ``torch.empty``: the
  sorted kernels write every row exactly once, so the reference's ``torch::zeros`` pass is not needed;
* ``sorted`` is a promise the reference never checks (its "sorted" kernels flush with atomics and survive a wrong
  one): here every index is probed once per content and an index with descents is reduced over its stable sort -
  deterministic, every reduction and dtype; the gather ops accept an unsorted ``dst_index`` the same way;
* ``reduce``: sum / mean / min(amin) / max(amax) / prod with the semantics of the reference's CPU path
  (csrc/cpu/index_scatter_cpu.cpp:124-134); the reference's GPU kernels parse the argument but always add;
* ``dim != 0`` is honoured (the reference validates ``dim`` but always reduces along dim 0);
* CPU tensors: ``index_scatter`` has a CPU key as in the reference (csrc/index_scatter.cpp:53; the plugin's own kernel,
  intended operand); every other operator is GPU-only, as in the reference, and refuses CPU tensors.  GPU tensors are
  served by the HIP kernels or not at all - there is no fallback.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _lib, hip

if not os.path.exists(_lib.PLUGIN_PATH):
    raise ImportError(f"Could not find module '_C' in {os.path.dirname(_lib.PLUGIN_PATH)}: build it with `make shim` (or "
                      "`python -c 'import __graft_entry__ as g; g.build()'`).  geot_amd has no fallback path.")
torch.ops.load_library(_lib.PLUGIN_PATH)

# The three differentiable operators are DEFINED in Python, where the reference defines them (torch.library.custom_op in
# geot/gather_scatter.py:7, geot/gather_weight_scatter.py:15, geot/csr_gws.py:25) - `_C.so` defines exactly what the
# reference's csrc/*.cpp define and nothing under these names, so the reference's unmodified wrappers load on it as well
# (INTEGRATION.md).  Only the schema lives here: `_C.so` registers the C++ kernels for the device keys of these names
# (TORCH_LIBRARY_IMPL), so a call goes from the dispatcher straight into the host layer - no Python hop.
PUBLIC_SCHEMAS = {
    "gather_scatter": "gather_scatter(Tensor src_index, Tensor dst_index, Tensor src) -> Tensor",
    "gather_weight_scatter": "gather_weight_scatter(Tensor src_index, Tensor dst_index, Tensor weight, Tensor src) -> Tensor",
    "csr_gws": "csr_gws(Tensor csrptr, Tensor csrind, Tensor weight, Tensor src) -> Tensor",
}
_fragment = torch.library.Library("geot", "FRAGMENT")


def _has_schema(name: str) -> bool:
    try:
        torch._C._dispatch_find_schema_or_throw(f"geot::{name}", "")
        return True
    except RuntimeError:
        return False


#: names some OTHER Python layer defined before this module was imported (the reference's wrappers in the same process):
#: their fake / autograd rules are that layer's, not ours
_FOREIGN = {name for name in PUBLIC_SCHEMAS if _has_schema(name)}
for _name, _schema in PUBLIC_SCHEMAS.items():
    if _name not in _FOREIGN:
        _fragment.define(_schema)

_REDUCE_ENUM = {"max": "max", "amax": "max", "mean": "mean", "min": "min", "amin": "min",
                "sum": "sum", "prod": "prod"}


def get_reduction_enum(reduce: str) -> str:
    """csrc/reduceutils.h:5-22 -- same accepted spellings, same error text (the C++ host layer has the same table)."""
    if reduce in _REDUCE_ENUM:
        return _REDUCE_ENUM[reduce]
    raise RuntimeError(
        f"reduce argument must be either sum, prod, mean, amax or amin, got {reduce}")


def _aggr_kind(reduce: str) -> str:
    """`reduce` of the gather ops as PyG call sites pass it (models/conv/gcnconv.py:259 forwards self.aggr):
    the reference's spellings plus PyG's 'add'."""
    return "sum" if reduce == "add" else get_reduction_enum(reduce)


# ---- host-layer switches and counters (geot_amd/csrc/host_state.cpp, host_cache.cpp) -----------------------------------------------
_KEEP = -(2 ** 63)
_UNSORTED = {"auto": 0, "sort": 1, "atomic": 2}
_SLAB = {"never": -1, "auto": 0, "always": 1}


def set_option(name: str, value) -> int:
    """speculate_rows (0|1), trust_version (0|1), unsorted_mode ('auto'|'sort'|'atomic'), slab_mode
    ('never'|'auto'|'always'), transpose_cache (entries), slab_keep (plans), cache_mb (byte budget of all cached
    artefacts together, MiB; 0 = 1/8 of the device's memory), slab_builder (0 device | 1 ATen), content_guard (0|1: every use
    of a remembered product re-reads the tensors it was derived from and compares fingerprints; a write behind the version
    counter then costs one repeated call instead of a wrong result).  Returns the previous value."""
    if name == "unsorted_mode" and isinstance(value, str):
        value = _UNSORTED[value]
    if name == "slab_mode" and isinstance(value, str):
        value = _SLAB[value]
    return int(torch.ops.geot._host_option(name, int(value)))


def get_option(name: str) -> int:
    return int(torch.ops.geot._host_option(name, _KEEP))


def clear_caches() -> None:
    """Drop the remembered index facts, transposed edge lists and slab plans."""
    torch.ops.geot._host_option("clear_caches", 0)


def _at_exit() -> None:
    """Drop what the host layer remembers while torch is still whole: tensors freed by static destructors at process exit - after
    captured graphs, streams or the allocator may be half gone - were seen to crash the exit (SIGSEGV with two CUDAGraphs alive)."""
    try:
        clear_caches()
        from . import hip
        hip.release_workspaces()
    except Exception:  # noqa: BLE001  (interpreter shutdown: nothing left to do)
        pass


import atexit as _atexit  # noqa: E402

_atexit.register(_at_exit)


def stats() -> dict:
    names = ("probes", "row_mismatches", "sorts", "transposes", "plans_built", "slab_calls", "plan_us", "facts", "transposed", "plans",
             "published", "alarms", "cache_bytes", "stale_products", "guard_checks", "plan_trials", "plans_rejected",
             "trial_plan_us", "trial_edges_us", "plans_declined", "last_coverage_permille")
    return dict(zip(names, torch.ops.geot._host_stats()))


# --------------------------------------------------------------------------------------------------
# fake-tensor rules (geot/gather_scatter.py:12-18 etc.: [dynamic rows, F])
# --------------------------------------------------------------------------------------------------
def _fake_rows(src_like: torch.Tensor, tail_shape) -> torch.Tensor:
    ctx = torch.library.get_ctx()
    rows = ctx.new_dynamic_size()
    return src_like.new_empty([rows, *tail_shape])


@torch.library.register_fake("geot::index_scatter")
def _(dim, index, src, reduce, sorted):
    ctx = torch.library.get_ctx()
    shape = list(src.shape)
    shape[dim] = ctx.new_dynamic_size()
    return src.new_empty(shape)


@torch.library.register_fake("geot::gather_scatter_impl")
def _(src_index, dst_index, src):
    return _fake_rows(src, [src.shape[1]])


@torch.library.register_fake("geot::gather_weight_scatter_impl")
def _(src_index, dst_index, weight, src):
    return _fake_rows(src, [src.shape[1]])


def _fake_gather_scatter(src_index, dst_index, src):
    return _fake_rows(src, [src.shape[1]])


if "gather_scatter" not in _FOREIGN:
    torch.library.register_fake("geot::gather_scatter")(_fake_gather_scatter)


def _fake_gather_weight_scatter(src_index, dst_index, weight, src):
    return _fake_rows(src, [src.shape[1]])


if "gather_weight_scatter" not in _FOREIGN:
    torch.library.register_fake("geot::gather_weight_scatter")(_fake_gather_weight_scatter)


@torch.library.register_fake("geot::gather_reduce")
def _(src_index, dst_index, weight, src, reduce):
    return _fake_rows(src, [src.shape[1]])


@torch.library.register_fake("geot::sddmm_coo_impl")
def _(src_index, dst_index, mat_1, mat_2):
    return mat_1.new_empty([dst_index.shape[0]])


@torch.library.register_fake("geot::csr_gws_impl")
def _(indptr, indices, weight, src):
    return src.new_empty([indptr.shape[0], src.shape[1]])


def _fake_csr_gws(csrptr, csrind, weight, src):
    ctx = torch.library.get_ctx()
    return src.new_empty([ctx.new_dynamic_size(), src.shape[1]])


if "csr_gws" not in _FOREIGN:
    torch.library.register_fake("geot::csr_gws")(_fake_csr_gws)


@torch.library.register_fake("geot::mh_spmm")
def _(src_index, dst_index, weight, src, reduce):
    return _fake_rows(src, [src.shape[1], src.shape[2]])


@torch.library.register_fake("geot::gather_scatter_rows")
def _(src_index, dst_index, src, rows):
    return src.new_empty([rows, src.shape[1]])


@torch.library.register_fake("geot::gather_weight_scatter_rows")
def _(src_index, dst_index, weight, src, rows):
    return src.new_empty([rows, src.shape[1]])


@torch.library.register_fake("geot::mh_spmm_rows")
def _(src_index, dst_index, weight, src, rows):
    return src.new_empty([rows, src.shape[1], src.shape[2]])


@torch.library.register_fake("geot::mh_sddmm")
def _(src_index, dst_index, mat_1, mat_2, head_major):
    nnz, heads = dst_index.shape[0], mat_1.shape[1]
    return mat_1.new_empty([heads, nnz] if head_major else [nnz, heads])


@torch.library.register_fake("geot::gather_select_backward")
def _(src_index, dst_index, weight, src, out, grad):
    return src.new_empty(src.shape), src.new_empty([dst_index.shape[0] if weight is not None else 0])


@torch.library.register_fake("geot::gather_rows")
def _(index, src):
    return src.new_empty([index.shape[0], *src.shape[1:]])


@torch.library.register_fake("geot::transposed_weight")
def _(src_index, dst_index, weight):
    return weight.new_empty(weight.shape)


@torch.library.register_fake("geot::transpose_edges_weighted")
def _(src_index, dst_index, weight):
    return (src_index.new_empty(src_index.shape), src_index.new_empty(src_index.shape), dst_index.new_empty(dst_index.shape),
            weight.new_empty(weight.shape))


@torch.library.register_fake("geot::transpose_edges")
def _(src_index, dst_index):
    return src_index.new_empty(src_index.shape), src_index.new_empty(src_index.shape), dst_index.new_empty(dst_index.shape)


# --------------------------------------------------------------------------------------------------
# autograd (geot/gather_scatter.py:21-39, geot/gather_weight_scatter.py:31-51)
# --------------------------------------------------------------------------------------------------
def _sorted_by_source(src_index, dst_index):
    """(perm, edges' sources ascending, their destinations): the transposed edge list.  The reference re-sorts on every
    backward call (geot/gather_scatter.py:30-33); the host layer keeps it per edge-list content (an opaque dispatcher
    op, so torch.compile / AOT autograd never trace into the cache)."""
    return torch.ops.geot.transpose_edges(src_index, dst_index)


def _gs_setup_context(ctx, inputs, output):
    src_index, dst_index, src = inputs[:3]
    ctx.save_for_backward(src_index, dst_index)
    ctx.src_rows = src.shape[0]


def _gs_backward(ctx, grad):
    """d/dsrc of gather_scatter = the same op on the transposed edge list (geot/gather_scatter.py:26-39).

    Unlike the reference the result has src.shape[0] rows even when the last source node has no
    out-edge (the reference returns max(src_index)+1 rows and autograd then rejects the shape).
    """
    src_index, dst_index = ctx.saved_tensors
    grad = grad.contiguous()
    _, dst_index_bwd, src_index_bwd = _sorted_by_source(src_index, dst_index)
    g = torch.ops.geot.gather_scatter_rows(src_index_bwd, dst_index_bwd, grad, ctx.src_rows)
    return None, None, g


def _gws_setup_context(ctx, inputs, output):
    src_index, dst_index, weight, src = inputs[:4]
    ctx.save_for_backward(src_index, dst_index, weight, src)


def _gws_backward(ctx, grad):
    """geot/gather_weight_scatter.py:36-51.

    d/dsrc   = gws on the transposed (source-sorted) edge list, as in the reference.
    d/dweight[e] = <grad[dst_index[e]], src[src_index[e]]> in ORIGINAL edge order.  The reference
    calls sddmm on the re-sorted lists with grad/src swapped, which returns the values in
    source-sorted order and against the wrong rows (verified against dense
    autograd); this is the mathematically correct gradient.
    """
    src_index, dst_index, weight, src = ctx.saved_tensors
    grad = grad.contiguous()
    # (only the gradients autograd asks for: a GCN's normalised adjacency does not require grad - no SDDMM then)
    need_w, need_src = ctx.needs_input_grad[2], ctx.needs_input_grad[3]
    src_grad = weight_grad = None
    if need_src:
        # the transposed edge list and the weights in its order, looked up in one operator call (kept by the host layer while
        # the tensors' content is unchanged; a weight that does not require grad - a normalised adjacency - is permuted once)
        _, dst_index_bwd, src_index_bwd, weight_bwd = torch.ops.geot.transpose_edges_weighted(src_index, dst_index, weight.detach())
        src_grad = torch.ops.geot.gather_weight_scatter_rows(src_index_bwd, dst_index_bwd, weight_bwd, grad,
                                                             src.shape[0])
    if need_w:
        weight_grad = torch.ops.geot.sddmm_coo_impl(src_index, dst_index, grad, src.detach())
    return None, None, weight_grad, src_grad


if "gather_scatter" not in _FOREIGN:
    torch.library.register_autograd("geot::gather_scatter", _gs_backward, setup_context=_gs_setup_context)
if "gather_weight_scatter" not in _FOREIGN:
    torch.library.register_autograd("geot::gather_weight_scatter", _gws_backward, setup_context=_gws_setup_context)
torch.library.register_autograd("geot::gather_scatter_rows", lambda ctx, grad: (*_gs_backward(ctx, grad), None),
                                setup_context=_gs_setup_context)
torch.library.register_autograd("geot::gather_weight_scatter_rows", lambda ctx, grad: (*_gws_backward(ctx, grad), None),
                                setup_context=_gws_setup_context)


def _gr_setup(ctx, inputs, output):
    src_index, dst_index, weight, src, reduce = inputs
    ctx.reduce = _aggr_kind(reduce)
    ctx.has_weight = weight is not None
    ctx.save_for_backward(src_index, dst_index, src, *((weight,) if weight is not None else ()), *((output,) if ctx.reduce in ("max", "min") else ()))


def _gr_backward(ctx, grad):
    """Backward of ``geot::gather_reduce`` (PyG's aggr= on the gather ops; the reference's GPU kernels ignore ``reduce`` and its
    wrappers differentiate the sum only, geot/gather_scatter.py:21-39, gather_weight_scatter.py:31-51).

    mean: out[d] = (1 / deg d) * sum_e w_e src[s_e]  ->  the SUM backward on grad / deg - d/dsrc over the cached transposed edge
    list, d/dweight by the SDDMM - with deg = edges per destination row (rows without edges keep gradient 0).
    max / min: the gradient of out[d, f] goes to the messages that attain it, divided evenly among ties (torch.scatter_reduce's rule;
    ``geot::gather_select_backward``: float32 / float64, sums into d/dsrc by float atomics).
    prod: no gradient here - refused loudly (as index_scatter does for its non-sum reductions) rather than a silent zero from
    autograd's fallback."""
    if ctx.reduce not in ("mean", "max", "min"):
        raise NotImplementedError(f"gather_scatter / gather_weight_scatter: backward is implemented for reduce='sum', 'mean', 'max' and 'min' only (got '{ctx.reduce}')")
    if ctx.reduce in ("max", "min"):
        src_index, dst_index, src, *rest = ctx.saved_tensors
        out = rest[-1]
        weight = rest[0] if ctx.has_weight else None
        gsrc, gw = torch.ops.geot.gather_select_backward(src_index, dst_index, None if weight is None else weight.detach(), src.detach(), out.detach(),
                                                         grad.contiguous())
        return None, None, (gw if ctx.has_weight and ctx.needs_input_grad[2] else None), (gsrc if ctx.needs_input_grad[3] else None), None
    src_index, dst_index, src, *rest = ctx.saved_tensors
    weight = rest[0] if rest else None
    rows = grad.shape[0]
    ones = torch.ones((dst_index.shape[0], 1), dtype=torch.float32, device=grad.device)
    deg = torch.ops.geot.index_scatter(0, dst_index, ones, "sum", True)          # [index[-1] + 1, 1] - the forward's row rule
    torch._check(deg.shape[0] == rows, lambda: "gather_reduce backward: the gradient's rows differ from the forward's (dst_index[-1] + 1)")
    g = (grad / deg.clamp_(min=1.0).to(grad.dtype)).contiguous()
    need_w = ctx.has_weight and ctx.needs_input_grad[2]
    need_src = ctx.needs_input_grad[3]
    src_grad = weight_grad = None
    if need_src:
        if weight is None:
            _, dst_index_bwd, src_index_bwd = _sorted_by_source(src_index, dst_index)
            src_grad = torch.ops.geot.gather_scatter_rows(src_index_bwd, dst_index_bwd, g, src.shape[0])
        else:
            _, dst_index_bwd, src_index_bwd, weight_bwd = torch.ops.geot.transpose_edges_weighted(src_index, dst_index, weight.detach())
            src_grad = torch.ops.geot.gather_weight_scatter_rows(src_index_bwd, dst_index_bwd, weight_bwd, g, src.shape[0])
    if need_w:
        weight_grad = torch.ops.geot.sddmm_coo_impl(src_index, dst_index, g, src.detach())
    return None, None, weight_grad, src_grad, None


torch.library.register_autograd("geot::gather_reduce", _gr_backward, setup_context=_gr_setup)


def _mh_setup(ctx, inputs, output):
    src_index, dst_index, weight, src = inputs[:4]
    ctx.save_for_backward(src_index, dst_index, weight, src)


def _mh_backward(ctx, grad):
    """Backward of ``geot::mh_spmm`` - the reference ships none (geot/mh_spmm.py:4-12), so a GAT layer rewritten onto it
    (geot/match_replace/fused_mh_spmm.py:4-50) is inference-only there.  Same pattern as gather_weight_scatter
    (geot/gather_weight_scatter.py:31-51), per head:

    d/dsrc[s, h, :]  = sum over the edges leaving s of w[e, h] * grad[d_e, h, :]  = mh_spmm over the transposed edge list with the
                       weights in its order (rows permuted by one row gather);
    d/dweight[e, h]  = <grad[d_e, h, :], src[s_e, h, :]>  = the multi-head SDDMM, in the weight's own layout and edge order."""
    src_index, dst_index, weight, src = ctx.saved_tensors
    grad = grad.contiguous()
    nnz = src_index.shape[0]
    head_major = not (weight.dim() == 2 and weight.shape[0] == nnz and weight.shape[1] == src.shape[1])   # (layout pick of mh_spmm_base.h:38-49)
    need_w, need_src = ctx.needs_input_grad[2], ctx.needs_input_grad[3]
    src_grad = weight_grad = None
    if need_src:
        perm, dst_index_bwd, src_index_bwd = _sorted_by_source(src_index, dst_index)
        w_em = weight.detach().t().contiguous() if head_major else weight.detach().contiguous()
        w_bwd = torch.ops.geot.gather_rows(perm, w_em)
        src_grad = torch.ops.geot.mh_spmm_rows(src_index_bwd, dst_index_bwd, w_bwd, grad, src.shape[0])
    if need_w:
        weight_grad = torch.ops.geot.mh_sddmm(src_index, dst_index, grad, src.detach(), head_major)
    return None, None, weight_grad, src_grad, None


torch.library.register_autograd("geot::mh_spmm", _mh_backward, setup_context=_mh_setup)
torch.library.register_autograd("geot::mh_spmm_rows", _mh_backward, setup_context=_mh_setup)


def _is_setup(ctx, inputs, output):
    dim, index, src, reduce, _sorted = inputs
    ctx.save_for_backward(index)
    ctx.dim, ctx.reduce = dim, reduce


def _is_backward(ctx, grad):
    """d/dsrc[e] = grad[index[e]]: the row gather the reference ships as gather_eb_sorted_kernel
    (csrc/cuda/index_scatter_kernel.cuh:266-315) but never wires to an op.  sum only, as in the reference."""
    if get_reduction_enum(ctx.reduce) != "sum":
        raise NotImplementedError(f"index_scatter: backward is implemented for reduce='sum' only (got '{ctx.reduce}')")
    (index,) = ctx.saved_tensors
    g = grad if ctx.dim == 0 else grad.movedim(ctx.dim, 0)
    out = torch.ops.geot.gather_rows(index, g.contiguous())
    return None, None, (out if ctx.dim == 0 else out.movedim(0, ctx.dim)), None, None


torch.library.register_autograd("geot::index_scatter", _is_backward, setup_context=_is_setup)


# --------------------------------------------------------------------------------------------------
# public functions (geot/*.py)
# --------------------------------------------------------------------------------------------------
def index_scatter(dim: int, src: torch.Tensor, index: torch.Tensor, reduce: str = "sum",
                  sorted: bool = True) -> torch.Tensor:
    """dst[index[i]] += src[i] along ``dim``; rows = index[-1] + 1 (geot/index_scatter.py:5-8).

    Mind the argument order: Python takes (dim, src, index, ...), the dispatcher op takes
    (dim, index, src, ...) -- exactly as in the reference.
    """
    return torch.ops.geot.index_scatter(dim, index, src, reduce, sorted)


def gather_scatter_impl(src_index: torch.Tensor, dst_index: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    return torch.ops.geot.gather_scatter_impl(src_index, dst_index, src)


def gather_weight_scatter_impl(src_index: torch.Tensor, dst_index: torch.Tensor, weight: torch.Tensor,
                               src: torch.Tensor) -> torch.Tensor:
    return torch.ops.geot.gather_weight_scatter_impl(src_index, dst_index, weight, src)


def sddmm_coo_impl(src_index: torch.Tensor, dst_index: torch.Tensor, mat_1: torch.Tensor,
                   mat_2: torch.Tensor) -> torch.Tensor:
    """out[e] = <mat_1[dst_index[e]], mat_2[src_index[e]]>  (geot/gather_weight_scatter.py:8-12)."""
    return torch.ops.geot.sddmm_coo_impl(src_index, dst_index, mat_1, mat_2)


def gather_scatter(src_index: torch.Tensor, dst_index: torch.Tensor, src: torch.Tensor,
                   reduce: str = "sum") -> torch.Tensor:
    """dst[dst_index[e]] += src[src_index[e]], dst_index ascending (geot/gather_scatter.py:7-9).

    The trailing ``reduce`` is what the reference's own callers pass (models/conv/spmm.py:5-8 forwards the
    layer's ``aggr``; test/test_gather_scatter.py:25): 'sum' / 'add' run the differentiable op of the
    reference; 'mean' / 'min' / 'max' / 'prod' aggregate the messages of every row ('mean', 'max' and 'min' are differentiable
    too; 'prod' refuses a backward pass loudly).
    """
    if _aggr_kind(reduce) == "sum":
        return torch.ops.geot.gather_scatter(src_index, dst_index, src)
    return torch.ops.geot.gather_reduce(src_index, dst_index, None, src, reduce)


def gather_weight_scatter(src_index: torch.Tensor, dst_index: torch.Tensor, weight: torch.Tensor,
                          src: torch.Tensor, reduce: str = "sum") -> torch.Tensor:
    """dst[dst_index[e]] += weight[e] * src[src_index[e]] (geot/gather_weight_scatter.py:15-18); ``reduce`` as in
    :func:`gather_scatter` (models/conv/spmm.py:10-14)."""
    if _aggr_kind(reduce) == "sum":
        return torch.ops.geot.gather_weight_scatter(src_index, dst_index, weight, src)
    return torch.ops.geot.gather_reduce(src_index, dst_index, weight, src, reduce)


def mh_spmm(src_index: torch.Tensor, dst_index: torch.Tensor, weight: torch.Tensor, src: torch.Tensor,
            reduce: str = "sum") -> torch.Tensor:
    """Multi-head weighted SpMM, src [N, H, F], weight [nnz, H] or [H, nnz] (geot/mh_spmm.py:4-6)."""
    return torch.ops.geot.mh_spmm(src_index, dst_index, weight, src, reduce)


def mh_spmm_transposed(src_index: torch.Tensor, dst_index: torch.Tensor, weight: torch.Tensor,
                       src: torch.Tensor, reduce: str = "sum") -> torch.Tensor:
    """geot/mh_spmm.py:8-12: transposes weight [nnz, H] -> [H, nnz] (contiguous) and calls mh_spmm."""
    weight = weight.transpose(0, 1).contiguous()
    return torch.ops.geot.mh_spmm(src_index, dst_index, weight, src, reduce)


# --------------------------------------------------------------------------------------------------
# CSR path: geot/csr_gws.py:3-37, geot/match_replace/format_transform.py:5-25
# --------------------------------------------------------------------------------------------------
def csr_gws_impl(csrptr: torch.Tensor, csrind: torch.Tensor, weight: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    return torch.ops.geot.csr_gws_impl(csrptr, csrind, weight, src)


def csr_gws(csrptr: torch.Tensor, csrind: torch.Tensor, weight: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    """CSR weighted SpMM: out[r] = sum_e weight[e] * src[csrind[e]] over the row's nonzeros
    (geot/csr_gws.py:25-28).  Output rows = csrptr.size(0) as in the reference."""
    return torch.ops.geot.csr_gws(csrptr, csrind, weight, src)


@torch.library.custom_op("geot::coo_to_csr", mutates_args=())
def coo_to_csr(coo_row: torch.Tensor) -> torch.Tensor:
    """COO row ids -> int32 CSR row pointers of length max(coo_row)+2
    (geot/match_replace/format_transform.py:5-18).  Any order of coo_row is accepted."""
    if coo_row.dim() != 1:
        raise RuntimeError("coo_row must be 1 dimensional")
    nrow = int(coo_row.max().item()) + 1
    rowptr = torch.empty(nrow + 1, dtype=torch.int32, device=coo_row.device)
    return hip.coo_to_csr_out(coo_row.to(torch.int64).contiguous(), rowptr, assume_sorted=False)


@torch.library.register_fake("geot::coo_to_csr")
def _(coo_row):
    ctx = torch.library.get_ctx()
    return coo_row.new_empty([ctx.new_dynamic_size()], dtype=torch.int32)
