"""torch.export + FX rewrite: message passing written with aten ops -> geot fused ops.

Counterpart of the reference's ``geot.match_replace.pattern_transform``
(geot/match_replace/match_replace.py:8-32 with fused_gs.py, fused_gws.py, fused_mh_spmm.py): export the
model, find

    dst.index_add(0, row, x.index_select(0, col))                       -> gather_scatter
    dst.index_add(0, row, x.index_select(0, col) * w.unsqueeze(-1))     -> gather_weight_scatter
    dst3d.index_add(0, row, x3d.index_select(0, col) * w2d.unsqueeze(-1))  -> mh_spmm

and replace each with one fused op.  Written from scratch as a local dataflow match on the
``index_add`` node (the reference walks the whole graph and keys on "the last select/index_select seen"),
with these guarantees the reference's pass does not give:

* only ``index_add`` into a tensor of ZEROS (new_zeros / zeros / zeros_like / full(0)) is rewritten -
  anything else is left untouched, so the rewrite never changes results;
* the fused node keeps the row count of the ``index_add`` it replaces (``dst.shape[0]``) through the
  ``geot::*_rows`` ops; the reference rewrites to ``csr_gws(coo_to_csr(row), ...)`` whose output has
  ``max(row) + 2`` rows;
* ``index_add`` is order-independent, the atomic-free kernels want ``row`` (the index_add index) ascending.
  PyG-style ``edge_index`` sorted by destination satisfies it and runs at full speed; any other order is
  still CORRECT: the ``geot::*_rows`` ops probe ``row`` once per content (``index_facts`` in geot_amd/csrc/host_cache.cpp)
  and reduce over its stable sort when it has descents.  ``sort_edges=True`` makes the pass insert that sort
  into the graph instead (useful when the exported program is to run somewhere the cache does not live).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.fx as fx
from torch.export import ExportedProgram, export

from . import ops  # noqa: F401  (registers torch.ops.geot.*)

aten = torch.ops.aten
_ZEROS = {aten.new_zeros.default, aten.zeros.default, aten.zeros_like.default}
_PASS_THROUGH = {aten.unsqueeze.default, aten.view.default, aten.reshape.default, aten._unsafe_view.default,
                 aten.expand.default, aten.alias.default, aten.contiguous.default, aten.clone.default}


def _is(node, target) -> bool:
    return isinstance(node, fx.Node) and node.op == "call_function" and node.target == target


def _is_zeros(node) -> bool:
    if not isinstance(node, fx.Node) or node.op != "call_function":
        return False
    if node.target in _ZEROS:
        return True
    if node.target in (aten.full.default, aten.new_full.default, aten.full_like.default):
        fill = node.args[-1] if node.target != aten.new_full.default else node.args[2]
        return isinstance(fill, (int, float)) and fill == 0
    return False


def _shape(node):
    val = node.meta.get("val") if isinstance(node, fx.Node) else None
    return tuple(val.shape) if val is not None else None


def _gathered_rows(node) -> Optional[Tuple[fx.Node, fx.Node]]:
    """node == x.index_select(0, col)  ->  (x, col)."""
    if _is(node, aten.index_select.default):
        x, dim, col = node.args[:3]
        nd = len(_shape(x) or ())
        if dim == 0 or (nd and dim == -nd):
            return x, col
    return None


def _edge_weight(node, nnz, want_dims) -> Optional[fx.Node]:
    """Strip unsqueeze/view wrappers from a per-edge factor; return the base tensor of shape
    [nnz] (want_dims == 1) or [nnz, H] (want_dims == 2), or None."""
    cur = node
    for _ in range(8):
        shp = _shape(cur)
        if shp is not None and len(shp) == want_dims and shp[0] == nnz:
            return cur
        if isinstance(cur, fx.Node) and cur.op == "call_function" and cur.target in _PASS_THROUGH:
            cur = cur.args[0]
            continue
        break
    return None


def _rows_arg(graph: fx.Graph, dst: fx.Node, before: fx.Node):
    r = _shape(dst)[0]
    if isinstance(r, int):
        return r
    with graph.inserting_before(before):
        return graph.call_function(aten.sym_size.int, (dst, 0))


def rewrite_graph(gm: fx.GraphModule, sort_edges: bool = False) -> int:
    """Rewrite every matching index_add in ``gm`` in place; returns the number of fused nodes."""
    graph = gm.graph
    fused = 0
    for node in list(graph.nodes):
        if not _is(node, aten.index_add.default) or len(node.args) < 4 or node.kwargs.get("alpha", 1) != 1:
            continue
        dst, dim, row, source = node.args[:4]
        dshape = _shape(dst)
        if dshape is None or not _is_zeros(dst) or not (dim == 0 or dim == -len(dshape)):
            continue
        nnz = (_shape(row) or (None,))[0]
        target = args = None
        g = _gathered_rows(source)
        if g is not None and len(dshape) == 2:
            x, col = g
            target, args, weight = torch.ops.geot.gather_scatter_rows.default, [col, row, x], None
        elif _is(source, aten.mul.Tensor):
            a, b = source.args
            for feat, fac in ((a, b), (b, a)):
                g = _gathered_rows(feat)
                if g is None:
                    continue
                x, col = g
                if len(dshape) == 2:
                    weight = _edge_weight(fac, nnz, 1)
                    target = torch.ops.geot.gather_weight_scatter_rows.default
                elif len(dshape) == 3:
                    weight = _edge_weight(fac, nnz, 2)
                    target = torch.ops.geot.mh_spmm_rows.default
                else:
                    weight = None
                if weight is not None:
                    args = [col, row, weight, x]
                    break
                target = None
        if target is None or args is None:
            continue
        rows = _rows_arg(graph, dst, node)
        with graph.inserting_before(node):
            if sort_edges:
                srt = graph.call_function(aten.sort.stable, (row,), {"stable": True})
                perm = graph.call_function(__import__("operator").getitem, (srt, 1))
                new_row = graph.call_function(__import__("operator").getitem, (srt, 0))
                col_s = graph.call_function(aten.index_select.default, (args[0], 0, perm))
                args[0], args[1] = col_s, new_row
                if len(args) == 4:
                    args[2] = graph.call_function(aten.index_select.default, (args[2], 0, perm))
            new = graph.call_function(target, tuple(args) + (rows,))
        new.meta["val"] = node.meta.get("val")
        node.replace_all_uses_with(new)
        graph.erase_node(node)
        fused += 1
    if fused:
        graph.eliminate_dead_code()
        graph.lint()
        gm.recompile()
    return fused


def pattern_transform(model: torch.nn.Module, args, sort_edges: bool = False, **kwargs) -> ExportedProgram:
    """Export ``model`` on ``args`` and fuse its message-passing patterns (same call shape as the
    reference's ``pattern_transform(model, args, **kwargs)``; extra keyword ``sort_edges``)."""
    exported = export(model, args, **kwargs)
    exported.geot_fused_nodes = rewrite_graph(exported.graph_module, sort_edges=sort_edges)
    return exported
