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
from . import graph as _graph_ops  # noqa: F401  (registers geot::graph_spmm)

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


def _placeholder_values(exported: ExportedProgram, args) -> dict:
    """placeholder name -> concrete tensor, for the exported graph's lifted parameters / buffers / constants and the user's inputs."""
    import torch.utils._pytree as pytree
    from torch.export.graph_signature import InputKind
    flat_user = list(pytree.tree_leaves(tuple(args)))
    values, user_i = {}, 0
    for spec in exported.graph_signature.input_specs:
        name = getattr(spec.arg, "name", None)
        if spec.kind == InputKind.USER_INPUT:
            values[name] = flat_user[user_i]
            user_i += 1
        elif spec.kind == InputKind.PARAMETER:
            values[name] = exported.state_dict[spec.target]
        elif spec.kind == InputKind.BUFFER:
            values[name] = exported.state_dict[spec.target] if spec.target in exported.state_dict else exported.constants[spec.target]
        else:
            values[name] = exported.constants.get(spec.target)
    return values


def _evaluate(node, known: dict, depth: int = 0):
    """The concrete value of ``node`` on the example inputs: placeholders from ``known``, call_function nodes by calling their target on
    the evaluated arguments (the edge list of a layer is a select / slice / index of an input: a handful of nodes)."""
    if not isinstance(node, fx.Node):
        return node
    if node.name in known:
        return known[node.name]
    if node.op != "call_function" or depth > 32:
        raise RuntimeError(f"static_graph: cannot evaluate {node.format_node()} at export time")
    a = fx.node.map_arg(node.args, lambda n: _evaluate(n, known, depth + 1))
    k = fx.node.map_arg(node.kwargs, lambda n: _evaluate(n, known, depth + 1))
    known[node.name] = node.target(*a, **k)
    return known[node.name]


def bind_static_graphs(exported: ExportedProgram, args, slab_mode: str = "auto") -> list:
    """Second pass of ``pattern_transform(..., static_graph=True)``: every fused ``geot::*_rows(col, row, [w,] x, rows)`` node whose edge
    list can be evaluated on the example inputs becomes ``geot::graph_spmm(handle, w, x)`` over ONE ``geot_amd.Graph`` per distinct edge
    list, built HERE - outside ``forward``.  The rewritten program then never reads its index tensors for these nodes: no content
    fingerprint, no cache lookup, no read-back of the row rule per call (the host layer's guard re-reads 3.7 GB per call at configs[3]).
    THE CALLER'S PROMISE: the program is run on the graph it was exported with (a static graph - full-batch training, inference on one
    graph); other inputs of the same shapes would silently use the exported graph's edges.  Returns the handles
    (``geot_amd.graph.release_graph`` frees one)."""
    from . import graph as _graph
    gm = exported.graph_module
    known = _placeholder_values(exported, args)
    fused = {torch.ops.geot.gather_scatter_rows.default: False, torch.ops.geot.gather_weight_scatter_rows.default: True,
             torch.ops.geot.mh_spmm_rows.default: True}
    graphs, handles, bound = {}, [], 0
    for node in list(gm.graph.nodes):
        if node.op != "call_function" or node.target not in fused:
            continue
        has_w = fused[node.target]
        col, row = node.args[0], node.args[1]
        w = node.args[2] if has_w else None
        x, rows = node.args[-2], node.args[-1]
        xs = _shape(x)
        if not isinstance(rows, int) or xs is None or not isinstance(xs[0], int):
            continue                                       # (symbolic sizes: the node keeps its tensor-based form)
        key = (col.name if isinstance(col, fx.Node) else id(col), row.name if isinstance(row, fx.Node) else id(row), rows, xs[0])
        if key not in graphs:
            col_t, row_t = _evaluate(col, known), _evaluate(row, known)
            if not (torch.is_tensor(col_t) and col_t.is_cuda):
                raise RuntimeError("pattern_transform(static_graph=True): the example edge list must live on the GPU (geot_amd.Graph has no CPU path)")
            if row_t.numel() > 1 and bool((row_t[1:] < row_t[:-1]).any()):
                raise ValueError("pattern_transform(static_graph=True): the index_add index (edge_index[0]) must ascend - sort the edge list by "
                                 "destination once, or use sort_edges / the default tensor-based rewrite")
            g = _graph.Graph(col_t, row_t, num_src=xs[0], num_dst=rows, slab_mode=slab_mode)
            graphs[key] = _graph.register_graph(g)
            handles.append(graphs[key])
        with gm.graph.inserting_before(node):
            new = gm.graph.call_function(torch.ops.geot.graph_spmm.default, (graphs[key], w, x))
        new.meta["val"] = node.meta.get("val")
        node.replace_all_uses_with(new)
        gm.graph.erase_node(node)
        bound += 1
    if bound:
        gm.graph.eliminate_dead_code()
        gm.graph.lint()
        gm.recompile()
    exported.geot_static_nodes = bound
    return handles


def pattern_transform(model: torch.nn.Module, args, sort_edges: bool = False, static_graph: bool = False, slab_mode: str = "auto",
                      **kwargs) -> ExportedProgram:
    """Export ``model`` on ``args`` and fuse its message-passing patterns (same call shape as the
    reference's ``pattern_transform(model, args, **kwargs)``, geot/match_replace/match_replace.py:8-32; extra keywords ``sort_edges``,
    ``static_graph``).  ``static_graph=True``: the fused nodes are bound to ``geot_amd.Graph`` handles built here, once, from the example
    inputs' edge list (see :func:`bind_static_graphs` for the promise that goes with it); ``exported.geot_graphs`` lists the handles."""
    exported = export(model, args, **kwargs)
    exported.geot_fused_nodes = rewrite_graph(exported.graph_module, sort_edges=sort_edges and not static_graph)
    exported.geot_graphs = bind_static_graphs(exported, args, slab_mode=slab_mode) if static_graph else []
    return exported
