"""One-time node renumbering for gathers on graphs that ship with UNORDERED node ids.

Not in the reference (its kernels gather wherever the rows lie).  Why it exists: the per-edge gather kernels serve a graph
whose sources sit near their destinations out of the XCDs' L2 (configs[2]-sized stand-in: 3.4 ms, fabric traffic 1.05x
compulsory), and the same graph with shuffled ids at the random-gather rate of the fabric (9.6-10.2 ms, 13.5x).  Public datasets
ship shuffled.  A device-built order that brings the members of a community together turns one into the other at the price
of two row permutations per call:

    x_new  = rows of x in the new order                       (2 x |x| bytes streamed, geot::gather_rows at ~7 TB/s)
    y_new  = gather_weight_scatter(new_src, new_dst, w_new, x_new)     # same kernels, renumbered + re-sorted edge list
    y      = rows of y_new back in the caller's order          (2 x |y| bytes)

Measured on MI355X (profiles/r04/exp_renumber.txt; block model at configs[2] size - 2.45 M nodes, 124 M edges, 222
communities of 2-20 k nodes, 90 % of a node's edges inside its community, ids shuffled, F=128): 10.19 ms as shipped ->
6.42 ms per call including both permutations (1.59x); the ceiling - the true community order - is 6.18 ms.  One-time cost:
label propagation ~0.3 s + renumber / re-sort 14 ms (device, torch ops).  A graph without community structure (uniform-random
sources) gains nothing: :func:`renumber` returns None for it.

The order is found by label propagation: every node repeatedly takes the most frequent label among its neighbours (sort-based,
synchronous, on the tensors' device).  Everything here is host logic over torch ops and the geot operators; gradients flow
(the permutations are differentiable row gathers, the operator in the middle is geot's own).
"""
from __future__ import annotations

from typing import Optional

import torch

__all__ = ["label_propagation", "rank_from_labels", "edge_locality", "RenumberedGraph", "renumber"]


def label_propagation(src_index: torch.Tensor, dst_index: torch.Tensor, num_nodes: int, sweeps: int = 10,
                      symmetric: bool = True, tol: float = 1e-3) -> torch.Tensor:
    """labels int64[num_nodes]: synchronous label propagation - label[v] <- the most frequent label among v's neighbours
    (ties: the larger label), starting from label[v] = v.  ``symmetric``: neighbours over both edge directions (a node
    without in-edges would otherwise keep its own label for ever).  Stops after ``sweeps`` sweeps or when fewer than
    ``tol`` of the nodes changed.  Sort-based: one sort of the (node, label) pairs per sweep, no atomics."""
    dev = dst_index.device
    if symmetric:
        a = torch.cat([dst_index, src_index])
        b = torch.cat([src_index, dst_index])
    else:
        a, b = dst_index, src_index
    label = torch.arange(num_nodes, device=dev, dtype=torch.int64)
    bits = max(1, int(num_nodes - 1).bit_length())
    if 2 * bits + 1 > 62:
        raise ValueError("label_propagation: too many nodes for the packed 64-bit sort key")
    mask = (1 << bits) - 1
    for _ in range(max(int(sweeps), 0)):
        key = torch.sort((a << bits) | label[b]).values
        uniq, cnt = torch.unique_consecutive(key, return_counts=True)
        del key
        score = (cnt << bits) | (uniq & mask)                   # most frequent first, larger label on ties
        best = torch.zeros(num_nodes, dtype=torch.int64, device=dev)
        best.scatter_reduce_(0, uniq >> bits, score, reduce="amax", include_self=True)
        new = torch.where(best > 0, best & mask, label)        # (a node without neighbours keeps its label)
        changed = int((new != label).sum().item())
        label = new
        if changed <= tol * num_nodes:
            break
    return label


def rank_from_labels(labels: torch.Tensor) -> torch.Tensor:
    """rank[v] = new id of node v: nodes ordered by (label, old id) - members of a label become neighbours."""
    order = torch.argsort(labels, stable=True)
    rank = torch.empty_like(order)
    rank[order] = torch.arange(order.numel(), device=order.device, dtype=order.dtype)
    return rank


def edge_locality(src_index: torch.Tensor, dst_index: torch.Tensor, window: int = 16384, sample: int = 4_000_000) -> float:
    """Fraction of (a sample of) the edges whose source lies within ``window`` rows of its destination - what the XCD-aware
    tile order of the gather kernels can turn into L2 hits."""
    n = dst_index.numel()
    if n == 0:
        return 1.0
    if n > sample:
        step = n // sample
        src_index, dst_index = src_index[::step], dst_index[::step]
    return float(((src_index - dst_index).abs() <= window).float().mean().item())


class _PermuteRows(torch.autograd.Function):
    """y[i] = x[idx[i]] for a PERMUTATION idx; the gradient is the same gather with the inverse permutation."""

    @staticmethod
    def forward(ctx, x, idx, inv):
        ctx.save_for_backward(inv)
        return torch.ops.geot.gather_rows(idx, x.contiguous())

    @staticmethod
    def backward(ctx, grad):
        (inv,) = ctx.saved_tensors
        return torch.ops.geot.gather_rows(inv, grad.contiguous()), None, None


class RenumberedGraph:
    """A dst-sorted edge list under a new node numbering.

    ``rank[v]`` = new id of node v (int64[num_nodes], a permutation).  Built once per graph: the edge list is renumbered and
    re-sorted by the new destination; per call the operands are permuted in, the geot operator runs on the new list, the
    result is permuted back.  The results equal the direct calls up to the order of the floating-point additions inside a
    row (the edges of a row are visited in another order).
    """

    def __init__(self, src_index: torch.Tensor, dst_index: torch.Tensor, num_nodes: int, rank: torch.Tensor):
        if rank.numel() != num_nodes or rank.dtype != torch.int64:
            raise ValueError("rank must be an int64 permutation of the node ids")
        if dst_index.numel() == 0:
            raise IndexError("index -1 is out of bounds for dimension 0 with size 0")
        self.num_nodes = int(num_nodes)
        self.rows = int(dst_index[-1].item()) + 1                 # the reference's row rule (csrc/gather_scatter.cpp:30)
        if self.rows > num_nodes:
            raise ValueError("dst_index names a row beyond num_nodes")
        self.rank = rank.contiguous()
        self.order = torch.argsort(self.rank)                     # new id -> old id
        new_dst, new_src = self.rank[dst_index], self.rank[src_index]
        self.edge_perm = torch.argsort(new_dst, stable=True)      # position in the new list -> original edge
        self.dst_index = new_dst[self.edge_perm].contiguous()
        self.src_index = new_src[self.edge_perm].contiguous()
        self.locality_before = edge_locality(src_index, dst_index)
        self.locality_after = edge_locality(self.src_index, self.dst_index)

    # ---- operands in / results out -------------------------------------------------------------------------------------
    def rows_in(self, x: torch.Tensor) -> torch.Tensor:
        """x [num_nodes, ...] in the new order."""
        if x.shape[0] != self.num_nodes:
            raise ValueError(f"expected {self.num_nodes} rows (one per node), got {x.shape[0]}")
        return _PermuteRows.apply(x, self.order, self.rank)

    def rows_out(self, y_new: torch.Tensor) -> torch.Tensor:
        """y_new [num_nodes, ...] back in the caller's order, cut to the row rule's ``rows``."""
        y = _PermuteRows.apply(y_new, self.rank, self.order)
        return y if self.rows == self.num_nodes else y[: self.rows]

    def edge_values(self, weight: torch.Tensor) -> torch.Tensor:
        """Per-edge values (weight [nnz] or [nnz, H]) in the new edge order.  For a STATIC weight (a normalised adjacency)
        call this once and pass the result with ``in_new_order=True``."""
        return weight.index_select(0, self.edge_perm)

    # ---- the operators --------------------------------------------------------------------------------------------------
    def gather_scatter(self, x: torch.Tensor) -> torch.Tensor:
        y = torch.ops.geot.gather_scatter_rows(self.src_index, self.dst_index, self.rows_in(x), self.num_nodes)
        return self.rows_out(y)

    def gather_weight_scatter(self, weight: torch.Tensor, x: torch.Tensor, in_new_order: bool = False) -> torch.Tensor:
        w = weight if in_new_order else self.edge_values(weight)
        y = torch.ops.geot.gather_weight_scatter_rows(self.src_index, self.dst_index, w, self.rows_in(x), self.num_nodes)
        return self.rows_out(y)

    # A multi-layer model can STAY in the new order: permute the input features in once (`rows_in`), run every layer on
    # (src_index, dst_index) of this object, permute the last layer's output back (`rows_out`) - per layer only the kernel is
    # paid (5.2 ms against 9.5-10.2 as shipped on the configs[2]-sized block model: 1.8-2.0x).
    def gather_weight_scatter_new_order(self, weight_new: torch.Tensor, x_new: torch.Tensor) -> torch.Tensor:
        """Operands and result in the NEW order (x_new = rows_in(x), weight_new = edge_values(w)); num_nodes rows out."""
        return torch.ops.geot.gather_weight_scatter_rows(self.src_index, self.dst_index, weight_new, x_new, self.num_nodes)

    def gather_scatter_new_order(self, x_new: torch.Tensor) -> torch.Tensor:
        return torch.ops.geot.gather_scatter_rows(self.src_index, self.dst_index, x_new, self.num_nodes)

    def mh_spmm(self, weight: torch.Tensor, x: torch.Tensor, in_new_order: bool = False) -> torch.Tensor:
        """weight [nnz, H] (edge-major), x [num_nodes, H, F]."""
        w = weight if in_new_order else self.edge_values(weight)
        y = torch.ops.geot.mh_spmm_rows(self.src_index, self.dst_index, w.contiguous(), self.rows_in(x), self.num_nodes)
        return self.rows_out(y)


def renumber(src_index: torch.Tensor, dst_index: torch.Tensor, num_nodes: int, sweeps: int = 10, symmetric: bool = True,
             min_gain: float = 0.25) -> Optional[RenumberedGraph]:
    """Label propagation + renumbering; None when the new order does not bring at least ``min_gain`` more of the edges within an
    L2-sized window of their destination than the shipped order does (a graph that is already ordered, or has no community
    structure to find: the permutations would be pure overhead)."""
    labels = label_propagation(src_index, dst_index, num_nodes, sweeps=sweeps, symmetric=symmetric)
    g = RenumberedGraph(src_index, dst_index, num_nodes, rank_from_labels(labels))
    return g if g.locality_after - g.locality_before >= min_gain else None
