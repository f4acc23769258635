"""Launcher surface of the reference's comparison kernels (geot/triton/__init__.py:1-5), served by the
HIP engine.  No Triton is involved: the reference keeps five hand-written Triton launchers next to its
CUDA operators as baselines for its benchmark scripts (benchmark/bench_index_scatter.py:8-9,56-57,
benchmark/bench_spmm.py:3,66-68); those scripts call

    launch_parallel_reduction(indices, input, output, num_edges, feature_size, BLOCK_SIZE)
    launch_serial_reduction(edges, input, output, num_edges, feature_size, group_size)
    launch_pr_spmm(indices, input, output, num_edges, feature_size, BLOCK_SIZE)
    launch_sr_spmm(edges, input, output, num_edges, feature_size, group_size)
    launch_torch_compile_spmm(in0, in1, out, num_edges, feature_size, XBLOCK)

with a caller-allocated, usually zeroed `output` that the kernels ACCUMULATE into with atomic adds
(geot/triton/seg_reduction.py:38,67; geot/triton/spmm.py:39,71; geot/triton/torch_compile.py:20).  The
functions below keep names, argument order and that accumulate-into-`output` contract, so those scripts run
unchanged on MI355X; the tiling arguments (BLOCK_SIZE / group_size / XBLOCK) are accepted and ignored -
the tile shape is the library's own shape-keyed rule (`make_plan`, geot_amd/csrc/seg_reduce.hip).

 * the two `*_reduction` launchers are sorted-index segment sums  (output[index[e]] += input[e]);
 * the two `*_spmm` launchers take `indices` = a contiguous [2, num_edges] tensor, row 0 the gathered
   (source) node of each edge, row 1 the dst-sorted destination (geot/triton/spmm.py:27-31,62-66):
   output[indices[1, e]] += input[indices[0, e]];
 * `launch_torch_compile_spmm` is the order-agnostic variant (every element its own atomic add in the
   reference): served by a row gather followed by the unsorted (atomic) segment sum, so it stays correct for
   unsorted destinations.
"""
from __future__ import annotations

import torch

from . import hip

__all__ = ["launch_pr_spmm", "launch_parallel_reduction", "launch_sr_spmm", "launch_serial_reduction",
           "launch_torch_compile_spmm"]


def _check(output: torch.Tensor, feature_size: int) -> int:
    if not output.is_contiguous():
        raise ValueError("output must be contiguous")
    feature_size = int(feature_size)
    if feature_size <= 0 or output.numel() % feature_size:
        raise ValueError("output size is not a multiple of feature_size")
    return output.numel() // feature_size


def _segment_sum(index, input, output, num_edges, feature_size):
    rows = _check(output, feature_size)
    num_edges = int(num_edges)
    if num_edges == 0:
        return
    tmp = torch.empty((rows, int(feature_size)), dtype=output.dtype, device=output.device)
    hip.index_scatter_out(index.reshape(-1)[:num_edges], input.reshape(-1)[: num_edges * int(feature_size)],
                          tmp, True)
    output.view(rows, -1).add_(tmp)


def _gather_sum(indices, input, output, num_edges, feature_size):
    rows = _check(output, feature_size)
    num_edges = int(num_edges)
    if num_edges == 0:
        return
    flat = indices.reshape(-1)
    tmp = torch.empty((rows, int(feature_size)), dtype=output.dtype, device=output.device)
    hip.gather_scatter_out(flat[:num_edges], flat[num_edges: 2 * num_edges], input.reshape(-1, int(feature_size)), tmp)
    output.view(rows, -1).add_(tmp)


def launch_parallel_reduction(indices, input, output, num_edges, feature_size, BLOCK_SIZE):
    """geot/triton/seg_reduction.py:76-78 (segmented associative scan per feature, atomic add per run)."""
    _segment_sum(indices, input, output, num_edges, feature_size)


def launch_serial_reduction(edges, input, output, num_edges, feature_size, group_size):
    """geot/triton/seg_reduction.py:81-83 (sequential walk over `group_size` edges, atomic add per run)."""
    _segment_sum(edges, input, output, num_edges, feature_size)


def launch_pr_spmm(indices, input, output, num_edges, feature_size, BLOCK_SIZE):
    """geot/triton/spmm.py:78-80."""
    _gather_sum(indices, input, output, num_edges, feature_size)


def launch_sr_spmm(edges, input, output, num_edges, feature_size, group_size):
    """geot/triton/spmm.py:83-85."""
    _gather_sum(edges, input, output, num_edges, feature_size)


def launch_torch_compile_spmm(in0, in1, out, num_edges, feature_size, XBLOCK):
    """geot/triton/torch_compile.py:23-25: out[in0[1, e]] += in1[in0[0, e]], any edge order."""
    rows = _check(out, feature_size)
    num_edges = int(num_edges)
    if num_edges == 0:
        return
    F = int(feature_size)
    flat = in0.reshape(-1)
    msg = torch.empty((num_edges, F), dtype=out.dtype, device=out.device)
    hip.gather_rows_out(flat[:num_edges], in1.reshape(-1, F), msg)
    tmp = torch.empty((rows, F), dtype=out.dtype, device=out.device)
    hip.index_scatter_out(flat[num_edges: 2 * num_edges], msg, tmp, False)
    out.view(rows, -1).add_(tmp)
