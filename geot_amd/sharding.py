"""Edge-list sharding across GPUs (one process per GPU, torch.distributed; backend "nccl" is RCCL).

The reference is single-GPU (no collective call site anywhere in its tree); this is the
multi-GPU form of the same path.  Segments are independent, so the dst-sorted edge list shards
into contiguous edge ranges, one per rank, and the output stays ROW-SHARDED:

    rank r holds edges [e_r, e_{r+1})  ->  owns dst rows (last_key_{r-1}, last_key_r]

Two ways to cut:
  * ``segment_aligned_cuts``  snaps every cut to a segment start -> no data-path collective;
  * ``equal_edge_cuts``       exact edge balance -> a segment may straddle a cut; each rank's
    partial first row is then exchanged with ONE small collective (all_gather of W rows of F
    values + 2 keys - latency-bound, microseconds over xGMI) and added by the owner in rank order
    (deterministic).  A hub that spans several ranks is handled by the same pass.

``local_op(index_local, src_local, rows) -> [rows, F]`` is the per-rank reduction; by default the
HIP operator.  Tests inject a CPU function to exercise the exchange logic under gloo.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def equal_edge_cuts(nnz: int, world: int) -> List[int]:
    """Edge offsets e_0..e_W with |e_{r+1}-e_r| differing by at most 1."""
    return [(nnz * r) // world for r in range(world + 1)]


def segment_aligned_cuts(index: torch.Tensor, world: int) -> List[int]:
    """Equal-edge cuts moved to the nearest segment start at or after them (index ascending)."""
    nnz = index.numel()
    cuts = equal_edge_cuts(nnz, world)
    out = [0]
    for r in range(1, world):
        e = max(cuts[r], out[-1])
        if 0 < e < nnz:
            key = index[e]
            if index[e - 1] == key:  # inside a segment: move to the first edge of the next key
                e = int(torch.searchsorted(index, key, right=True).item())
        out.append(min(e, nnz))
    out.append(nnz)
    return out


def _default_local_op(index_local: torch.Tensor, src_local: torch.Tensor, rows: int) -> torch.Tensor:
    from . import hip
    out = torch.empty((rows,) + tuple(src_local.shape[1:]), dtype=src_local.dtype, device=src_local.device)
    return hip.index_scatter_out(index_local, src_local.contiguous(), out, sorted=True)


def sharded_index_scatter(index_shard: torch.Tensor, src_shard: torch.Tensor,
                          group: Optional[dist.ProcessGroup] = None,
                          local_op: Optional[Callable] = None,
                          exchange: bool = True,
                          key_offset: Optional[int] = None) -> Tuple[torch.Tensor, int]:
    """Row-sharded index_scatter over the ranks of ``group``.

    ``index_shard`` / ``src_shard`` are this rank's contiguous slice of the globally dst-sorted
    edge list (rank order = edge order; every rank holds at least one edge).  Returns
    ``(out_rows, first_row)``: this rank's rows of the global result and the global row number of
    its first row.  Concatenating the ranks' ``out_rows`` in rank order gives exactly
    ``index_scatter(0, src, index)`` of the unsharded problem.

    ``exchange=False`` asserts the cuts are segment-aligned (no key is shared by two ranks) and
    skips the collective.

    ``key_offset``: the shard's index is already rank-local and its first key is 0
    (global key = local key + key_offset); saves the pass that re-bases the keys.
    """
    local_op = local_op or _default_local_op
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if index_shard.numel() == 0:
        raise ValueError("every rank must hold at least one edge")
    feat_shape = tuple(src_shard.shape[1:])
    F = int(src_shard[0].numel())
    dev = src_shard.device

    # local reduction on keys shifted to start at 0: rows [first_key, last_key]
    if key_offset is None:
        ends = torch.stack([index_shard[0], index_shard[-1]]).cpu()  # one D2H sync, as index[-1].item()
        first_key, last_key = int(ends[0]), int(ends[1])
        local = local_op(index_shard - first_key, src_shard, last_key - first_key + 1)
    else:
        local_rows = int(index_shard[-1].item()) + 1               # the operator's row rule
        first_key, last_key = key_offset, key_offset + local_rows - 1
        local = local_op(index_shard, src_shard, local_rows)

    if world == 1:
        if first_key > 0:
            local = torch.cat([local.new_zeros((first_key,) + feat_shape), local])
        return local, 0

    # boundary record: [first_key, last_key, first_row(F)] per rank, gathered by everyone
    rec = torch.empty(2 + F, dtype=torch.float64, device=dev)
    rec[0], rec[1] = first_key, last_key
    rec[2:] = local[0].reshape(-1).to(torch.float64)
    # RCCL ("nccl") moves device tensors directly over xGMI; a gloo group (CPU tests, or two test
    # ranks sharing one GPU) stages the few hundred bytes through the host
    via_host = dev.type == "cuda" and dist.get_backend(group) == "gloo"
    cdev = torch.device("cpu") if via_host else dev
    send = rec.to(cdev) if exchange else rec[:2].contiguous().to(cdev)
    recv = torch.empty(world * send.numel(), dtype=torch.float64, device=cdev)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.to(dev).view(world, -1)
    allrec = torch.zeros(world, 2 + F, dtype=torch.float64, device=dev)
    allrec[:, : recv.shape[1]] = recv
    keys_host = allrec[:, :2].to(torch.int64).cpu()
    firsts = [int(k) for k in keys_host[:, 0]]
    lasts = [int(k) for k in keys_host[:, 1]]

    # a rank owns key k if it is the lowest rank holding edges of k
    owns_first = rank == 0 or lasts[rank - 1] != first_key
    if not exchange and not owns_first:
        raise RuntimeError("exchange=False but a segment straddles the cut between ranks "
                           f"{rank - 1} and {rank}")
    # add the first-row partials of the following ranks that continue my last key, in rank order
    if exchange:
        r2 = rank + 1
        while r2 < world and firsts[r2] == last_key:
            local[-1] += allrec[r2, 2:].to(local.dtype).view(feat_shape)
            if lasts[r2] != last_key:
                break  # that rank has further keys: the run ends inside it
            r2 += 1
    # my rows: (last_key_{rank-1}, last_key]; drop a first row owned by a lower rank,
    # prepend zero rows for the empty keys between the previous rank's last key and my first key
    prev_last = lasts[rank - 1] if rank > 0 else -1
    if not owns_first:
        local = local[1:]
        first_row = first_key + 1
    else:
        gap = first_key - (prev_last + 1)
        if gap > 0:
            local = torch.cat([local.new_zeros((gap,) + feat_shape), local])
        first_row = prev_last + 1
    return local, first_row


def sharded_gather_scatter(src_index_shard: torch.Tensor, dst_index_shard: torch.Tensor,
                           src: torch.Tensor, weight_shard: Optional[torch.Tensor] = None,
                           group: Optional[dist.ProcessGroup] = None,
                           local_op: Optional[Callable] = None, exchange: bool = True,
                           key_offset: Optional[int] = None) -> Tuple[torch.Tensor, int]:
    """Row-sharded gather_scatter / gather_weight_scatter (BASELINE.json configs[4]).

    The edge list (src_index, dst_index[, weight]) is sharded by contiguous dst-sorted edge ranges
    exactly like :func:`sharded_index_scatter`; ``src`` (node features) is REPLICATED on every rank
    (56.9 GB per GPU at papers100M scale fits 288 GB).  Same boundary-row
    exchange, same return value.  ``local_op(src_index, dst_index_local, weight, src, rows)``
    defaults to the HIP operators.
    """
    if local_op is None:
        from . import hip

        def local_op(si, di, w, x, rows):
            out = torch.empty((rows, x.shape[1]), dtype=x.dtype, device=x.device)
            if w is None:
                return hip.gather_scatter_out(si.contiguous(), di, x, out)
            return hip.gather_weight_scatter_out(si.contiguous(), di, w.contiguous(), x, out)

    def as_index_scatter(index_local, _unused, rows):
        return local_op(src_index_shard, index_local, weight_shard, src, rows)

    # the per-edge operand is only used for its feature shape: hand over one row of src
    proto = src[:1].expand(dst_index_shard.numel(), *src.shape[1:])
    return sharded_index_scatter(dst_index_shard, proto, group=group, local_op=as_index_scatter,
                                 exchange=exchange, key_offset=key_offset)


def shard_edges(index: torch.Tensor, src: torch.Tensor, world: int, rank: int, aligned: bool = False):
    """Slice a replicated (index, src) pair for ``rank`` (helper for tests and examples)."""
    cuts = segment_aligned_cuts(index, world) if aligned else equal_edge_cuts(index.numel(), world)
    return index[cuts[rank]:cuts[rank + 1]], src[cuts[rank]:cuts[rank + 1]]
