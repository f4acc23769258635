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


# (first_key, last_key) of THIS rank's shard, remembered per index identity: GNN edge lists are static, so
# the local row count is almost always the same as last time.  It is only a GUESS for launching the local
# kernels early: the true keys are read back underneath them and verified before anything is sent.
_ends_seen: dict = {}
_tls = __import__("threading").local()


def _pinned_slot():
    slot = getattr(_tls, "slot", None)
    if slot is None:
        slot = _tls.slot = (torch.empty(2, dtype=torch.int64).pin_memory(), torch.cuda.Event())
    return slot


def _ident(index: torch.Tensor, world: int, rank: int, key_offset):
    try:
        version = index._version
    except RuntimeError:      # inference tensors keep no version counter: never remembered
        return None
    return (index.device.type, index.device.index, index.data_ptr(), index.numel(), version, world, rank, key_offset)


def _apply_boundaries(local, head, allrows, firsts, lasts, rank, world, feat_shape, exchange):
    """Ownership rules given every rank's (first_key, last_key): returns (rows, first_row)."""
    first_key, last_key = firsts[rank], lasts[rank]
    owns_first = rank == 0 or lasts[rank - 1] != first_key
    if not exchange and not owns_first:
        raise RuntimeError("exchange=False but a segment straddles the cut between ranks "
                           f"{rank - 1} and {rank}")
    if exchange:
        # add the first-row partials of the following ranks that continue my last key, in rank order
        r2 = rank + 1
        tail = None
        while r2 < world and firsts[r2] == last_key:
            part = allrows[r2].view(feat_shape)          # float64 record: exact transport of the fp32 / 16-bit row
            tail = part if tail is None else tail + part
            if lasts[r2] != last_key:
                break  # that rank has further keys: the run ends inside it
            r2 += 1
        if tail is not None:
            local[-1].add_(tail)  # in place (one kernel): `local` is this call's own buffer
    # my rows: (last_key_{rank-1}, last_key]; drop a first row owned by a lower rank,
    # prepend zero rows for the empty keys between the previous rank's last key and my first key
    prev_last = lasts[rank - 1] if rank > 0 else -1
    if not owns_first:
        return local[1:], first_key + 1
    gap = first_key - (prev_last + 1)
    if gap > 0:
        local = torch.cat([local.new_zeros((gap,) + feat_shape), local])
    return local, prev_last + 1


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
    skips the row exchange.

    ``key_offset``: the shard's index is already rank-local and its first key is 0
    (global key = local key + key_offset); saves the pass that re-bases the keys.

    Host synchronisation without stalling the GPU, and without ever issuing a second collective
    (every rank always makes exactly ONE all_gather per call, so ranks cannot fall out of step):
      1. this rank's end keys are remembered per index identity; the local kernels are launched for the
         remembered row count while the D2H copy of the real end keys completes underneath; a mismatch
         relaunches the LOCAL kernels only - nothing has been sent yet;
      2. the all_gather carries every rank's exact keys and first-row partial; one small D2H copy brings
         the keys to the host (the only other host sync of the call);
      3. the ownership step (add the following ranks' partials to my last row, drop / pad the first row)
         is decided from those exact keys and queued behind everything else.
    """
    local_op = local_op or _default_local_op
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if index_shard.numel() == 0:
        raise ValueError("every rank must hold at least one edge")
    feat_shape = tuple(src_shard.shape[1:])
    F = int(src_shard[0].numel())
    dev = src_shard.device
    ident = _ident(index_shard, world, rank, key_offset)
    guess = _ends_seen.get(ident) if ident is not None else None

    def run_local(lo, hi):
        if key_offset is None:
            return local_op(index_shard - lo if lo else index_shard, src_shard, hi - lo + 1)
        out = local_op(index_shard, src_shard, hi + 1)
        return out[lo:] if lo else out

    # ---- 1. local reduction, rows [first_key, last_key] ------------------------------------------------
    ends_dev = index_shard[::max(index_shard.numel() - 1, 1)][:2]     # [first, last] as one strided view, no kernel
    if ends_dev.numel() == 1:
        ends_dev = ends_dev.expand(2)
    if guess is None:
        ends = ends_dev.cpu()                                       # first call: D2H sync, as index[-1].item()
        lo, hi = int(ends[0]), int(ends[1])
        local = run_local(lo, hi)
    else:
        lo, hi = guess["local_ends"]
        if dev.type == "cuda":
            host, ev = _pinned_slot()
            host.copy_(ends_dev, non_blocking=True)
            ev.record(torch.cuda.current_stream(dev))
            local = run_local(lo, hi)                               # queued behind the copy
            ev.synchronize()
            true_ends = (int(host[0]), int(host[1]))
        else:                                                       # host tensors: nothing to overlap
            true_ends = (int(ends_dev[0]), int(ends_dev[1]))
            local = run_local(*true_ends)
        if true_ends != (lo, hi):                                   # index changed under the same identity
            if dev.type == "cuda":                                  # (the HIP kernels ignore out-of-range keys)
                local = run_local(*true_ends)
            lo, hi = true_ends
            guess = None
    first_key, last_key = (key_offset or 0) + lo, (key_offset or 0) + hi
    head = local[0]

    if world == 1:
        if ident is not None:
            _ends_seen[ident] = {"local_ends": (lo, hi)}
        if first_key > 0:
            local = torch.cat([local.new_zeros((first_key,) + feat_shape), local])
        return local, 0

    # ---- 2. the one collective: [first_key, last_key, first_row(F)] of every rank ------------------------
    # built from device tensors only (no scalar host->device writes on the step path): two small kernels
    # write the record, the collective moves it, one small D2H copy brings every rank's keys to the host
    n_rec = 2 + (F if exchange else 0)
    rec = torch.empty(n_rec, dtype=torch.float64, device=dev)
    torch.add(ends_dev, int(key_offset or 0), out=rec[:2])          # keys < 2^53: exact in float64
    if exchange:
        rec[2:].copy_(head.reshape(-1))
    # RCCL ("nccl") moves device tensors directly over xGMI; a gloo group (CPU tests, or two test
    # ranks sharing one GPU) stages the few hundred bytes through the host
    via_host = dev.type == "cuda" and dist.get_backend(group) == "gloo"
    send = rec.cpu() if via_host else rec
    recv = torch.empty(world * n_rec, dtype=torch.float64, device=send.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv_host = recv.cpu().view(world, n_rec)                       # host sync: everything above is queued
    firsts = [int(v) for v in recv_host[:, 0]]
    lasts = [int(v) for v in recv_host[:, 1]]
    allrows = recv.to(dev).view(world, n_rec)[:, 2:] if exchange else None

    # ---- 3. ownership: at most two more small kernels, queued behind the local reduction -----------------
    # (they run while the host is already preparing the next call)
    if ident is not None:
        _ends_seen[ident] = {"local_ends": (lo, hi)}
        if len(_ends_seen) > 64:
            _ends_seen.pop(next(iter(_ends_seen)))
    return _apply_boundaries(local, head, allrows, firsts, lasts, rank, world, feat_shape, exchange)


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
