"""Edge-list sharding across GPUs (one process per GPU, torch.distributed; backend "nccl" is RCCL).

The reference is single-GPU (no collective call site anywhere in its tree); this is the
multi-GPU form of the same path.  Segments are independent, so the dst-sorted edge list shards
into contiguous edge ranges, one per rank, and the output stays ROW-SHARDED:

    rank r holds edges [e_r, e_{r+1})  ->  owns dst rows (last_key_{r-1}, last_key_r]

Two ways to cut:
  * ``segment_aligned_cuts``  snaps every cut to a segment start -> no data-path collective;
  * ``equal_edge_cuts``       exact edge balance -> a segment may straddle a cut; each rank's
    partial first row is then exchanged with ONE small collective - an all_gather of W rows of F
    values added by the owner in rank order, or a reduce_scatter of a [W, F] buffer (``collective=``;
    latency-bound either way, microseconds over xGMI).  A hub that spans several ranks is handled by
    the same pass.
Which case applies is decided per call from every rank's end keys (a 16-byte-per-rank all_gather that runs
underneath the local kernels), so the host never waits for the reduction and the GPU never idles on the host.

``local_op(index_local, src_local, rows) -> [rows, F]`` is the per-rank reduction; by default the
HIP operator.  Tests inject a CPU function to exercise the exchange logic under gloo.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import weakref

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


def _default_local_op(index_local: torch.Tensor, src_local: torch.Tensor, rows: int, reduce: str = "sum") -> torch.Tensor:
    from . import hip
    out = torch.empty((rows,) + tuple(src_local.shape[1:]), dtype=src_local.dtype, device=src_local.device)
    return hip.index_scatter_out(index_local.contiguous(), src_local.contiguous(), out, sorted=True, reduce=reduce)


# (first_key, last_key) of THIS rank's shard, remembered per index identity: GNN edge lists are static, so
# the local row count is almost always the same as last time.  It is only a GUESS for launching the local
# kernels early: the true keys of every rank arrive underneath them and are verified before any row is sent.
_ends_seen: dict = {}
_tls = __import__("threading").local()


def _pinned_slot(dev: torch.device, world: int):
    slots = getattr(_tls, "slots", None)
    if slots is None:
        slots = _tls.slots = {}
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), world)
    slot = slots.get(key)             # per device: a CUDA event is bound to the device of its first record
    if slot is None:
        slot = slots[key] = (torch.empty(2 * world, dtype=torch.int64).pin_memory(), torch.cuda.Event())
    return slot


def _side_stream(dev: torch.device) -> "torch.cuda.Stream":
    streams = getattr(_tls, "side", None)
    if streams is None:
        streams = _tls.side = {}
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in streams:
        streams[key] = torch.cuda.Stream(device=dev)
    return streams[key]


def _ident(index: torch.Tensor, world: int, rank: int, key_offset):
    try:
        version = index._version
    except RuntimeError:      # inference tensors keep no version counter: never remembered
        return None
    return (index.device.type, index.device.index, index.data_ptr(), index.numel(), version, world, rank, key_offset)


def boundary_plan(firsts: List[int], lasts: List[int], rank: int) -> dict:
    """Ownership rules, decided on the HOST from every rank's (first_key, last_key).  Every rank computes the same
    table, so all ranks agree on whether a row exchange is needed at all.

    rank r owns dst rows (last_key_{r-1}, last_key_r].  `joins` = the following ranks whose first-row partial
    belongs to my last row (a hub may span several ranks); `owns_first` = my first row is mine (the previous
    rank does not end on the same key); `gap` = empty keys between the previous rank's last key and my first."""
    world = len(firsts)
    first_key, last_key = firsts[rank], lasts[rank]
    owns_first = rank == 0 or lasts[rank - 1] != first_key
    joins = []
    r2 = rank + 1
    while r2 < world and firsts[r2] == last_key:
        joins.append(r2)
        if lasts[r2] != last_key:
            break                      # that rank has further keys: the run ends inside it
        r2 += 1
    if not owns_first and first_key == last_key:
        joins = []                     # my whole shard lies inside a run owned by a lower rank: it adds my row
    prev_last = lasts[rank - 1] if rank > 0 else -1
    any_shared = any(lasts[r - 1] == firsts[r] for r in range(1, world))
    # who owns my first row: the lowest rank that ENDS on my first key (ranks hold contiguous ranges of a sorted list, so
    # every rank between it and me ends on that key too); myself when the previous rank ends on another key
    owner = rank if owns_first else min(r for r in range(rank) if lasts[r] == first_key)
    return {"owns_first": owns_first, "joins": joins, "gap": (first_key - (prev_last + 1)) if owns_first else 0,
            "first_row": (prev_last + 1) if owns_first else first_key + 1, "any_shared": any_shared, "owner": owner}


_IDENTITY = {"sum": 0.0, "mean": 0.0, "max": float("-inf"), "min": float("inf"), "prod": 1.0}
_ALIASES = {"amax": "max", "amin": "min", "add": "sum"}


def _reduce_op(reduce: str):
    return {"sum": dist.ReduceOp.SUM, "mean": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN,
            "prod": dist.ReduceOp.PRODUCT}[reduce]


def _combine(reduce: str, acc: torch.Tensor, other: torch.Tensor) -> torch.Tensor:
    """acc (op)= other, in place; max / min propagate NaN like ATen (csrc/cpu/index_scatter_cpu.cpp:124-134)."""
    if reduce in ("sum", "mean"):
        return acc.add_(other)
    if reduce == "prod":
        return acc.mul_(other)
    return acc.copy_(torch.maximum(acc, other) if reduce == "max" else torch.minimum(acc, other))


def _sharded_reduce(index_shard: torch.Tensor, feat_shape: tuple, dtype: torch.dtype, dev: torch.device, local_fn: Callable,
                    group, exchange: bool, key_offset: Optional[int], timing: Optional[dict], collective: str,
                    reduce: str) -> Tuple[torch.Tensor, int]:
    """The protocol of :func:`sharded_index_scatter` over an abstract local reduction:
    ``local_fn(e0, e1, lo, rows, reduce) -> [rows, *feat]`` reduces the shard's edges [e0, e1) with keys ``index - lo``."""
    if collective not in ("all_gather", "reduce_scatter"):
        raise ValueError("collective must be 'all_gather' or 'reduce_scatter'")
    reduce = _ALIASES.get(reduce, reduce)
    if reduce not in _IDENTITY:
        raise ValueError(f"reduce argument must be either sum, prod, mean, amax or amin, got {reduce}")
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nnz = index_shard.numel()
    if nnz == 0:
        raise ValueError("every rank must hold at least one edge")
    F = 1
    for d in feat_shape:
        F *= int(d)
    on_gpu = dev.type == "cuda"
    ident = _ident(index_shard, world, rank, key_offset)
    guess = _ends_seen.get(ident) if ident is not None else None
    off = int(key_offset or 0)

    def run_local(lo, hi):
        return local_fn(0, nnz, lo, hi - lo + 1, reduce)

    ends_dev = index_shard[::max(nnz - 1, 1)][:2]     # [first, last] as one strided view, no kernel
    if ends_dev.numel() == 1:
        ends_dev = ends_dev.expand(2)

    # ---- 1. keys of every rank: a tiny collective on a SIDE stream, so the local kernels do not queue behind it ------
    # RCCL ("nccl") moves device tensors directly over xGMI; a gloo group (CPU tests, or test ranks sharing
    # one GPU) stages the few bytes through the host
    via_host = world > 1 and on_gpu and dist.get_backend(group) == "gloo"
    waiter = None
    if on_gpu and not via_host:
        main, side = torch.cuda.current_stream(dev), _side_stream(dev)
        side.wait_stream(main)                                      # the index is ready wherever main is now
        with torch.cuda.stream(side):                               # (tensors made here live in the side stream's pool)
            kev = None
            if timing is not None:
                kev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                kev[0].record(side)
            mine = (ends_dev + off) if off else ends_dev.contiguous()
            if world > 1:
                allkeys = torch.empty(2 * world, dtype=torch.int64, device=dev)
                dist.all_gather_into_tensor(allkeys, mine, group=group)
            else:
                allkeys = mine
            host, waiter = _pinned_slot(dev, world)
            host.copy_(allkeys, non_blocking=True)
            waiter.record(side)
            if kev is not None:
                kev[1].record(side)
                timing.setdefault("key_events", []).append(kev)
    elif world > 1:
        import time
        t0 = time.perf_counter()
        mine = (ends_dev + off) if off else ends_dev.contiguous()
        send = mine.cpu() if via_host else mine
        host = torch.empty(2 * world, dtype=torch.int64, device=send.device)
        dist.all_gather_into_tensor(host, send, group=group)
        if timing is not None:
            timing.setdefault("key_wall_ms", []).append((time.perf_counter() - t0) * 1e3)
    else:
        host = (ends_dev + off) if off else ends_dev

    # ---- 2. local reduction, rows [first_key, last_key], launched on the remembered keys ------------------
    local = run_local(*guess) if (guess is not None and on_gpu) else None   # host tensors: nothing to overlap
    if waiter is not None:
        waiter.synchronize()                                        # the keys landed while the kernels run
    keys = host.tolist()
    firsts, lasts = keys[0::2], keys[1::2]
    lo, hi = firsts[rank] - off, lasts[rank] - off
    if local is None or guess != (lo, hi):                          # first call, or the index changed under the same identity
        local = run_local(lo, hi)                                   # (the HIP kernels ignore out-of-range keys)
    if ident is not None:
        _ends_seen[ident] = (lo, hi)
        if len(_ends_seen) > 64:
            _ends_seen.pop(next(iter(_ends_seen)))

    plan = boundary_plan(firsts, lasts, rank)
    if not exchange and plan["any_shared"]:
        r = next(r for r in range(1, world) if lasts[r - 1] == firsts[r])
        raise RuntimeError(f"exchange=False but a segment straddles the cut between ranks {r - 1} and {r}")

    # ---- 3. first-row partials, only when some key is shared (all ranks agree: they all hold all keys) -----
    if plan["any_shared"]:
        ev = None
        if timing is not None and on_gpu:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(torch.cuda.current_stream(dev))
        mean = reduce == "mean"
        # what travels per rank: its first-row partial in a type that holds it exactly - fp32 (fp32 / 16-bit rows) or fp64;
        # mean: the partial SUM and the edge count of the row, F + 1 doubles (the owner divides once, at the end)
        xdtype = torch.float64 if (local.dtype == torch.float64 or mean) else torch.float32
        width = F + 1 if mean else F
        if mean:
            # the local kernels wrote MEANS; the shared rows travel as (partial sum, edge count).  The partial sum is the row's
            # mean times its edge count, formed in float64 ON THE DEVICE: never a sum in the storage type (a hub of > 65 k
            # edges of O(1) values overflows an fp16 sum; the mean is always representable), no extra kernel launch over the
            # row's edges, no host read-back.  Against the unsharded op: fp32 rows within 2 ulp, 16-bit rows within one ulp of the
            # storage type (each rank's mean is rounded once before it travels).
            rec = torch.zeros(width, dtype=xdtype, device=dev)
            own_sum, own_cnt = None, None
            if not plan["owns_first"]:
                c_first = torch.searchsorted(index_shard, index_shard[:1], right=True)[0].to(xdtype)   # edges of my first key
                rec[:F] = local[0].reshape(-1).to(xdtype) * c_first
                rec[F] = c_first
            if plan["joins"]:
                own_cnt = (nnz - torch.searchsorted(index_shard, index_shard[-1:], right=False)[0]).to(xdtype)
                own_sum = local[-1].reshape(-1).to(xdtype) * own_cnt
        else:
            rec = local[0].reshape(-1).to(xdtype)
        if collective == "reduce_scatter":
            # row `owner` of my [W, width] buffer = my first-row partial (if another rank owns that row), every other row the
            # reduction's identity; the reduction over ranks of row o is what rank o has to combine into its last row
            send = torch.full((world, width), _IDENTITY[reduce], dtype=xdtype, device=rec.device)
            if not plan["owns_first"]:
                send[plan["owner"]].copy_(rec)
            send = send.view(-1).cpu() if via_host else send.view(-1)
            recv = torch.empty(width, dtype=xdtype, device=send.device)
            dist.reduce_scatter_tensor(recv, send, op=_reduce_op(reduce), group=group)
            joined = (recv.to(dev) if via_host else recv) if plan["joins"] else None
        else:
            send = rec.cpu() if via_host else rec
            recv = torch.empty(world * width, dtype=xdtype, device=send.device)
            dist.all_gather_into_tensor(recv, send, group=group)
            allrows = (recv.to(dev) if via_host else recv).view(world, width)
            joined = None
            if mean and plan["joins"]:
                joined = allrows[plan["joins"]].sum(0)               # (fp64: exact for fp32 partial sums of any realistic length)
        if plan["joins"]:
            if mean:
                total = (own_sum + joined[:F]) / (joined[F] + own_cnt)
                local[-1].copy_(total.view(feat_shape).to(local.dtype))
            elif joined is not None:
                _combine(reduce, local[-1], joined.view(feat_shape).to(local.dtype))
            else:
                for r2 in plan["joins"]:                            # rank order: deterministic
                    _combine(reduce, local[-1], allrows[r2].view(feat_shape).to(local.dtype))   # in place: `local` is this call's own buffer
        if ev is not None:
            ev[1].record(torch.cuda.current_stream(dev))
            timing.setdefault("exchange_events", []).append(ev)

    if not plan["owns_first"]:
        return local[1:], plan["first_row"]
    if plan["gap"] > 0:                                             # empty keys in front of my first key (rare)
        local = torch.cat([local.new_zeros((plan["gap"],) + tuple(feat_shape)), local])
    return local, plan["first_row"]


def sharded_index_scatter(index_shard: torch.Tensor, src_shard: torch.Tensor,
                          group: Optional[dist.ProcessGroup] = None,
                          local_op: Optional[Callable] = None,
                          exchange: bool = True,
                          key_offset: Optional[int] = None,
                          timing: Optional[dict] = None,
                          collective: str = "all_gather",
                          reduce: str = "sum") -> Tuple[torch.Tensor, int]:
    """Row-sharded index_scatter over the ranks of ``group``.

    ``index_shard`` / ``src_shard`` are this rank's contiguous slice of the globally dst-sorted
    edge list (rank order = edge order; every rank holds at least one edge).  Returns
    ``(out_rows, first_row)``: this rank's rows of the global result and the global row number of
    its first row.  Concatenating the ranks' ``out_rows`` in rank order gives exactly
    ``index_scatter(0, src, index, reduce)`` of the unsharded problem.

    ``exchange=False`` asserts the cuts are segment-aligned (no key is shared by two ranks).

    ``key_offset``: the shard's index is already rank-local and its first key is 0
    (global key = local key + key_offset); saves the pass that re-bases the keys.

    Protocol (the host never waits for the local reduction, the GPU never waits for the host):
      1. KEYS FIRST: an all_gather of every rank's (first_key, last_key) - 16 bytes per rank, independent of the
         reduction - runs on a side stream beside the local kernels and is copied to the host underneath them;
      2. the local kernels are launched for the REMEMBERED row count of this index tensor while those keys are
         in flight; when they arrive the guess is verified (a mismatch relaunches the local kernels only);
      3. every rank now knows every boundary: if no key is shared by two ranks (segment-aligned cuts) the call
         is done - no data-path collective at all; otherwise ONE collective of the ranks' first-row partials
         (W x F values) follows the local kernels on the stream, and the owner combines the partials that belong to its
         last row in rank order (deterministic) - all queued, no host wait.
    Every rank issues the same collectives in the same order whatever its local verification says.

    ``collective``: how the first-row partials travel in step 3 (both give the same rows; W x F values either way):
      * ``"all_gather"`` (default): every rank receives every rank's first-row partial and the owner combines the ones that
        belong to its last row, in rank order;
      * ``"reduce_scatter"`` (the north star's wording): every rank sends a [W, F] buffer that holds the reduction's identity
        except for row ``owner(my first key)`` = its first-row partial; ``reduce_scatter`` hands rank o the reduction of
        the partials it owns - the combining happens inside RCCL - and o combines that one row into its last row.
    ``reduce``: 'sum' (default) | 'mean' | 'max' / 'amax' | 'min' / 'amin' | 'prod' - the reductions of the reference's CPU path
    (csrc/cpu/index_scatter_cpu.cpp:124-134).  max / min / prod travel as the row itself (identity elsewhere, ReduceOp.MAX /
    MIN / PRODUCT in the reduce_scatter form); mean ships (partial sum, edge count) as F + 1 doubles and the owner divides once.
    ``local_op(index_local, src_local, rows[, reduce=...]) -> [rows, F]`` (the HIP operator by default) must accept ``reduce``
    when a reduction other than sum is asked for.
    ``timing``: optional dict; receives hipEvent pairs around the exchange ("exchange_events") and around the key
    all_gather + copy on the side stream ("key_events"; "key_wall_ms" where the keys travel through the host).
    """
    local_op = local_op or _default_local_op

    def local_fn(e0, e1, lo, rows, red):
        idx = index_shard if (e0 == 0 and e1 == index_shard.numel()) else index_shard[e0:e1]
        src = src_shard if (e0 == 0 and e1 == index_shard.numel()) else src_shard[e0:e1]
        idx = idx - lo if lo else idx
        return local_op(idx, src, rows) if red == "sum" else local_op(idx, src, rows, reduce=red)

    return _sharded_reduce(index_shard, tuple(src_shard.shape[1:]), src_shard.dtype, src_shard.device, local_fn, group, exchange,
                           key_offset, timing, collective, reduce)


def sharded_gather_scatter(src_index_shard: torch.Tensor, dst_index_shard: torch.Tensor,
                           src: torch.Tensor, weight_shard: Optional[torch.Tensor] = None,
                           group: Optional[dist.ProcessGroup] = None,
                           local_op: Optional[Callable] = None, exchange: bool = True,
                           key_offset: Optional[int] = None, timing: Optional[dict] = None,
                           collective: str = "all_gather", reduce: str = "sum") -> Tuple[torch.Tensor, int]:
    """Row-sharded gather_scatter / gather_weight_scatter (BASELINE.json configs[4]).

    The edge list (src_index, dst_index[, weight]) is sharded by contiguous dst-sorted edge ranges
    exactly like :func:`sharded_index_scatter`; ``src`` (node features) is REPLICATED on every rank
    (56.9 GB per GPU at papers100M scale fits 288 GB).  Same boundary-row
    exchange, same return value, same ``reduce`` (PyG's ``aggr``).  ``local_op(src_index, dst_index_local, weight, src, rows
    [, reduce=...])`` defaults to the HIP operators.
    """
    if local_op is None:
        from . import hip

        def local_op(si, di, w, x, rows, reduce="sum"):
            out = torch.empty((rows, x.shape[1]), dtype=x.dtype, device=x.device)
            if reduce != "sum":
                return hip.gather_reduce_out(si.contiguous(), di, None if w is None else w.contiguous(), x, out, reduce)
            if w is None:
                return hip.gather_scatter_out(si.contiguous(), di, x, out)
            return hip.gather_weight_scatter_out(si.contiguous(), di, w.contiguous(), x, out)

    nnz = dst_index_shard.numel()

    def local_fn(e0, e1, lo, rows, red):
        whole = e0 == 0 and e1 == nnz
        si = src_index_shard if whole else src_index_shard[e0:e1]
        di = dst_index_shard if whole else dst_index_shard[e0:e1]
        w = weight_shard if (whole or weight_shard is None) else weight_shard[e0:e1]
        di = di - lo if lo else di
        return local_op(si, di, w, src, rows) if red == "sum" else local_op(si, di, w, src, rows, reduce=red)

    return _sharded_reduce(dst_index_shard, tuple(src.shape[1:]), src.dtype, src.device, local_fn, group, exchange, key_offset, timing,
                           collective, reduce)


# ---- node-sharded source features (second option) -----------------------------------------------------------
# Replicating `src` costs 56.9 GB per GPU at configs[4] and caps the graph at what one GPU holds.  With the nodes partitioned into
# contiguous ranges - rank r holds src[node_offsets[r] : node_offsets[r + 1]] - a rank's edge range references a set of source rows
# that is FIXED per edge list: found once (`HaloPlan`), fetched per call with ONE all_to_all_single (every rank sends each other rank
# the rows it asked for), and the local kernel runs over the compact table of exactly those rows (src_index renumbered once).  When
# the halo is most of the table anyway (> `all_gather_above` of it: uniform-random sources at small world sizes) one all_gather of
# the shards is cheaper and needs no renumbering of the features - the plan then says so.  The reference has no counterpart
# (single-GPU; csrc/gather_scatter.cpp:25-34 reads the whole table).
class HaloPlan:
    """What rank `rank` must fetch to run its edge range, for one (edge list, node partition).  Built collectively (`build`)."""

    def __init__(self):
        self.mode = "halo"            # "halo": all_to_all_single of the rows asked for | "all_gather": every shard, padded to the largest
        self.compact_index = None     # int64 [nnz]: src_index renumbered into the table the local kernel gathers from
        self.table_rows = 0
        self.send_rows = None         # int64: MY local rows the ranks asked for, grouped by asking rank (mine included)
        self.send_splits = self.recv_splits = None
        self.pad_rows = 0             # (all_gather) rows per rank in the gathered table
        self.rows_fetched = 0         # rows this rank receives from OTHER ranks per call
        self.rows_table_total = 0

    @staticmethod
    def build(src_index_shard: torch.Tensor, node_offsets: List[int], group=None, all_gather_above: float = 0.5) -> "HaloPlan":
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        if len(node_offsets) != world + 1 or any(node_offsets[i] > node_offsets[i + 1] for i in range(world)):
            raise ValueError("node_offsets: world + 1 ascending row offsets")
        dev = src_index_shard.device
        host = dev.type == "cuda" and dist.get_backend(group) == "gloo"     # (test ranks sharing one GPU: stage through the host)
        offs = torch.tensor(node_offsets, dtype=torch.int64, device=dev)
        n_total = int(node_offsets[-1])
        plan = HaloPlan()
        plan.rows_table_total = n_total
        needed = torch.unique(src_index_shard)                                # ascending global ids = owner-major (owners are contiguous ranges)
        if needed.numel() and (int(needed[0]) < 0 or int(needed[-1]) >= n_total):
            raise ValueError("src_index outside the node partition")
        # every rank takes the same decision: the LARGEST halo decides (one collective either way)
        frac = torch.tensor([needed.numel() / max(n_total, 1)], dtype=torch.float64, device="cpu" if host or dev.type == "cpu" else dev)
        dist.all_reduce(frac, op=dist.ReduceOp.MAX, group=group)
        if float(frac.item()) > all_gather_above:
            plan.mode = "all_gather"
            plan.pad_rows = max(node_offsets[r + 1] - node_offsets[r] for r in range(world))
            owner = torch.bucketize(src_index_shard, offs[1:], right=True)
            plan.compact_index = (owner * plan.pad_rows + (src_index_shard - offs[owner])).contiguous()
            plan.table_rows = world * plan.pad_rows
            plan.rows_fetched = n_total - (node_offsets[rank + 1] - node_offsets[rank])
            return plan
        plan.compact_index = torch.searchsorted(needed, src_index_shard).contiguous()
        plan.table_rows = int(needed.numel())
        bounds = torch.searchsorted(needed, offs)                               # needed[bounds[o] : bounds[o + 1]] live on rank o
        want = (bounds[1:] - bounds[:-1])
        plan.recv_splits = [int(v) for v in want.tolist()]
        asked = torch.empty(world, dtype=torch.int64, device="cpu" if host else dev)
        dist.all_to_all_single(asked, want.cpu() if host else want, group=group)  # how many rows each rank asks ME for
        plan.send_splits = [int(v) for v in asked.tolist()]
        ids = torch.empty(sum(plan.send_splits), dtype=torch.int64, device="cpu" if host else dev)
        dist.all_to_all_single(ids, needed.cpu() if host else needed, plan.send_splits, plan.recv_splits, group=group)
        plan.send_rows = (ids.to(dev) - node_offsets[rank]).contiguous()
        plan.rows_fetched = plan.table_rows - plan.recv_splits[rank]
        return plan

    def fetch(self, src_shard: torch.Tensor, group=None) -> torch.Tensor:
        """The table the local kernel gathers from, [table_rows, *feat]: one collective."""
        dev = src_shard.device
        host = dev.type == "cuda" and dist.get_backend(group) == "gloo"
        feat = tuple(src_shard.shape[1:])
        if self.mode == "all_gather":
            mine = src_shard
            if mine.shape[0] < self.pad_rows:
                mine = torch.cat([mine, mine.new_zeros((self.pad_rows - mine.shape[0],) + feat)])
            send = mine.contiguous().cpu() if host else mine.contiguous()
            table = torch.empty((self.table_rows,) + feat, dtype=src_shard.dtype, device=send.device)
            dist.all_gather_into_tensor(table, send, group=group)
            return table.to(dev) if host else table
        send = src_shard.index_select(0, self.send_rows)
        send = send.cpu() if host else send
        table = torch.empty((self.table_rows,) + feat, dtype=src_shard.dtype, device=send.device)
        dist.all_to_all_single(table, send, self.recv_splits, self.send_splits, group=group)
        return table.to(dev) if host else table

    def bytes_fetched(self, row_bytes: int) -> int:
        return int(self.rows_fetched) * int(row_bytes)


_halo_seen: dict = {}


def sharded_gather_scatter_node(src_index_shard: torch.Tensor, dst_index_shard: torch.Tensor, src_shard: torch.Tensor,
                                node_offsets: List[int], weight_shard: Optional[torch.Tensor] = None,
                                group: Optional[dist.ProcessGroup] = None, local_op: Optional[Callable] = None,
                                exchange: bool = True, key_offset: Optional[int] = None, timing: Optional[dict] = None,
                                collective: str = "all_gather", reduce: str = "sum", halo: Optional[HaloPlan] = None,
                                all_gather_above: float = 0.5) -> Tuple[torch.Tensor, int]:
    """:func:`sharded_gather_scatter` with NODE-SHARDED source features: ``src_shard`` = this rank's rows
    ``src[node_offsets[rank] : node_offsets[rank + 1]]`` (``src_index_shard`` keeps GLOBAL node ids).  Same rows as the replicated
    form (the local kernel sees the same values in the same edge order: bit-equal).  ``halo``: a plan built earlier with
    ``HaloPlan.build`` (a collective); otherwise built on the first call with this edge list and remembered per index identity.
    ``timing`` receives "halo_plan" (the plan: mode, rows fetched) and, on the GPU, event pairs around the fetch ("fetch_events")."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if halo is None:
        # The plan's compact_index REPLACES src_index_shard, so a remembered plan is used only for the very tensor OBJECT it was
        # built from, still alive and unwritten (weak reference + version counter): an equally long shard at a recycled address
        # (fixed-fanout sampling + the caching allocator) is a different object and misses - on every rank alike, so all ranks
        # enter the collective build together.  A caller that makes a fresh view of its shard per step passes `halo=` instead.
        ident = _ident(src_index_shard, world, rank, tuple(node_offsets))
        seen = _halo_seen.get(ident) if ident is not None else None
        halo = seen[1] if seen is not None and seen[0]() is src_index_shard else None
        if halo is None:
            halo = HaloPlan.build(src_index_shard, node_offsets, group, all_gather_above)
            if ident is not None:
                _halo_seen[ident] = (weakref.ref(src_index_shard), halo)
                for k in [k for k, (ref, _) in _halo_seen.items() if ref() is None]:
                    del _halo_seen[k]
                if len(_halo_seen) > 16:
                    _halo_seen.pop(next(iter(_halo_seen)))
    ev = None
    if timing is not None and src_shard.is_cuda:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    table = halo.fetch(src_shard, group)
    if ev is not None:
        ev[1].record()
        timing.setdefault("fetch_events", []).append(ev)
    if timing is not None:
        timing["halo_plan"] = halo
    return sharded_gather_scatter(halo.compact_index, dst_index_shard, table, weight_shard, group, local_op, exchange, key_offset, timing,
                                  collective, reduce)


def shard_edges(index: torch.Tensor, src: torch.Tensor, world: int, rank: int, aligned: bool = False):
    """Slice a replicated (index, src) pair for ``rank`` (helper for tests and examples)."""
    cuts = segment_aligned_cuts(index, world) if aligned else equal_edge_cuts(index.numel(), world)
    return index[cuts[rank]:cuts[rank + 1]], src[cuts[rank]:cuts[rank + 1]]
