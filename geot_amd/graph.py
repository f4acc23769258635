"""``geot_amd.Graph`` - an opt-in handle for a STATIC graph that owns everything derived from its edge list.

The operators of ``geot_amd.ops`` follow the reference's contract: plain tensors in, every call stands alone
(csrc/gather_scatter.cpp:25-34 reads the caller's tensors each time).  What the host layer remembers between such calls - the
source-blocked plan of a dense graph, the transposed edge list of the backward pass - it must re-verify at every use (the content
guard: both index arrays re-read per call, +4.6 % at configs[3], +14 % on a training step).  A caller who
KNOWS the graph is static can say so once::

    g = geot_amd.Graph(src_index, dst_index)          # clones the two index tensors: nobody else can write to them
    y = g.gather_weight_scatter(w, x)                  # = geot_amd.gather_weight_scatter(src_index, dst_index, w, x)
    y = g.mh_spmm(w, x)                                # = geot_amd.mh_spmm(src_index, dst_index, w, x)
    s = g.mh_sddmm(q, k, plan_order=True)              # attention scores that never leave the plan's edge order ...
    a = s.with_values(torch.exp(s.values))             #   ... elementwise work on .values ...
    y = g.mh_spmm(a, v)                                #   ... and straight back into the SpMM: no permutation in between

The handle owns its clones, the forward plan and the plan of the transposed list (built on first use, tried once against the
per-edge kernels like the host layer's `plan_or_edges`), the transposed edge list and the permutations between the orders; its
methods call the C ABI directly (geot_amd/hip.py, geot_amd/slab.py): no fingerprint, no cache lookup, no per-call read of the
index arrays beyond what the kernels read.  All methods are differentiable (sum; mean for the unweighted / weighted gather).
Same kernels, same results as the operators (tests/test_gpu_graph_handle.py: bit-equal on the same path).

Reference semantics: csrc/util/check.cuh:90-111 (gather / gws), test/test_mh_spmm.py:4-10 (multi-head), autograd pattern
geot/gather_weight_scatter.py:31-51.
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch

from . import hip, slab

_SLAB_DTYPES = (torch.float32, torch.float16, torch.bfloat16)


class PlanOrdered:
    """Per-edge values ([nnz] or [nnz, H]) in the edge order of one of a Graph's plans.  ``values`` is an ordinary tensor (autograd
    flows through it); ``with_values`` wraps the result of elementwise work; ``edge_order()`` leaves the plan's order."""

    def __init__(self, graph: "Graph", plan, values: torch.Tensor):
        """``plan is None``: the graph has no plan for this shape (not dense enough, a row width the source-blocked kernels do not
        serve) - the values are in the ORIGINAL edge order, and every method still works."""
        self.graph, self.plan, self.values = graph, plan, values

    def with_values(self, values: torch.Tensor) -> "PlanOrdered":
        if values.shape[0] != self.values.shape[0]:
            raise ValueError("with_values: one entry per edge, in the same order")
        return PlanOrdered(self.graph, self.plan, values)

    def edge_order(self) -> torch.Tensor:
        """The values in ORIGINAL edge order (one scatter through the plan's permutation)."""
        if self.plan is None:
            return self.values
        return _PlanToEdge.apply(self.graph, self.plan, self.values)

    @property
    def dst(self) -> torch.Tensor:
        """Destination row of every position (int64 [nnz]) - the key of a per-row softmax over ``values``."""
        return self.graph.dst_index if self.plan is None else self.graph._plan_endpoints(self.plan)[0]

    @property
    def src(self) -> torch.Tensor:
        return self.graph.src_index if self.plan is None else self.graph._plan_endpoints(self.plan)[1]


def _rows(t: torch.Tensor) -> int:
    return int(t.shape[0])


_GATHER_DTYPES = (torch.float32, torch.float64, torch.float16, torch.bfloat16)


def _rows_at(values: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """``values[index]`` (int64 ``index``, rows of ``values``) by the library's row gather (geot_gather_rows).  NOT torch's advanced
    indexing: on this stack ``t[idx]`` of a [115 M, 8] bfloat16 tensor came back with garbage in its last 2^26 rows, different on
    every call (found by the full-size test of eight heads: tests/test_gpu_round6.py); ``t[idx] = v`` goes through the same
    machinery - the scatters of this module are gathers through the inverse permutation instead."""
    v = values.contiguous()
    if not (v.is_cuda and v.dtype in _GATHER_DTYPES and index.dtype == torch.int64 and v.shape[0] > 0):
        return v[index]
    out = torch.empty((index.numel(),) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
    if index.numel():
        hip.gather_rows_out(index.contiguous(), v, out)
    return out


class Graph:
    def __init__(self, src_index: torch.Tensor, dst_index: torch.Tensor, num_src: Optional[int] = None, num_dst: Optional[int] = None,
                 clone: bool = True, slab_mode: str = "auto"):
        """``dst_index`` ascending (checked once).  ``clone=False``: the caller promises never to write to the two tensors again.
        ``slab_mode``: 'auto' (density rule + one trial per plan), 'always', 'never'."""
        if src_index.dim() != 1 or dst_index.dim() != 1 or src_index.shape != dst_index.shape:
            raise RuntimeError("src_index and dst_index must be 1 dimensional")     # (csrc/cuda/gather_scatter_cuda.cu:18)
        if not src_index.is_cuda:
            raise RuntimeError("geot_amd.Graph: GPU tensors only (the package has no CPU path)")
        if slab_mode not in ("auto", "always", "never"):
            raise ValueError("slab_mode: 'auto' | 'always' | 'never'")
        self.src_index = src_index.to(torch.int64).contiguous()
        self.dst_index = dst_index.to(torch.int64).contiguous()
        if clone:
            self.src_index = self.src_index.clone() if self.src_index.data_ptr() == src_index.data_ptr() else self.src_index
            self.dst_index = self.dst_index.clone() if self.dst_index.data_ptr() == dst_index.data_ptr() else self.dst_index
        self.nnz = int(self.dst_index.numel())
        if self.nnz == 0:
            raise IndexError("index -1 is out of bounds for dimension 0 with size 0")
        probe = hip.index_probe_range_out(self.dst_index, torch.empty(4, dtype=torch.int64, device=self.dst_index.device)).tolist()
        if probe[1] != 0 or probe[2] < 0:
            raise ValueError("geot_amd.Graph: dst_index must be ascending and non-negative (sort the edge list by destination once)")
        smin_max = torch.stack([self.src_index.min(), self.src_index.max()]).tolist()
        if smin_max[0] < 0:
            raise ValueError("geot_amd.Graph: negative source index")
        self.rows = int(num_dst) if num_dst is not None else probe[0] + 1       # the reference's row rule, read once
        self.src_rows = int(num_src) if num_src is not None else smin_max[1] + 1
        if probe[0] >= self.rows or smin_max[1] >= self.src_rows:
            raise ValueError("geot_amd.Graph: an index exceeds num_dst / num_src")
        self.slab_mode = slab_mode
        self._t = None                  # (perm, t_src = dst_index[perm], t_dst = src_index sorted): the transposed list
        self._plans = {}                # (which, rowbytes, R, units) -> SlabPlan | None
        self._verdict = {}              # (plan id, kind) -> use the plan?
        self._inv = {}                  # plan id -> inverse of e_perm (edge id -> plan position)
        self._p64 = {}                  # plan id -> e_perm as int64
        self._ends = {}                 # plan id -> (dst per plan position, src per plan position)
        self._to_bwd = {}               # (fwd plan id | None, bwd plan id | None) -> gather index into the forward-side values
        self._deg = None
        self.stats = {"plans_built": 0, "trials": 0, "plan_launches": 0, "edge_launches": 0}

    # ---- derived artefacts (each made once) -------------------------------------------------------------------------------------
    def _transposed(self):
        if self._t is None:
            t_dst, perm = hip.sort_index(self.src_index, self.src_rows - 1)      # stable: destinations ascend inside a source's run
            self._t = (perm, self.dst_index[perm].contiguous(), t_dst)
        return self._t

    def _edges(self, which: str):
        if which == "fwd":
            return self.src_index, self.dst_index, self.rows
        perm, t_src, t_dst = self._transposed()
        return t_src, t_dst, self.src_rows

    def degree(self) -> torch.Tensor:
        """Edges per destination row, float32 [rows] (mean aggregation)."""
        if self._deg is None:
            ones = torch.ones(self.nnz, 1, dtype=torch.float32, device=self.dst_index.device)
            self._deg = hip.index_scatter_out(self.dst_index, ones, torch.empty(self.rows, 1, dtype=torch.float32, device=ones.device)).view(-1)
        return self._deg

    def _plan(self, which: str, rowbytes: int, wmode: int, heads: int, dtype: torch.dtype, table_rows: int, reduce: str = "sum"):
        if self.slab_mode == "never" or dtype not in _SLAB_DTYPES or rowbytes not in (256, 512, 1024) or self.nnz >= 2 ** 31:
            return None
        si, di, rows = self._edges(which)
        if table_rows * rowbytes >= 2 ** 32:
            return None
        # (the host operator's rule, csrc/host_plan.cpp slab_plan_for: 16-bit sums over 256- / 512-byte rows with one weight per edge or
        #  none take the multi-head cut - waves, <= 16 rows per group - which the matrix-core kernels run)
        cut = 2 if (dtype != torch.float32 and rowbytes in (256, 512) and reduce in ("sum", "mean") and wmode in (0, 1)) else wmode
        R = slab.rows_per_group(cut, heads, dtype, rowbytes)
        units = int(slab._lib.load().geot_slab_units_for(cut, rowbytes))
        key = (which, rowbytes, R, units)
        if key not in self._plans:
            plan = None
            if self.slab_mode == "always" or slab.worthwhile(self.nnz, rows, table_rows, rowbytes):
                plan = slab.build_plan(si, di, rows, table_rows, rowbytes, wmode, heads, rows_per_group=R, units=units)
                self.stats["plans_built"] += 1
            self._plans[key] = plan
        return self._plans[key]

    def _use_plan(self, plan, kind: str, run_plan, run_edges) -> bool:
        """One trial per (plan, kind), like the host layer's plan_or_edges: both ways once untimed, then timed twice alternating."""
        if plan is None:
            return False
        if self.slab_mode == "always":
            return True
        key = (id(plan), kind)
        if key not in self._verdict:
            if torch.cuda.is_current_stream_capturing():
                return False
            run_plan()
            run_edges()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            ev[0].record()
            for r in range(2):
                run_plan()
                ev[2 * r + 1].record()
                run_edges()
                ev[2 * r + 2].record()
            ev[4].synchronize()
            t_plan = min(ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3]))
            t_edges = min(ev[1].elapsed_time(ev[2]), ev[3].elapsed_time(ev[4]))
            self._verdict[key] = t_plan <= t_edges
            self.stats["trials"] += 1
        return self._verdict[key]

    def _perm64(self, plan) -> torch.Tensor:
        """The plan's permutation (plan position -> edge id) as the int64 index the row gather takes, made once per plan."""
        p64 = self._p64.get(id(plan))
        if p64 is None:
            p64 = self._p64[id(plan)] = plan.tensors["e_perm"].long()
        return p64

    def _inverse(self, plan) -> torch.Tensor:
        inv = self._inv.get(id(plan))
        if inv is None:
            e_perm = plan.tensors["e_perm"].long()
            inv = torch.empty_like(e_perm)
            inv[e_perm] = torch.arange(e_perm.numel(), device=e_perm.device)
            self._inv[id(plan)] = inv
        return inv

    def _plan_endpoints(self, plan):
        ends = self._ends.get(id(plan))
        if ends is None:
            e = plan.tensors["e_perm"].long()
            ends = self._ends[id(plan)] = (self.dst_index[e].contiguous(), self.src_index[e].contiguous())
        return ends

    def _values_for_bwd(self, fwd_plan, bwd_plan) -> torch.Tensor:
        """Gather index i -> position in the FORWARD-side value array (plan order if fwd_plan, else edge order) of the edge at
        position i of the BACKWARD side (the transposed plan's order if bwd_plan, else the transposed list's)."""
        key = (id(fwd_plan) if fwd_plan is not None else None, id(bwd_plan) if bwd_plan is not None else None)
        idx = self._to_bwd.get(key)
        if idx is None:
            perm = self._transposed()[0]                                          # transposed position -> edge id
            idx = perm if bwd_plan is None else perm[bwd_plan.tensors["e_perm"].long()]
            if fwd_plan is not None:
                idx = self._inverse(fwd_plan)[idx]
            idx = self._to_bwd[key] = idx.contiguous()
        return idx

    def _bwd_reader(self, fwd_plan, bwd_plan):
        """The transposed plan reading its weights THROUGH a composed index straight out of the forward side's value array (edge order,
        or `fwd_plan`'s order): the per-step permutation of the weights (2.1 ms for 115 M values, 13.4 ms step) becomes an indexed read
        inside the row loop (~1 ms)."""
        key = ("reader", id(fwd_plan) if fwd_plan is not None else None, id(bwd_plan))
        twin = self._to_bwd.get(key)
        if twin is None:
            twin = self._to_bwd[key] = bwd_plan.with_e_perm(self._values_for_bwd(fwd_plan, bwd_plan).to(torch.int32).contiguous())
        return twin

    # ---- the kernels (no autograd) ------------------------------------------------------------------------------------------------
    def _spmm(self, which: str, weight, x: torch.Tensor, reduce: str = "sum") -> torch.Tensor:
        """out[d, h, :] = reduce_e w(e, h) x[s_e, h, :] over the forward or the transposed list.  x [N, F] or [N, H, F]; weight:
        None | tensor in the LIST's edge order ([nnz] or [nnz, H]) | (plan, values in that plan's order)."""
        si, di, rows = self._edges(which)
        x = x.contiguous()
        mh = x.dim() == 3
        H, F = (x.shape[1], x.shape[2]) if mh else (1, x.shape[1])
        rowbytes = H * F * x.element_size()
        out = torch.empty((rows, H, F) if mh else (rows, F), dtype=x.dtype, device=x.device)
        via = isinstance(weight, tuple) and len(weight) == 4          # ("via", twin plan, values, list-order fallback): see _bwd_reader
        po = isinstance(weight, tuple) and not via
        wmode_plan = 0 if weight is None else (2 if mh else 1)
        plan = self._plan(which, rowbytes, wmode_plan, H, x.dtype, _rows(x), reduce) if (reduce != "prod") else None
        if mh and ((F * x.element_size()) % 16 != 0 or H > 16):     # (the host operator's rule too: mh_spmm_common in csrc/torch_ops.cpp)
            plan = None
        w_edge = [None]

        def edge_weight():                      # the weight in the list's edge order (made on demand, once)
            if w_edge[0] is None and weight is not None:
                if via:
                    w_edge[0] = weight[3]()
                elif po:
                    wplan, values = weight
                    w_edge[0] = _rows_at(values, self._inverse(wplan))
                else:
                    w_edge[0] = weight.contiguous()
            return w_edge[0]

        def run_edges():
            w = edge_weight()
            self.stats["edge_launches"] += 1
            if mh:
                if w is None:
                    raise RuntimeError("mh_spmm needs a weight")
                return hip.mh_spmm_out(si, di, w, x, out, False)
            if reduce != "sum":
                return hip.gather_reduce_out(si, di, w, x, out, reduce)
            if w is None:
                return hip.gather_scatter_out(si, di, x, out)
            return hip.gather_weight_scatter_out(si, di, w, x, out)

        def run_plan():
            self.stats["plan_launches"] += 1
            if weight is None:
                return slab.slab_spmm_out(plan, None, 0, x, out, H, F, reduce)
            if via and getattr(weight[1], "base", None) is plan:
                return slab.slab_spmm_out(weight[1], weight[2].contiguous(), 2 if mh else 1, x, out, H, F, reduce, stage_weights=False)
            if po and weight[0] is plan:
                return slab.slab_spmm_out(plan, weight[1].contiguous(), 5 if mh else 4, x, out, H, F, reduce)
            return slab.slab_spmm_out(plan, edge_weight(), 2 if mh else 1, x, out, H, F, reduce)

        if self._use_plan(plan, "spmm", run_plan, run_edges):
            run_plan()
        else:
            run_edges()
        return out

    def _sddmm(self, m1: torch.Tensor, m2: torch.Tensor, plan_order: bool):
        """(plan | None, values): <m1[d_e], m2[s_e]> per edge (and head); values in the plan's order when a plan is returned."""
        m1, m2 = m1.contiguous(), m2.contiguous()
        mh = m1.dim() == 3
        H, F = (m1.shape[1], m1.shape[2]) if mh else (1, m1.shape[1])
        rowbytes = H * F * m1.element_size()
        shape = (self.nnz, H) if mh else (self.nnz,)
        out = torch.empty(shape, dtype=m1.dtype, device=m1.device)
        plan = self._plan("fwd", rowbytes, 2 if mh else 1, H, m1.dtype, _rows(m2))
        if plan is not None and (H not in (1, 2, 4, 8) or (H * m1.element_size()) not in (2, 4, 8, 16)):
            plan = None
        a, b = (m1, m2) if mh else (m1.unsqueeze(1), m2.unsqueeze(1))

        def run_edges():
            self.stats["edge_launches"] += 1
            if mh:
                return hip.mh_sddmm_coo_out(self.src_index, self.dst_index, m1, m2, out, False)
            return hip.sddmm_coo_out(self.src_index, self.dst_index, m1, m2, out)

        def run_plan(keep_plan_order=False):
            self.stats["plan_launches"] += 1
            staging = torch.empty((self.nnz, H), dtype=m1.dtype, device=m1.device)
            if keep_plan_order:
                return slab.slab_mh_sddmm_out(plan, a, b, None, staging).view(shape)
            slab.slab_mh_sddmm_out(plan, a, b, out.view(self.nnz, H), staging)
            return out

        if self._use_plan(plan, "sddmm", run_plan, run_edges):
            if plan_order:
                return plan, run_plan(True)
            return None, run_plan()
        run_edges()
        if plan_order and plan is not None:                # the caller asked for plan order: give it (one gather)
            return plan, _rows_at(out, self._perm64(plan))
        return None, out

    # ---- public, differentiable -------------------------------------------------------------------------------------------------------
    def gather_scatter(self, x: torch.Tensor, reduce: str = "sum") -> torch.Tensor:
        """dst[d] = reduce over the edges into d of x[s_e]  (geot.gather_scatter, geot/gather_scatter.py:7-9)."""
        return _SpmmFn.apply(self, None, None, x, _kind(reduce))

    def gather_weight_scatter(self, weight: Union[torch.Tensor, PlanOrdered], x: torch.Tensor, reduce: str = "sum") -> torch.Tensor:
        """dst[d] = reduce over the edges into d of w_e x[s_e]  (geot.gather_weight_scatter, geot/gather_weight_scatter.py:15-18)."""
        plan, values = _split(self, weight, 1)
        return _SpmmFn.apply(self, plan, values, x, _kind(reduce))

    def mh_spmm(self, weight: Union[torch.Tensor, PlanOrdered], x: torch.Tensor) -> torch.Tensor:
        """dst[d, h, :] = sum_e w[e, h] x[s_e, h, :]; weight [nnz, H] (edge order), [H, nnz], or PlanOrdered (geot/mh_spmm.py:4-6)."""
        if not isinstance(weight, PlanOrdered) and weight.dim() == 2 and weight.shape[0] != self.nnz and weight.shape[1] == self.nnz:
            weight = weight.t()
        plan, values = _split(self, weight, 2)
        if x.dim() != 3:
            raise RuntimeError("src must be 3 dimensional")                        # (csrc/cuda/mh_spmm_cuda.cu:29)
        if values.shape[1] != x.shape[1]:
            raise RuntimeError("Invalid weight size")                              # (csrc/cuda/wrapper/mh_spmm_base.h:49)
        return _SpmmFn.apply(self, plan, values, x, "sum")

    def sddmm(self, mat_1: torch.Tensor, mat_2: torch.Tensor, plan_order: bool = False):
        """out[e] = <mat_1[dst_e], mat_2[src_e]>  (geot.sddmm_coo_impl, geot/gather_weight_scatter.py:8-12); ``plan_order=True``:
        a PlanOrdered whose values never pass through the edge permutation (falls back to edge order, as a PlanOrdered over no
        plan, on graphs that are not dense enough for a plan)."""
        return self._sddmm_public(mat_1, mat_2, plan_order)

    def mh_sddmm(self, mat_1: torch.Tensor, mat_2: torch.Tensor, plan_order: bool = False):
        """out[e, h] = <mat_1[dst_e, h, :], mat_2[src_e, h, :]>, mat_* [rows, H, F]: attention scores / d/dweight of mh_spmm."""
        if mat_1.dim() != 3 or mat_2.dim() != 3:
            raise RuntimeError("mat_1 and mat_2 must be 3 dimensional with the same heads and feature dimensions")
        return self._sddmm_public(mat_1, mat_2, plan_order)

    def _sddmm_public(self, m1, m2, plan_order):
        holder = []
        values = _SddmmFn.apply(self, m1, m2, plan_order, holder)
        if plan_order:                                    # always a PlanOrdered (over no plan = edge order): the caller's code is one shape
            return PlanOrdered(self, holder[0], values)
        return values

    def plan_order(self, values: torch.Tensor, like: Union[PlanOrdered, torch.Tensor]):
        """Edge-order values -> plan order (one gather).  ``like``: a PlanOrdered (its plan), or the feature table a following
        ``gather_weight_scatter`` / ``mh_spmm`` will read ([N, F] / [N, H, F]) - the plan those calls use for rows of that shape and
        type.  The way to hand over coefficients that do not change between calls (a normalised adjacency, frozen attention): permute
        once, pass the PlanOrdered every call - the row loop then reads them as a stream.  Returns ``values`` unchanged when the graph
        has no plan for that shape (not dense enough, or a row width the source-blocked kernels do not serve)."""
        if isinstance(like, PlanOrdered):
            plan = like.plan
        else:
            mh = like.dim() == 3
            H, F = (like.shape[1], like.shape[2]) if mh else (1, like.shape[1])
            plan = self._plan("fwd", H * F * like.element_size(), 2 if mh else 1, H, like.dtype, _rows(like))
            if mh and ((F * like.element_size()) % 16 != 0 or H > 16):
                plan = None
            if plan is None:
                return values
        if plan is None:
            return PlanOrdered(self, None, values)
        return PlanOrdered(self, plan, _EdgeToPlan.apply(self, plan, values))


def _fit_rows(t: torch.Tensor, rows: int) -> torch.Tensor:
    """A gradient over the graph's row count brought to the row count of the tensor it belongs to: zero rows appended for trailing
    nodes without an edge, surplus rows (all zero by construction) dropped."""
    if t.shape[0] == rows:
        return t
    if t.shape[0] > rows:
        return t[:rows].contiguous()
    return torch.nn.functional.pad(t, [0, 0] * (t.dim() - 1) + [0, rows - t.shape[0]])


def _kind(reduce: str) -> str:
    from .ops import _aggr_kind
    return _aggr_kind(reduce)


def _split(g: Graph, weight, dims: int):
    if isinstance(weight, PlanOrdered):
        if weight.graph is not g:
            raise ValueError("PlanOrdered values of another Graph")
        if weight.values.dim() != dims or weight.values.shape[0] != g.nnz:
            raise RuntimeError("weight must be 1 dimensional with one value per edge" if dims == 1 else "Invalid weight size")
        return weight.plan, weight.values
    if weight.shape[0] != g.nnz or weight.dim() != dims:
        raise RuntimeError("weight must be 1 dimensional with one value per edge" if dims == 1 else "Invalid weight size")
    return None, weight


class _PlanToEdge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, plan, values):
        ctx.g, ctx.plan = g, plan
        return _rows_at(values, g._inverse(plan))

    @staticmethod
    def backward(ctx, grad):
        return None, None, _rows_at(grad, ctx.g._perm64(ctx.plan))


class _EdgeToPlan(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, plan, values):
        ctx.g, ctx.plan = g, plan
        return _rows_at(values, g._perm64(plan))

    @staticmethod
    def backward(ctx, grad):
        return None, None, _rows_at(grad, ctx.g._inverse(ctx.plan))


class _SpmmFn(torch.autograd.Function):
    """gather_scatter / gather_weight_scatter / mh_spmm over the handle's lists.  d/dx = the same operator over the transposed list
    with the weights in ITS order; d/dweight = the (multi-head) SDDMM, returned in the order the weight came in - a weight in plan
    order gets its gradient in plan order, straight from the persistent kernel."""

    @staticmethod
    def forward(ctx, g, plan, values, x, reduce):
        if reduce not in ("sum", "mean", "max", "min", "prod"):
            raise RuntimeError(f"reduce argument must be either sum, prod, mean, amax or amin, got {reduce}")
        if reduce != "sum" and x.dim() == 3:
            raise NotImplementedError("mh_spmm: only reduce='sum'")
        weight = None if values is None else ((plan, values.detach()) if plan is not None else values.detach())
        ctx.g, ctx.plan, ctx.reduce = g, plan, reduce
        ctx.save_for_backward(values if values is not None else x.new_empty(0), x)
        ctx.has_w = values is not None
        return g._spmm("fwd", weight, x.detach(), reduce)

    @staticmethod
    def backward(ctx, grad):
        values, x = ctx.saved_tensors
        gx, gw = _spmm_backward(ctx.g, ctx.plan, values if ctx.has_w else None, x, grad, ctx.reduce, ctx.needs_input_grad[3],
                                ctx.has_w and ctx.needs_input_grad[2])
        return None, None, gw, gx, None


def _spmm_backward(g: "Graph", plan, values, x, grad, reduce: str, need_x: bool, need_w: bool):
    """(d/dx, d/dweight) of out = g._spmm('fwd', weight, x): d/dx = the same operator over the transposed list with the weights read in
    ITS order; d/dweight = the (multi-head) SDDMM, returned in the order the weight came in (``plan``: plan order; None: edge order).
    Shared by the handle's autograd.Function and the ``geot::graph_spmm`` operator of a rewritten exported program."""
    if reduce not in ("sum", "mean"):
        raise NotImplementedError(f"geot_amd.Graph: backward is implemented for reduce='sum' and 'mean' only (got '{reduce}')")
    has_w = values is not None
    grad = grad.contiguous()
    if reduce == "mean":
        deg = g.degree().clamp(min=1.0).to(grad.dtype)
        grad = grad / deg.view(-1, *([1] * (grad.dim() - 1)))
    gx = gw = None
    if need_x:
        w_t = None
        if has_w:
            mh = x.dim() == 3
            H, F = (x.shape[1], x.shape[2]) if mh else (1, x.shape[1])
            bwd_plan = g._plan("bwd", H * F * x.element_size(), 2 if mh else 1, H, x.dtype, _rows(grad))
            if bwd_plan is not None and not g._verdict.get((id(bwd_plan), "spmm"), True):
                bwd_plan = None                       # (lost its trial: the per-edge kernels read the transposed list's order)
            vals = values.detach().contiguous()

            def list_order(vals=vals, plan=plan):       # the weights in the transposed LIST's order (per-edge kernels, trials)
                return hip.gather_rows_out(g._values_for_bwd(plan, None), vals, torch.empty_like(vals))
            w_t = ("via", g._bwd_reader(plan, bwd_plan), vals, list_order) if bwd_plan is not None else list_order()
        gx = _fit_rows(g._spmm("bwd", w_t, grad), x.shape[0])     # (x may have trailing rows no edge reads: src_rows = max(src_index) + 1)
    if has_w and need_w:
        wplan, gw = g._sddmm(grad, x.detach(), plan is not None)      # (plan order asked for when the weight came in plan order)
        if plan is not None and wplan is not plan:       # the scores came back in another order than the weight's: re-order once
            gw_edge = gw if wplan is None else _rows_at(gw, g._inverse(wplan))
            gw = _rows_at(gw_edge, g._perm64(plan))
        elif plan is None and wplan is not None:
            gw = _rows_at(gw, g._inverse(wplan))
    return gx, gw


class _SddmmFn(torch.autograd.Function):
    """values[e (, h)] = <m1[d_e], m2[s_e]>.  d/dm1 = the weighted gather over the forward list (weights = the incoming gradient, in
    the order it is in), d/dm2 = the same over the transposed list."""

    @staticmethod
    def forward(ctx, g, m1, m2, plan_order, holder):
        plan, values = g._sddmm(m1.detach(), m2.detach(), plan_order)
        holder.append(plan)
        ctx.g, ctx.plan = g, plan
        ctx.save_for_backward(m1, m2)
        return values

    @staticmethod
    def backward(ctx, grad):
        g, plan = ctx.g, ctx.plan
        m1, m2 = ctx.saved_tensors
        grad = grad.contiguous()
        g1 = g2 = None
        if ctx.needs_input_grad[1]:
            g1 = g._spmm("fwd", (plan, grad) if plan is not None else grad, m2.detach())
            g1 = _fit_rows(g1, m1.shape[0])                  # (more rows in m1 than index[-1] + 1: the tail has no edges)
        if ctx.needs_input_grad[2]:
            mh = m1.dim() == 3
            H, F = (m1.shape[1], m1.shape[2]) if mh else (1, m1.shape[1])
            bwd_plan = g._plan("bwd", H * F * m1.element_size(), 2 if mh else 1, H, m1.dtype, _rows(m1))
            if bwd_plan is not None and not g._verdict.get((id(bwd_plan), "spmm"), True):
                bwd_plan = None
            def list_order(grad=grad, plan=plan):
                return hip.gather_rows_out(g._values_for_bwd(plan, None), grad, torch.empty_like(grad))
            w_t = ("via", g._bwd_reader(plan, bwd_plan), grad, list_order) if bwd_plan is not None else list_order()
            m1d = m1.detach()
            if m1d.shape[0] < g.rows:
                m1d = torch.nn.functional.pad(m1d, [0, 0] * (m1d.dim() - 1) + [0, g.rows - m1d.shape[0]])
            g2 = g._spmm("bwd", w_t, m1d)
            g2 = _fit_rows(g2, m2.shape[0])
        return None, g1, g2, None, None


# ---- handles inside an exported program (geot_amd.match_replace.pattern_transform(..., static_graph=True)) -------------------------------
# An FX graph carries tensors and numbers, not Python objects: the rewritten program names its Graph by an integer, and two operators
# defined here (where the reference defines its own, geot/gather_weight_scatter.py:15-51: torch.library.custom_op + register_fake +
# register_autograd) run the handle's kernels - no index tensors in the call, so no fingerprint, no cache lookup, no row rule.
_HANDLES: dict = {}
_NEXT_HANDLE = [1]


def register_graph(g: Graph) -> int:
    """Keep ``g`` alive under a fresh integer (what ``geot::graph_spmm`` takes); ``release_graph`` drops it."""
    h = _NEXT_HANDLE[0]
    _NEXT_HANDLE[0] += 1
    _HANDLES[h] = g
    return h


def release_graph(handle: int) -> None:
    _HANDLES.pop(int(handle), None)


def _handle(handle: int) -> Graph:
    g = _HANDLES.get(int(handle))
    if g is None:
        raise RuntimeError(f"geot::graph_spmm: no geot_amd.Graph is registered under handle {handle} (released, or another process)")
    return g


@torch.library.custom_op("geot::graph_spmm", mutates_args=())
def graph_spmm(handle: int, weight: Optional[torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """dst[d] = sum over the registered graph's edges into d of (w_e) x[s_e]; x [N, F] with weight None | [nnz], or [N, H, F] with
    weight [nnz, H] (gather_scatter / gather_weight_scatter / mh_spmm over a static graph)."""
    return _handle(handle)._spmm("fwd", None if weight is None else weight.contiguous(), x)


@graph_spmm.register_fake
def _(handle, weight, x):
    return x.new_empty((_handle(handle).rows,) + tuple(x.shape[1:]))


@torch.library.custom_op("geot::graph_spmm_backward", mutates_args=())
def graph_spmm_backward(handle: int, weight: Optional[torch.Tensor], x: torch.Tensor, grad: torch.Tensor, need_x: bool,
                        need_w: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    gx, gw = _spmm_backward(_handle(handle), None, weight, x, grad, "sum", need_x, need_w and weight is not None)
    return (gx if gx is not None else x.new_empty(0)), (gw if gw is not None else x.new_empty(0))


@graph_spmm_backward.register_fake
def _(handle, weight, x, grad, need_x, need_w):
    return (x.new_empty(x.shape) if need_x else x.new_empty(0)), (weight.new_empty(weight.shape) if (need_w and weight is not None) else x.new_empty(0))


def _graph_spmm_setup(ctx, inputs, output):
    handle, weight, x = inputs
    ctx.handle, ctx.has_w = handle, weight is not None
    ctx.save_for_backward(weight if weight is not None else x.new_empty(0), x)


def _graph_spmm_bwd(ctx, grad):
    weight, x = ctx.saved_tensors
    need_x, need_w = ctx.needs_input_grad[2], ctx.has_w and ctx.needs_input_grad[1]
    gx, gw = torch.ops.geot.graph_spmm_backward(ctx.handle, weight if ctx.has_w else None, x, grad, need_x, need_w)
    return None, (gw if need_w else None), (gx if need_x else None)


torch.library.register_autograd("geot::graph_spmm", _graph_spmm_bwd, setup_context=_graph_spmm_setup)
