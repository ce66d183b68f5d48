"""Row-sharded full-graph encoder across the GPUs of one node (SURVEY.md 8e "destination-row
partition"; the reference itself is single-device, main.py:71-72, and recomputes the whole-graph
encoder every step, model.py:150-151).

Rank r owns the destination rows [r*S, (r+1)*S), S = ceil(N / world) (rounded up to 4), of
  * the CSR (its rows list sources anywhere in the graph),
  * the embedding table and its Adam state,
  * every layer's output.
One training step on rank r (plnlp_amd.BaseModel.train_step_sharded):

  forward   layer l:  y_r = conv(x_full, A_r)          aggregation + GEMM over S rows only
            between layers: x_full = all-gather(y_r)   (AllGatherRows; backward = reduce-scatter)
            scorer: rank r scores ITS slice of the global edge batch and needs h only at the nodes
            that slice touches: one all-to-all moves exactly those rows (ExchangeRows) -- at
            B = 65 536 over 8 ranks that is ~30 K rows per rank instead of the whole [N, h] matrix
  backward  the row gradients travel back through the same all-to-all and are added per owner in
            rank order (fixed order -> reproducible); each layer's input gradient is a partial sum
            over all N rows and is reduce-scattered to the row owners; the few small weights are
            SUM all-reduced (they stay bit-identical on every rank)
  update    fused Adam on the small weights (replicated) and on the OWNED embedding rows only,
            then one in-place all-gather refreshes every rank's copy of the table.

What every rank needs to know about the OTHER ranks' slices (which of my rows they will ask for) it
computes locally: all ranks hold the same global batch (same seeds), so the request lists need no
exchange of their own (ShardPlan) -- the only host read-back of a step is the [world, world] table of
row counts that sizes the all-to-all.

Everything here is torch + torch.distributed (RCCL on the GPU box, gloo in the CPU tests); the
arithmetic stays in the HIP kernels (ops.SAGEConvBlockFn, the fused scorer)."""
from __future__ import annotations

from typing import List, Optional

import math

import torch
import torch.distributed as dist


class RowPartition:
    """contiguous, equal-height row blocks; the last ones may be partly (or wholly) padding"""

    def __init__(self, n: int, world: int, rank: int):
        self.n, self.world, self.rank = int(n), int(world), int(rank)
        s = (self.n + self.world - 1) // self.world
        self.rows = max(4, (s + 3) // 4 * 4)          # S: multiple of 4 keeps every block 16-byte aligned
        self.padded = self.rows * self.world
        self.lo = self.rank * self.rows
        self.owned = max(0, min(self.n, self.lo + self.rows) - self.lo)     # real rows in this block

    def __repr__(self):
        return f"RowPartition(n={self.n}, world={self.world}, rank={self.rank}, rows={self.rows})"


class _AllGatherRows(torch.autograd.Function):
    """[S, F] block of every rank -> [world * S, F]; backward: reduce-scatter of the (partial-sum)
    gradient to the row owners"""

    @staticmethod
    def forward(ctx, block, group):
        ctx.group = group
        block = block.contiguous()
        world = dist.get_world_size(group)
        out = torch.empty(world * block.shape[0], block.shape[1], dtype=block.dtype, device=block.device)
        dist.all_gather_into_tensor(out, block, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        world = dist.get_world_size(ctx.group)
        g = g.contiguous()
        out = torch.empty(g.shape[0] // world, g.shape[1], dtype=g.dtype, device=g.device)
        dist.reduce_scatter_tensor(out, g, group=ctx.group)
        return out, None


class _ShardedLeaf(torch.autograd.Function):
    """the full (already synchronised) table as a function of the OWNED rows: forward hands out the
    replica, backward reduce-scatters the partial gradient of all rows to the owners"""

    @staticmethod
    def forward(ctx, shard_param, full, group):
        ctx.group = group
        ctx.rows = shard_param.shape[0]
        return full.view_as(full)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        out = torch.empty(ctx.rows, g.shape[1], dtype=g.dtype, device=g.device)
        dist.reduce_scatter_tensor(out, g, group=ctx.group)
        return out, None, None


class ShardContext:
    """what the sharded encoder / trainer need of the process group and the partition"""

    def __init__(self, group, partition: RowPartition):
        self.group, self.part = group, partition
        self.rank, self.world = partition.rank, partition.world
        self.lo, self.rows = partition.lo, partition.rows

    def all_gather(self, block: torch.Tensor) -> torch.Tensor:
        return _AllGatherRows.apply(block, self.group)

    def leaf(self, shard_param: torch.Tensor, full: torch.Tensor) -> torch.Tensor:
        return _ShardedLeaf.apply(shard_param, full, self.group)

    @torch.no_grad()
    def sync_table(self, full: torch.Tensor, shard: torch.Tensor):
        """full[r*S:(r+1)*S] = rank r's shard, for every r.  `shard` is this rank's block of `full`
        (same storage): on RCCL the all-gather runs in place (the input is the output's own slice at the
        rank offset, the in-place form the collective defines); gloo gets a copy of the block.
        Returns the pending work: the table's readers wait for it (the next step's first aggregation --
        its edge pre-processing, which does not read the table, overlaps the transfer)."""
        src = shard.detach()
        if not full.is_cuda:
            src = src.clone()
        return dist.all_gather_into_tensor(full, src, group=self.group, async_op=True)


_pinned = {"ring": None, "next": 0}


def _pinned_i64(n: int) -> torch.Tensor:
    """pinned staging for the per-step count table (a small ring: a few plans may be in flight)"""
    if _pinned["ring"] is None or _pinned["ring"].shape[1] < n:
        _pinned["ring"] = torch.empty(16, max(n, 128), dtype=torch.int64, pin_memory=True)
        _pinned["next"] = 0
    i = _pinned["next"]
    _pinned["next"] = (i + 1) % 16
    return _pinned["ring"][i, :n]


class ShardPlan:
    """Who needs which rows for ONE global edge batch.

    Rank q scores the positives [q*per, (q+1)*per) and their negatives; U_q = the distinct endpoints of
    that slice, in increasing node order, is the row list rank q gathers into its compact matrix
    H_q [|U_q|, h].  Every rank holds the whole batch, so every rank can tell which of ITS rows each
    other rank asks for -- flags over the (slice, node) pairs, a prefix sum and one [world, world]
    count table (the step's one host read-back), no request messages.

      src_c / dst_c : the slice's edges in compact coordinates (row of H_q), positives then negatives
      send_rows     : block-local ids of my rows the ranks 0 .. W-1 ask for, grouped by asker
      in_splits     : rows I send to each rank          out_splits : rows each rank sends me
      count         : |U_me|"""

    def __init__(self, part: RowPartition, pos_edge: torch.Tensor, neg_flat: torch.Tensor, num_neg: int, per: int,
                 stream=None):
        """phase 1: everything that depends on the batch alone, enqueued on `stream` (the trainer's side
        stream, one batch AHEAD of the step that uses it) or on the current stream, ending with the
        asynchronous copy of the count table to pinned memory.  finish() picks the table up."""
        W, S, me = part.world, part.rows, part.rank
        dev = pos_edge.device
        self._part, self._k, self._n, self._dev = part, int(num_neg), pos_edge.size(0), dev
        n, k = self._n, self._k
        span = W * S                                      # one slice's id space (node ids < N <= W * S)
        lo, hi = min(me * per, n), min((me + 1) * per, n)
        self.lo, self.hi, self.local = lo, hi, hi - lo
        self._stream = stream if dev.type == "cuda" else None
        self._finished = False
        self._joined = None
        ctx = torch.cuda.stream(self._stream) if self._stream is not None else _NullCtx()
        with ctx:
            # slice of every edge of the batch: positives, then the k negatives of each positive
            sid_pos = torch.arange(n, device=dev, dtype=torch.int64).div_(max(per, 1), rounding_mode="floor")
            sid = torch.cat([sid_pos, sid_pos.repeat_interleave(k)]) if k > 0 else sid_pos
            src = torch.cat([pos_edge[:, 0], neg_flat[:, 0]]).to(torch.int64)
            dst = torch.cat([pos_edge[:, 1], neg_flat[:, 1]]).to(torch.int64)
            base = sid * span
            vid = torch.cat([base + src, base + dst])          # (slice, node) pairs touched by the batch
            flags = torch.zeros(W * span, dtype=torch.int32, device=dev)
            flags.index_fill_(0, vid, 1)
            csum = torch.cumsum(flags, 0, dtype=torch.int64)
            cnt = flags.view(W, W, S).sum(dim=-1, dtype=torch.int64)          # [asker, owner]
            off = torch.zeros(W * W + 1, dtype=torch.int64, device=dev)
            torch.cumsum(cnt.reshape(-1), 0, out=off[1:])
            cap = min(vid.numel(), W * span)
            rows_v = torch.empty(max(cap, 1), dtype=torch.int64, device=dev)
            pos_v = csum - 1                                    # compact position of a flagged pair
            rows_v.index_copy_(0, pos_v.index_select(0, vid), vid)            # duplicates write the same value
            # rows of MY block that ANY slice touches: the only rows of the encoder output anybody reads (and the only
            # ones whose gradient is not zero) -- the row-sparse last layer of the sharded encoder computes just these
            mine_any = flags.view(W, W, S)[:, me, :].amax(dim=0)                      # [S]
            self._any_csum = torch.cumsum(mine_any, 0, dtype=torch.int64)
            self._any = mine_any
            off = torch.cat([off, self._any_csum[-1:]])                              # one more count for the host
            self._src, self._dst, self._rows_v, self._pos_v, self._off = src, dst, rows_v, pos_v, off
            # ---- the one host read-back: the count table, copied asynchronously
            self._host = self._ev = None
            if dev.type == "cuda":
                self._host = _pinned_i64(off.numel())
                self._host.copy_(off, non_blocking=True)
                self._ev = torch.cuda.Event()
                self._ev.record()

    def finish(self, build_incidence: bool = False) -> "ShardPlan":
        """phase 2: wait for the count table (host), then cut the lists this rank needs -- on the plan's
        stream; join() makes the consuming stream wait for them.
        build_incidence: also build the node-sorted incidence list of the slice in compact coordinates
        (the deterministic backward of the fused scorers, ops.prepare_edge_backward)."""
        if self._finished:
            return self
        part = self._part
        W, S, me = part.world, part.rows, part.rank
        n, k = self._n, self._k
        if self._ev is not None:
            self._ev.synchronize()
            offs = self._host.tolist()
        else:
            offs = self._off.tolist()
        span = W * S
        self.block_count = offs[W * W + 1]           # rows of my block touched by the global batch
        self.count = offs[(me + 1) * W] - offs[me * W]
        self.out_splits = [offs[me * W + r + 1] - offs[me * W + r] for r in range(W)]
        self.in_splits = [offs[q * W + me + 1] - offs[q * W + me] for q in range(W)]
        ctx = torch.cuda.stream(self._stream) if self._stream is not None else _NullCtx()
        with ctx:
            rows_v, pos_v, src, dst = self._rows_v, self._pos_v, self._src, self._dst
            segs = [rows_v[offs[q * W + me]:offs[q * W + me + 1]] for q in range(W)]
            self.send_rows = (torch.cat(segs) if W > 1 else segs[0]).remainder(S)
            # ---- the touched rows of my block as a compact row set (global ids, increasing), padded to 32 rows
            # (at least one group of 32: a rank whose block no edge of the batch touches still launches the layer -- over
            # padding rows only -- instead of handing zero-row operands to every kernel of it; found by the world-2 HIP run,
            # tests/test_hip_multirank.py::empty_slice_shard)
            pad = min(part.padded, max(32, (self.block_count + 31) // 32 * 32))
            local = torch.nonzero(self._any).reshape(-1)                               # block-local ids, sorted
            rows_g = torch.zeros(max(pad, 1), dtype=torch.int32, device=self._dev)
            rows_g[:self.block_count] = (local + part.lo).to(torch.int32)
            node_map = torch.full((part.padded,), -1, dtype=torch.int32, device=self._dev)
            node_map[part.lo:part.lo + S] = torch.where(self._any > 0, (self._any_csum - 1).to(torch.int32),
                                                        torch.full_like(self._any, -1))
            self.block_rows = BlockRows(rows_g[:pad], node_map, self.block_count, pad)
            self.send_pos = node_map.index_select(0, self.send_rows + part.lo).long()  # where a sent row sits in it
            # gradient rows coming back through the all-to-all are added per compact row in asker order: the lists
            # of one asker are distinct rows and the askers are concatenated in rank order, so a STABLE sort by row
            # gives that fixed order; as a CSR it is ONE deterministic aggregation launch (no per-asker loop)
            order = torch.argsort(self.send_pos, stable=True)
            seg = torch.zeros(pad + 1, dtype=torch.int64, device=self._dev)
            if self.send_pos.numel():
                torch.cumsum(torch.bincount(self.send_pos, minlength=pad), 0, out=seg[1:])
            self.back_csr = (seg, order.to(torch.int32))
            # ---- my slice in compact coordinates
            lo, hi = self.lo, self.hi
            mine = me * span
            first = offs[me * W]
            src_m = torch.cat([src[lo:hi], src[n + lo * k:n + hi * k]])
            dst_m = torch.cat([dst[lo:hi], dst[n + lo * k:n + hi * k]])
            self.src_c = pos_v.index_select(0, src_m + mine) - first
            self.dst_c = pos_v.index_select(0, dst_m + mine) - first
            self.rows = rows_v[first:first + self.count] - mine      # global node id of every row of H_me (sorted)
            self.incidence = None
            if build_incidence and self.local > 0 and self._dev.type == "cuda":
                from . import ops
                self.incidence = ops.prepare_edge_backward(self.src_c, self.dst_c, max(self.count, 1), compact=False)
            if self._stream is not None:
                self._joined = torch.cuda.Event()
                self._joined.record()
        self._src = self._dst = self._rows_v = self._pos_v = self._off = self._any = self._any_csum = None
        self._finished = True
        return self

    def join(self, record_streams: bool = True) -> "ShardPlan":
        """the current stream waits for the lists (no-op when they were built on it, or when they are already
        complete -- started a batch ahead they usually are).  record_streams=False: the caller keeps this plan
        alive until the consuming step has finished on the device (ops.StepThrottle) instead of telling the
        allocator, which would record one event per tensor on the consuming stream when the plan dies."""
        if self._joined is not None:
            cur = torch.cuda.current_stream(self._dev)
            if not self._joined.query():
                cur.wait_event(self._joined)
            if record_streams:
                for t in (self.send_rows, self.src_c, self.dst_c, self.rows, self.send_pos, self.block_rows.rows,
                          self.block_rows.node_map, self.back_csr[0], self.back_csr[1]):
                    t.record_stream(cur)
                inc = self.incidence
                if inc is not None:
                    for name in ("item_edge", "item_other", "seg_ptr"):
                        getattr(inc, name).record_stream(cur)
                    if getattr(inc, "_split", None) is not None:
                        inc._split._buf.record_stream(cur)
            self._joined = None
        return self


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class BlockRows:
    """the rows of this rank's block that the global batch touches, as ops.SAGEConvFn's `out_rows` wants them:
    rows (int32 global ids, increasing, padded with 0 to n_rows), node_map (int32 [N padded]: compact position or
    -1), count"""
    __slots__ = ("rows", "node_map", "count", "n_rows")

    def __init__(self, rows, node_map, count, n_rows):
        self.rows, self.node_map, self.count, self.n_rows = rows, node_map, int(count), int(n_rows)


class _BackRows:
    """the gradient rows returned by the all-to-all as a CSR over compact rows (ShardPlan.back_csr): what
    ops.csr_aggregate walks to add them in fixed order"""

    def __init__(self, seg, order, n_src):
        self.rowptr, self.col, self.val, self.val_index = seg, order, None, None
        self.n_rows, self.n_cols = seg.numel() - 1, int(n_src)
        self._split = None

    def row_split(self, threshold: int):
        if self._split is None:
            from .graph import RowSplit
            self._split = RowSplit(self.rowptr, self.col.numel(), threshold)
        return self._split


class ExchangeCompactRows(torch.autograd.Function):
    """ExchangeRows over a ROW-RESTRICTED encoder output: h_c [R, h] holds only the rows of this rank's block that
    the global batch touches (plan.block_rows); the rows each asker wants are gathered from it by position
    (plan.send_pos).  Backward: the returned gradient rows are added per compact row in asker order by ONE
    deterministic segmented aggregation and handed to the conv's backward ROW-SPARSE through `channel` -- the dense
    [S, h] gradient block of the plain form (and its zero fill, and its per-asker index_add_ launches) never exists."""

    @staticmethod
    def forward(ctx, h_c, plan: "ShardPlan", group, channel):
        ctx.plan, ctx.group, ctx.channel = plan, group, channel
        ctx.set_materialize_grads(False)
        send = h_c.index_select(0, plan.send_pos)
        recv = torch.empty(plan.count, h_c.shape[1], dtype=h_c.dtype, device=h_c.device)
        dist.all_to_all_single(recv, send, output_split_sizes=plan.out_splits, input_split_sizes=plan.in_splits,
                               group=group)
        return recv

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        br = plan.block_rows
        if g is None:
            return None, None, None, None
        g = g.contiguous()
        back = torch.empty(plan.send_pos.numel(), g.shape[1], dtype=g.dtype, device=g.device)
        dist.all_to_all_single(back, g, output_split_sizes=plan.in_splits, input_split_sizes=plan.out_splits,
                               group=ctx.group)
        if back.is_cuda and back.dtype == torch.float32 and back.shape[0] > 0:
            from . import ops
            vals = ops.csr_aggregate(_BackRows(plan.back_csr[0], plan.back_csr[1], back.shape[0]), back, "sum", False)
        else:                                   # host path of the gloo tests: the same fixed order, spelled out
            vals = torch.zeros(br.n_rows, g.shape[1], dtype=g.dtype, device=g.device)
            at = 0
            for c in plan.in_splits:
                if c:
                    vals.index_add_(0, plan.send_pos[at:at + c], back[at:at + c])
                at += c
        from .ops import RowSparseGrad
        ctx.channel.grad = RowSparseGrad(br.rows, br.node_map, vals, br.node_map.numel(), br.count)
        return None, None, None, None


class ExchangeRows(torch.autograd.Function):
    """H_me = rows U_me of the (row-sharded) encoder output, gathered from their owners by ONE
    all-to-all; backward: the row gradients return the same way and are added to the owner's block in
    asker order (each asker's list has distinct rows, the askers are taken one after the other: a
    fixed summation order)."""

    @staticmethod
    def forward(ctx, h_block, plan: ShardPlan, group):
        ctx.plan, ctx.group, ctx.block_shape = plan, group, h_block.shape
        send = h_block.index_select(0, plan.send_rows)
        recv = torch.empty(plan.count, h_block.shape[1], dtype=h_block.dtype, device=h_block.device)
        dist.all_to_all_single(recv, send, output_split_sizes=plan.out_splits, input_split_sizes=plan.in_splits,
                               group=group)
        return recv

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        g = g.contiguous()
        back = torch.empty(plan.send_rows.numel(), g.shape[1], dtype=g.dtype, device=g.device)
        dist.all_to_all_single(back, g, output_split_sizes=plan.in_splits, input_split_sizes=plan.out_splits,
                               group=ctx.group)
        out = torch.zeros(ctx.block_shape, dtype=g.dtype, device=g.device)
        at = 0
        for c in plan.in_splits:                       # asker 0, 1, ..: distinct rows within one asker
            if c:
                out.index_add_(0, plan.send_rows[at:at + c], back[at:at + c])
            at += c
        return out, None, None


def allreduce_sum(tensors: List[torch.Tensor], group) -> None:
    """SUM all-reduce of the (small, replicated) weight gradients, in list order on every rank"""
    works = [dist.all_reduce(t, group=group, async_op=True) for t in tensors]
    for w in works:
        w.wait()


class RowShardedSAGEForward:
    """Forward-only SAGE encoder over a destination-row partition -- the layout of BASELINE.json config 5
    (R-MAT 50 M nodes / 1 B edges, h = 512: a training replica of it does not exist, X alone is 102 GB;
    SURVEY.md 8e).  Rank r holds ONLY its CSR block (rows [lo, lo + S), sources anywhere) and a full replica of
    the source matrix X [W * S, F]; every layer computes the rank's S output rows

        y_r = act( mean_agg_block(X) @ Wl^T + b + X[lo : lo + S] @ Wr^T )        (plnlp/layer.py:18-36)

    and between layers ONE all-gather writes the blocks back into X's own storage (the layer's output replaces
    its input: no second 102 GB matrix).  relu after every layer but the last (layer.py:20-23; dropout is
    inference-off).  Same kernels as the training path (ops.csr_aggregate, ops.gemm concat-K).
    layers: [(Wl [out, in], b [out], Wr [out, in]), ...] with in == out == X's width."""

    def __init__(self, block, part: RowPartition, layers, group=None):
        from . import ops, _lib
        self.block, self.part, self.layers, self.group = block, part, layers, group
        self._ops, self._lib = ops, _lib
        f = layers[0][0].shape[1]
        for wl, b, wr in layers:
            assert wl.shape == (f, f) and wr.shape == (f, f) and b.shape == (f,), "in == out == width of X"
        dev = layers[0][0].device
        self.agg = torch.empty(part.rows, f, dtype=torch.float32, device=dev)
        self.out = [torch.empty(part.rows, f, dtype=torch.float32, device=dev) for _ in range(min(2, len(layers)))]

    def layer(self, x_full, i: int, out: torch.Tensor) -> torch.Tensor:
        ops, part = self._ops, self.part
        wl, b, wr = self.layers[i]
        last = i == len(self.layers) - 1
        ops.csr_aggregate(self.block, x_full, "mean", False, out=self.agg)
        return ops.gemm([(self.agg, wl), (x_full[part.lo:part.lo + part.rows], wr)], False, True, out=out,
                        epilogue=self._lib.make_epilogue(bias=b, relu=not last))

    def forward(self, x_full: torch.Tensor) -> torch.Tensor:
        """x_full [W * S, F] is OVERWRITTEN by the intermediate layers' outputs; returns this rank's rows of
        the last layer"""
        part = self.part
        assert x_full.shape[0] == part.padded and x_full.is_contiguous()
        y = None
        for i in range(len(self.layers)):
            y = self.layer(x_full, i, self.out[i % len(self.out)])
            if i + 1 < len(self.layers):
                if self.group is not None:
                    dist.all_gather_into_tensor(x_full, y, group=self.group)
                else:
                    x_full[part.lo:part.lo + part.rows].copy_(y)
        return y


def collective_self_test(group, device) -> dict:
    """Every collective the data-parallel forms use (all-reduce, broadcast, all-gather, reduce-scatter, uneven
    all-to-all), once, on a tiny tensor with a known answer -- run at start-up, BEFORE the first training step,
    so that a broken fabric / environment (HSA_ENABLE_IPC_MODE_LEGACY, a missing peer) shows up as a named
    failure on every rank instead of as a hang inside step 1.  Returns {collective: True}; raises RuntimeError
    naming the first collective whose result is wrong.  Cheap (six tiny collectives)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    f = torch.float32
    done = {}

    def check(name, ok):
        if not bool(ok):
            raise RuntimeError(f"collective self-test: {name} returned a wrong result on rank {rank} of {world}")
        done[name] = True

    t = torch.full((4,), float(rank + 1), dtype=f, device=device)
    dist.all_reduce(t, group=group)
    check("all_reduce_sum", torch.all(t == world * (world + 1) / 2))
    t = torch.full((4,), float(rank + 1), dtype=f, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    check("all_reduce_max", torch.all(t == world))
    t = torch.full((3,), 7.0 if rank == 0 else -1.0, dtype=f, device=device)
    dist.broadcast(t, dist.get_global_rank(group, 0), group=group)
    check("broadcast", torch.all(t == 7.0))
    mine = torch.full((2, 3), float(rank), dtype=f, device=device)
    full = torch.empty(2 * world, 3, dtype=f, device=device)
    dist.all_gather_into_tensor(full, mine, group=group)
    want = torch.arange(world, dtype=f, device=device).repeat_interleave(2).unsqueeze(1).expand(-1, 3)
    check("all_gather_into_tensor", torch.equal(full, want))
    part = torch.arange(world, dtype=f, device=device).repeat_interleave(2).unsqueeze(1).expand(-1, 3).contiguous() + rank
    out = torch.empty(2, 3, dtype=f, device=device)
    dist.reduce_scatter_tensor(out, part, group=group)
    check("reduce_scatter_tensor", torch.all(out == world * rank + world * (world - 1) / 2))
    # uneven all-to-all: rank r sends (q + 1) rows holding r * 100 + q to every rank q
    send = torch.cat([torch.full((q + 1, 2), float(rank * 100 + q), dtype=f, device=device) for q in range(world)])
    recv = torch.empty(world * (rank + 1), 2, dtype=f, device=device)
    dist.all_to_all_single(recv, send, output_split_sizes=[rank + 1] * world,
                           input_split_sizes=[q + 1 for q in range(world)], group=group)
    want = torch.cat([torch.full((rank + 1, 2), float(q * 100 + rank), dtype=f, device=device) for q in range(world)])
    check("all_to_all_single_uneven", torch.equal(recv, want))
    dist.barrier(group=group)
    done["barrier"] = True
    return done


def cost_model(*, n_nodes: int, emb_width: int, hidden: int, param_bytes_small: int, batch_per_rank: int, num_neg: int,
               world: int, step_ms_1gpu: float, table_adam_ms: float, scorer_has_params: bool,
               bus_GBps: float = 300.0, step_ms_1rank: Optional[dict] = None) -> dict:
    """PREDICTED ms per step of the three exchange forms at `world` ranks (weak scaling: every rank brings
    batch_per_rank positives) -- a model to hold a measured SCALE run against, not a measurement.
    Inputs that are MEASUREMENTS of the run that asks (bench.py measures them on every rank before it chooses a form):
    step_ms_1gpu (the plain one-process step), table_adam_ms (Adam over the whole embedding table), and step_ms_1rank
    {form: ms} -- each form's step through a ONE-rank process group, i.e. with its whole host / plan / compaction /
    separate-optimiser overhead and every collective degenerate.  At world = 1 the model returns exactly those
    measurements (round 3's model predicted 1.58 / 1.69 ms for `shard` / `grads` where the same run measured 2.44 /
    1.75: the forms' fixed overheads were missing and its inputs were constants from another run).
    Assumptions beyond the measurements: RCCL ring collectives at `bus_GBps` bus bandwidth on the xGMI mesh (all-gather /
    reduce-scatter move bytes * (W-1)/W per rank, all-reduce twice that); a step grows by 17 % per extra batch-equivalent
    when ONE GPU back-propagates W batches (measured on collab: 3.49 ms at 8x vs 1.60 ms); half of `grads`' all-reduce
    and a quarter of `shard`'s table traffic hide behind compute."""
    W = max(1, int(world))
    table = n_nodes * emb_width * 4
    frac = (W - 1) / W
    ag = table * frac / (bus_GBps * 1e9) * 1e3
    one = dict(step_ms_1rank or {})
    # without per-form measurements: the one-process step plus what each form is known to add on one rank
    one.setdefault("grads", step_ms_1gpu + 0.6 * table_adam_ms)
    one.setdefault("scores", step_ms_1gpu + 0.75 * table_adam_ms)
    one.setdefault("shard", step_ms_1gpu + 0.86)
    out = {}
    # replicated encoder, SUM all-reduce of every gradient (the table's Adam is its own launch: the in-epilogue update of
    # the one-process step needs the REDUCED gradient -- that cost is inside the 1-rank measurement)
    out["grads"] = one["grads"] + 2 * (table + param_bytes_small) * frac / (bus_GBps * 1e9) * 1e3 * 0.5
    # replicated encoder, every rank back-propagates the global batch (parameter-free scorer only)
    if not scorer_has_params:
        out["scores"] = one["scores"] * (1.0 + 0.17 * (W - 1))
    # row-sharded encoder: blocks of rows, table all-gather + gradient reduce-scatter, touched rows by all-to-all.
    # Of the 1-rank step, the plan / exchange / compaction overhead (what it costs beyond the plain step) and 0.3 ms
    # of launches do not shrink with W; the rest is divided by W
    touched = n_nodes * (1.0 - math.exp(-2.0 * batch_per_rank * (1 + num_neg) / n_nodes))      # per slice
    rows_ms = 2 * touched * hidden * 4 * frac / (bus_GBps * 1e9) * 1e3
    fixed = min(one["shard"], max(0.0, one["shard"] - step_ms_1gpu) + min(0.3, 0.5 * step_ms_1gpu))
    out["shard"] = fixed + (one["shard"] - fixed) * (1.0 + 0.17 * (W - 1)) / W + 2 * ag * 0.75 + rows_ms
    best = min(out, key=out.get)
    return {"ms_per_step_predicted": out, "choice": best, "world": W,
            "inputs": {"step_ms_1gpu": step_ms_1gpu, "table_adam_ms": table_adam_ms,
                       "step_ms_1rank": {k: one[k] for k in out}, "measured_1rank_forms": sorted(step_ms_1rank or {})},
            "assumptions": "ring collectives at %.0f GB/s bus bandwidth; see plnlp_amd/shard.py::cost_model" % bus_GBps}
