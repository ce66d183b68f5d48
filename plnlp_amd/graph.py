"""Graph -- the adjacency object of the MI355X path.

It stands where `torch_sparse.SparseTensor` stands in the reference
(`data.adj_t`: main.py:81-83,110,124,136,186; utils.py:83-97) and exposes the
members those call sites touch (`coo`, `size`, `sum(dim=1)`, `set_diag`,
`set_value`, `to_symmetric`, `to`, `t`).  Layout in HBM, fixed for the whole
run (the graph is static across training steps):

    rowptr  int64 [n_rows+1]      row i lists the SOURCES j of messages into i
    col     int32 [nnz]           sorted by (row, col); duplicates kept
    val     fp32  [nnz] | None

plus, built once on first use, the transposed CSR (`t()`), which the backward
pass of the aggregation kernel walks -- so backward is a gather too and needs
no atomics -- and `inv_deg = 1/max(rowlen,1)`.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


class RowSplit:
    """Tables for plnlp_row_split (include/plnlp_hip.h): which rows are longer than
    `threshold` edges and how they are cut into chunks.  Built on the device by
    plnlp_row_split_build with capacities that are upper bounds (a CSR with nnz
    entries has at most nnz/threshold long rows), so no host sync is needed; idle
    slots hold -1.  `exact=True` (static graphs, built once) reads the counters back
    once to shrink the launch grids to the used sizes."""

    def __init__(self, rowptr: torch.Tensor, nnz: int, threshold: int = 256, exact: bool = False):
        from . import _lib as L
        lib = L.load()
        L.require_device(rowptr)
        self.threshold = int(threshold)
        dev = rowptr.device
        n_rows = rowptr.numel() - 1
        self.n_long = max(1, min(n_rows, nnz // threshold))
        self.n_chunks = max(1, nnz // threshold + self.n_long)
        # one buffer [counters(4 i64) | long_rows | chunk_beg | chunk_cnt | chunk_long]: cleared by one fill
        nl, nc = self.n_long, self.n_chunks
        nbytes = 32 + 20 * nl + 4 * nc
        buf = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev)
        raw = buf.view(torch.uint8)
        self._buf = buf
        self.counters = buf[:4]
        self.long_rows = buf[4:4 + nl]
        self.chunk_beg = buf[4 + nl:4 + 2 * nl]
        self.chunk_cnt = raw[32 + 16 * nl:32 + 20 * nl].view(torch.int32)
        self.chunk_long = raw[32 + 20 * nl:32 + 20 * nl + 4 * nc].view(torch.int32)
        L.check(lib.plnlp_row_split_build(rowptr.data_ptr(), n_rows, self.threshold, self.n_long, self.n_chunks,
                                          self.long_rows.data_ptr(), self.chunk_beg.data_ptr(),
                                          self.chunk_cnt.data_ptr(), self.chunk_long.data_ptr(),
                                          self.counters.data_ptr(), L.stream_ptr()), "plnlp_row_split_build")
        if exact:
            used_long, used_chunks, overflow = (int(v) + 1 for v in self.counters[:3].tolist())
            assert overflow == 0
            self.n_long, self.n_chunks = used_long, used_chunks     # arrays keep their capacity; grids shrink

    @property
    def active(self) -> bool:
        return self.n_long > 0 and self.n_chunks > 0


class SourceOrderedSplit:
    """Long-row tables with EXPLICIT chunks cut by source range (plnlp_row_split.seg_*), for a STATIC graph whose
    source matrix sits in the Infinity Cache but not in one XCD's L2 (the collab-shaped graph: 241 MB at F = 256).

    Position chunks (RowSplit) of a hub row each gather 128 source rows spread over the whole id range, and the
    chunks that run at the same time belong to a handful of hubs: nothing is shared, every gathered row is a fabric
    request (measured: L2 hit rate 21 % in that pass against 38 % in the main pass, profiles/r03_agg_pmc_step_launches.json).
    Here a hub row is cut where its (sorted) source ids cross multiples of `part_rows`, pieces longer than `max_len` are
    cut again, and the chunks are processed RANGE-major: the waves in flight all gather from the same few thousand source
    rows, which the hubs share (a source row is an entry of ~3 hub rows on that graph).  Together with slabs pinned to the
    XCDs (PLNLP_AGG_SLABS_XCD) a source row piece crosses the fabric about once.  The partial sums of a row are still
    added in slot (= position) order by the finalize pass: deterministic, but a different summation tree than RowSplit's
    -- a graph uses one form for ALL its launches at a width (the tuned form, ops._agg_tune: AGG_HUB_RANGES)."""

    def __init__(self, rowptr: torch.Tensor, col: torch.Tensor, threshold: int, part_rows: int = 8192,
                 max_len: int = 128):
        dev = rowptr.device
        self.threshold, self.part_rows, self.max_len = int(threshold), int(part_rows), int(max_len)
        deg = rowptr[1:] - rowptr[:-1]
        long_rows = torch.nonzero(deg > threshold).flatten()
        self.n_long = int(long_rows.numel())
        self.long_rows = long_rows.contiguous()
        self.n_chunks = 0
        if self.n_long == 0:
            return
        ldeg = deg[long_rows]
        lbeg = rowptr[long_rows]
        # every entry of a long row: its position, its row's slot, its source range
        slot_of = torch.repeat_interleave(torch.arange(self.n_long, device=dev), ldeg)
        first = torch.cumsum(ldeg, 0) - ldeg
        pos = lbeg[slot_of] + (torch.arange(slot_of.numel(), device=dev) - first[slot_of])
        part = col[pos].to(torch.int64) // self.part_rows
        n_parts = int(part.max()) + 1
        key = slot_of * n_parts + part
        # runs of equal (row, range): sorted columns give one run per pair; unsorted ones simply give more runs
        start = torch.ones_like(key, dtype=torch.bool)
        start[1:] = key[1:] != key[:-1]
        run_first = torch.nonzero(start).flatten()
        run_len = torch.diff(run_first, append=torch.tensor([key.numel()], device=dev))
        # cut runs longer than max_len
        pieces = (run_len + self.max_len - 1) // self.max_len
        run_of = torch.repeat_interleave(torch.arange(run_first.numel(), device=dev), pieces)
        pfirst = torch.cumsum(pieces, 0) - pieces
        j = torch.arange(run_of.numel(), device=dev) - pfirst[run_of]
        seg_off = run_first[run_of] + j * self.max_len                       # offset into `pos`
        seg_len = torch.minimum(run_len[run_of] - j * self.max_len, torch.tensor(self.max_len, device=dev))
        seg_beg = pos[seg_off]
        seg_row = slot_of[seg_off]
        seg_part = part[seg_off]
        n = int(seg_beg.numel())
        # slots: position order within the row (= the order built above); processing order: range-major
        order = torch.argsort(seg_part, stable=True)
        self.n_chunks = n
        self.seg_beg = seg_beg[order].contiguous()
        self.seg_len = seg_len[order].to(torch.int32).contiguous()
        self.seg_slot = order.to(torch.int32).contiguous()
        cnt = torch.bincount(seg_row, minlength=self.n_long)
        self.chunk_cnt = cnt.to(torch.int32).contiguous()
        self.chunk_beg = (torch.cumsum(cnt, 0) - cnt).contiguous()
        self.chunk_long = seg_row.to(torch.int32).contiguous()              # (not read by the explicit form)

    @property
    def active(self) -> bool:
        return self.n_long > 0 and self.n_chunks > 0


class Graph:
    def __init__(self, rowptr: torch.Tensor, col: torch.Tensor, val: Optional[torch.Tensor],
                 n_rows: int, n_cols: int):
        assert rowptr.dtype == torch.int64 and col.dtype == torch.int32
        assert rowptr.numel() == n_rows + 1
        self.rowptr, self.col, self.val = rowptr.contiguous(), col.contiguous(), val
        if val is not None:
            assert val.dtype == torch.float32 and val.numel() == col.numel()
            self.val = val.contiguous()
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self._t: Optional["Graph"] = None
        self._inv_deg: Optional[torch.Tensor] = None
        self._split: Optional[RowSplit] = None
        self._range_split: Optional[SourceOrderedSplit] = None
        self._agg_tune = {}      # measured choice of aggregation kernel form per feature width (ops._agg_tune);
                                 # shared with the transposed views: the backward passes cannot be timed themselves

    def row_split(self, threshold: int = 256, ranges=None):
        """long-row tables of this (static) graph, built once: position chunks (RowSplit), or -- ranges = (part_rows,
        max_len) -- chunks cut by source range and processed range-major (SourceOrderedSplit)"""
        if ranges is not None:
            key = (threshold,) + tuple(ranges)
            if getattr(self._range_split, "_key", None) != key:
                self._range_split = SourceOrderedSplit(self.rowptr, self.col, threshold, *ranges)
                self._range_split._key = key
            return self._range_split
        if self._split is None or self._split.threshold != threshold:
            self._split = RowSplit(self.rowptr, self.nnz, threshold, exact=True)
        return self._split

    # ---- construction ---------------------------------------------------------
    @classmethod
    def from_coo(cls, row: torch.Tensor, col: torch.Tensor, value: Optional[torch.Tensor] = None,
                 num_rows: Optional[int] = None, num_cols: Optional[int] = None) -> "Graph":
        """SparseTensor(row=, col=, value=): entries ordered by (row, col)."""
        row, col = row.to(torch.int64), col.to(torch.int64)
        n_rows = int(row.max()) + 1 if num_rows is None else int(num_rows)
        n_cols = n_rows if num_cols is None else int(num_cols)
        if col.numel() and int(col.max()) >= 2 ** 31:
            raise ValueError("column ids must fit int32")
        order = torch.argsort(row * n_cols + col, stable=True)
        row, col = row[order], col[order]
        val = None if value is None else value[order].to(torch.float32)
        rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=row.device)
        rowptr[1:] = torch.cumsum(torch.bincount(row, minlength=n_rows), 0)
        return cls(rowptr, col.to(torch.int32), val, n_rows, n_cols)

    @classmethod
    def from_edge_index(cls, edge_index: torch.Tensor, edge_weight: Optional[torch.Tensor] = None,
                        num_nodes: Optional[int] = None) -> "Graph":
        """T.ToSparseTensor() (main.py:81): adj_t = transposed adjacency, i.e. row =
        edge_index[1] (target), col = edge_index[0] (source)."""
        n = int(edge_index.max()) + 1 if num_nodes is None else int(num_nodes)
        return cls.from_coo(edge_index[1], edge_index[0], edge_weight, n, n)

    # ---- SparseTensor-like surface -----------------------------------------------
    @property
    def nnz(self) -> int:
        return self.col.numel()

    @property
    def device(self):
        return self.col.device

    def size(self, dim: int) -> int:
        return (self.n_rows, self.n_cols)[dim]

    def sparse_sizes(self) -> Tuple[int, int]:
        return self.n_rows, self.n_cols

    def row_index(self) -> torch.Tensor:
        deg = self.rowptr[1:] - self.rowptr[:-1]
        return torch.repeat_interleave(torch.arange(self.n_rows, device=self.device), deg)

    def coo(self):
        return self.row_index(), self.col.to(torch.int64), self.val

    def degree(self) -> torch.Tensor:
        return self.rowptr[1:] - self.rowptr[:-1]

    def inv_degree(self) -> torch.Tensor:
        """1 / max(rowlen, 1), fp32 -- the per-row factor of `mean` aggregation."""
        if self._inv_deg is None:
            self._inv_deg = 1.0 / self.degree().clamp(min=1).to(torch.float32)
        return self._inv_deg

    def sum(self, dim: int = 1) -> torch.Tensor:
        """adj_t.sum(dim=1) (main.py:146, utils.py:85,93)."""
        assert dim == 1
        if self.val is None:
            return self.degree().to(torch.float32)
        out = torch.zeros(self.n_rows, dtype=torch.float32, device=self.device)
        return out.index_add_(0, self.row_index(), self.val)

    def set_value(self, value: Optional[torch.Tensor]) -> "Graph":
        g = Graph(self.rowptr, self.col, value, self.n_rows, self.n_cols)
        return g

    def set_diag(self) -> "Graph":
        """diagonal := 1 (existing diagonal entries replaced) -- utils.py:84."""
        r, c, v = self.coo()
        v = torch.ones(c.numel(), device=self.device) if v is None else v
        off = r != c
        n = min(self.n_rows, self.n_cols)
        ar = torch.arange(n, device=self.device)
        return Graph.from_coo(torch.cat([r[off], ar]), torch.cat([c[off], ar]),
                              torch.cat([v[off], torch.ones(n, device=self.device)]),
                              self.n_rows, self.n_cols)

    def scale(self, row_scale: Optional[torch.Tensor], col_scale: Optional[torch.Tensor]) -> "Graph":
        """dense[N,1] * adj_t * dense[1,N] (utils.py:88,96)."""
        r, c, v = self.coo()
        v = torch.ones(c.numel(), device=self.device) if v is None else v
        if row_scale is not None:
            v = row_scale[r] * v
        if col_scale is not None:
            v = v * col_scale[c]
        return Graph(self.rowptr, self.col, v.to(torch.float32), self.n_rows, self.n_cols)

    def to_symmetric(self) -> "Graph":
        """[3P] SparseTensor.to_symmetric (main.py:110): union of A and A^T,
        duplicates summed when valued."""
        r, c, v = self.coo()
        rr, cc = torch.cat([r, c]), torch.cat([c, r])
        n = max(self.n_rows, self.n_cols)
        key = rr * n + cc
        uniq, inverse = torch.unique(key, return_inverse=True)
        val = None
        if v is not None:
            val = torch.zeros(uniq.numel(), device=self.device).index_add_(0, inverse, torch.cat([v, v]))
        return Graph.from_coo(uniq // n, uniq % n, val, n, n)

    def to(self, device) -> "Graph":
        g = Graph(self.rowptr.to(device), self.col.to(device),
                  None if self.val is None else self.val.to(device), self.n_rows, self.n_cols)
        return g

    def cuda(self):
        return self.to("cuda")

    def row_block(self, lo: int, n_rows: int, n_cols: Optional[int] = None) -> "Graph":
        """destination rows [lo, lo + n_rows) as their own (rectangular) graph: the CSR slice a rank of a
        row-sharded encoder owns.  Rows past this graph's end are empty rows (every rank's
        block has the same height, ceil(N / world)); n_cols may be padded the same way."""
        hi = min(self.n_rows, lo + n_rows)
        lo_c = min(lo, self.n_rows)
        e0, e1 = int(self.rowptr[lo_c]), int(self.rowptr[hi])
        rp = self.rowptr[lo_c:hi + 1] - e0
        if rp.numel() < n_rows + 1:
            rp = torch.cat([rp, rp[-1:].expand(n_rows + 1 - rp.numel())])
        val = None if self.val is None else self.val[e0:e1].contiguous()
        return Graph(rp.contiguous(), self.col[e0:e1].contiguous(), val, n_rows,
                     self.n_cols if n_cols is None else int(n_cols))

    def row_block_square(self, lo: int, n_rows: int, n_pad: int) -> "Graph":
        """the same destination rows [lo, lo + n_rows) as row_block, but kept at their GLOBAL row positions in an
        [n_pad, n_pad] graph whose other rows are empty.  A row-restricted conv (ops.SAGEConvFn with out_rows) on
        it sees global row ids everywhere -- the root operand, the dropout counters, the node map of the
        row-sparse backward -- so the sharded encoder's last layer runs the single-process kernels unchanged."""
        hi = min(self.n_rows, lo + n_rows)
        lo_c = min(lo, self.n_rows)
        e0, e1 = int(self.rowptr[lo_c]), int(self.rowptr[hi])
        rp = torch.zeros(n_pad + 1, dtype=torch.int64, device=self.device)
        rp[lo_c + 1:hi + 1] = self.rowptr[lo_c + 1:hi + 1] - e0
        rp[hi + 1:] = e1 - e0
        val = None if self.val is None else self.val[e0:e1].contiguous()
        return Graph(rp, self.col[e0:e1].contiguous(), val, n_pad, n_pad)

    # ---- transposed view (backward pass) --------------------------------------------
    def t(self) -> "Graph":
        if self._t is None:
            r, c, v = self.coo()
            self._t = Graph.from_coo(c, r, v, self.n_cols, self.n_rows)
            self._t._t = self
            self._t._agg_tune = self._agg_tune
        return self._t

    def t_pos(self) -> torch.Tensor:
        """int32 [nnz], aligned with t(): the row-relative position, in THIS graph's row, of the entry each
        transposed entry mirrors (what the backward of the max aggregation compares the saved arg
        with).  t() orders the entries by (col, row) with a stable sort; the same sort is repeated here."""
        if getattr(self, "_t_pos", None) is None:
            r, c, _ = self.coo()
            rel = torch.arange(self.nnz, device=self.device) - self.rowptr[r]
            order = torch.argsort(c * self.n_rows + r, stable=True)
            self._t_pos = rel[order].to(torch.int32).contiguous()
        return self._t_pos

    def dense_counts(self, min_density: float, min_rows: int, max_rows: int):
        """The adjacency as a dense matrix of entry COUNTS in bf16, [n_rows, n_cols rounded up to 16] (duplicates count twice,
        like the CSR sums) -- what plnlp_dense_aggregate_f32 multiplies on the matrix cores.  Only for graphs dense and small
        enough that this beats the CSR gather (ogbl-ddi: 4 267 nodes at 11.7 %: 36 MB); None otherwise, decided and built once
        per static graph.  The counts are the PATTERN's: a caller that wants the values too may use them only where the values depend on
        the column alone (t_mean(): A^T D^-1, whose column weights go to the kernel as src_scale: `_col_scale`)."""
        hit = getattr(self, "_dense_counts", False)
        if hit is not False:
            return hit
        out = None
        cells = self.n_rows * self.n_cols
        if (min_rows <= self.n_rows <= max_rows and self.n_cols <= max_rows and self.device.type == "cuda"
                and self.nnz >= min_density * cells):
            kp = (self.n_cols + 15) // 16 * 16
            dense = torch.zeros(self.n_rows, kp, dtype=torch.float32, device=self.device)
            dense.view(-1).index_add_(0, self.row_index() * kp + self.col.long(),
                                      torch.ones(self.nnz, dtype=torch.float32, device=self.device))
            if float(dense.max()) <= 256.0:                 # (bf16 holds integers up to 256 exactly)
                out = dense.to(torch.bfloat16).contiguous()
        self._dense_counts = out
        return out

    def t_mean(self) -> "Graph":
        """The operator of the mean aggregation's backward as ONE valued CSR: A^T D^-1, i.e. the
        transposed structure with entry value 1 / max(deg(source row of A), 1).  Built once per static
        graph; the backward then reads its weight next to the column index instead of gathering
        `inv_degree()[col]` behind it (one dependent round trip less per row)."""
        if getattr(self, "_t_mean", None) is None:
            gt = self.t()
            g = Graph(gt.rowptr, gt.col, self.inv_degree()[gt.col.long()].contiguous(), gt.n_rows, gt.n_cols)
            g._col_scale = self.inv_degree()          # (its values depend on the column only: val[e] = _col_scale[col[e]])
            g._agg_tune = self._agg_tune
            self._t_mean = g
        return self._t_mean

    def __repr__(self):
        return (f"Graph(n_rows={self.n_rows}, n_cols={self.n_cols}, nnz={self.nnz}, "
                f"valued={self.val is not None}, device={self.device})")


def gcn_normalization(adj_t: Graph) -> Graph:
    """plnlp/utils.py:83-89: D^-1/2 (A with diag := 1) D^-1/2, inf -> 0."""
    a = adj_t.set_diag()
    deg = a.sum(dim=1).to(torch.float)
    dis = deg.pow(-0.5)
    dis[dis == float("inf")] = 0
    return a.scale(dis, dis)


def adj_normalization(adj_t: Graph) -> Graph:
    """plnlp/utils.py:92-97: D^-1 A."""
    deg = adj_t.sum(dim=1).to(torch.float)
    inv = deg.pow(-1)
    inv[inv == float("inf")] = 0
    return adj_t.scale(inv, None)
