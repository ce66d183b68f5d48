"""CPU restatement of the PLNLP training hot path (TEST INFRASTRUCTURE -- see
oracle/__init__.py).  Pure torch on CPU, float32 by default, float64 on request.

Citations are relative to /root/reference (SURVEY.md convention).  Third-party
semantics (PyG 2.0.1 / torch_sparse / ogb 1.3.2) follow SURVEY.md Appendix A and
are marked [3P]; those parts are "parity unpinned" (no reference-side vectors
exist) and are cross-checked in tests/test_oracle.py instead.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

__all__ = [
    "CSR", "spmm", "spmm_max", "spmm_dense_check", "gcn_norm_csr",
    "SAGEConvRef", "GCNConvRef", "GNNRef", "MLPPredictorRef", "DotPredictorRef",
    "LOSSES", "pairwise_loss", "select_loss",
    "dropout_keep_mask", "counter_dropout", "random_walk_ref", "rmat_edges_ref", "rmat_thresholds",
    "batch_permutation", "local_neg_sample_ref", "pad_negatives_ref",
    "perm_copy_ref", "structured_negative_sampling_ref", "global_neg_sample_ref",
    "pos_neg_edges_ref", "hits_at_k", "mrr_list", "evaluate_hits_ref",
    "evaluate_mrr_ref", "clip_grad_norm_ref", "TrainerRef", "adjust_lr_ref", "collab_graph_prep_ref",
    "random_walk_pairs_ref", "LoggerRef", "run_loop_ref",
]


# --------------------------------------------------------------------------
# adjacency + SpMM  ([3P] torch_sparse.SparseTensor / matmul, Appendix A.1-3)
# --------------------------------------------------------------------------
class CSR:
    """Row-compressed adjacency, the role `data.adj_t` plays in main.py:81-83.

    Row i lists the *sources* j of messages into i (adj_t is the transposed
    adjacency), values optional.  int64 indices like torch_sparse.
    """

    def __init__(self, rowptr: torch.Tensor, col: torch.Tensor,
                 val: Optional[torch.Tensor], n_cols: int):
        self.rowptr, self.col, self.val, self.n_cols = rowptr, col, val, int(n_cols)
        self.n_rows = rowptr.numel() - 1
        self._t: Optional["CSR"] = None

    # -- construction -------------------------------------------------------
    @staticmethod
    def from_coo(row, col, val, n_rows, n_cols=None) -> "CSR":
        """SparseTensor(row=, col=, value=) as used at main.py:124,136,186:
        entries sorted by (row, col), duplicates kept."""
        n_cols = n_rows if n_cols is None else n_cols
        row = row.to(torch.int64)
        col = col.to(torch.int64)
        key = row * n_cols + col
        order = torch.argsort(key, stable=True)
        row, col = row[order], col[order]
        val = None if val is None else val[order]
        counts = torch.bincount(row, minlength=n_rows)
        rowptr = torch.zeros(n_rows + 1, dtype=torch.int64)
        rowptr[1:] = torch.cumsum(counts, 0)
        return CSR(rowptr, col, val, n_cols)

    # -- views --------------------------------------------------------------
    def row_index(self) -> torch.Tensor:
        return torch.repeat_interleave(torch.arange(self.n_rows), self.rowptr[1:] - self.rowptr[:-1])

    def coo(self):
        """SparseTensor.coo() (main.py:82,185,229)."""
        return self.row_index(), self.col, self.val

    def degree(self) -> torch.Tensor:
        return (self.rowptr[1:] - self.rowptr[:-1])

    def rowsum(self) -> torch.Tensor:
        """SparseTensor.sum(dim=1) (utils.py:85): stored values, or counts."""
        if self.val is None:
            return self.degree().to(torch.float32)
        out = torch.zeros(self.n_rows, dtype=self.val.dtype)
        return out.index_add_(0, self.row_index(), self.val)

    def t(self) -> "CSR":
        if self._t is None:
            r, c, v = self.coo()
            self._t = CSR.from_coo(c, r, v, self.n_cols, self.n_rows)
        return self._t

    def to_dense(self, dtype=torch.float64) -> torch.Tensor:
        d = torch.zeros(self.n_rows, self.n_cols, dtype=dtype)
        r, c, v = self.coo()
        v = torch.ones(c.numel(), dtype=dtype) if v is None else v.to(dtype)
        return d.index_put_((r, c), v, accumulate=True)

    def to_torch_csr(self, val: Optional[torch.Tensor] = None, dtype=torch.float32):
        v = self.val if val is None else val
        if v is None:
            v = torch.ones(self.col.numel(), dtype=dtype)
        return torch.sparse_csr_tensor(self.rowptr, self.col, v.to(dtype),
                                       size=(self.n_rows, self.n_cols))


class _SpMM(torch.autograd.Function):
    """[3P] torch_sparse.matmul autograd (Appendix A.3): forward CSR SpMM,
    gradient w.r.t. the dense operand = SpMM over the transposed matrix; for
    'mean' the per-edge value is 1/max(rowcount,1) of the *forward* row."""

    @staticmethod
    def forward(ctx, x, A, At):
        ctx.At = At
        return torch.sparse.mm(A, x) if x.dim() == 2 else None

    @staticmethod
    def backward(ctx, g):
        return torch.sparse.mm(ctx.At, g.contiguous()), None, None


def spmm(adj: CSR, x: torch.Tensor, reduce: str = "sum", use_values: bool = True,
         impl: str = "index_add") -> torch.Tensor:
    """out[i] = reduce_{e in row i} (val[e] *) x[col[e]]  -- [3P] torch_sparse
    spmm_{sum,mean}.  `mean` divides by max(rowcount,1) and ignores nothing:
    duplicates count twice, empty rows give 0.

    impl='index_add' is the transparent formulation used for parity;
    impl='sparse_csr' (MKL-backed torch.sparse.mm, fwd+bwd through the cached
    transpose exactly like torch_sparse) is what the CPU baseline times.
    """
    val = adj.val if use_values else None
    if impl == "sparse_csr":
        cache = getattr(adj, "_torch_csr", {})
        key = (reduce, use_values, x.dtype)
        if key not in cache:
            v = torch.ones(adj.col.numel(), dtype=x.dtype) if val is None else val.to(x.dtype)
            if reduce == "mean":
                inv = 1.0 / adj.degree().clamp(min=1).to(x.dtype)
                v = v * inv[adj.row_index()]
            A = adj.to_torch_csr(v, x.dtype)
            r, c, _ = adj.coo()
            At = CSR.from_coo(c, r, v, adj.n_cols, adj.n_rows).to_torch_csr(dtype=x.dtype)
            cache[key] = (A, At)
            adj._torch_csr = cache
        A, At = cache[key]
        return _SpMM.apply(x, A, At)
    if reduce == "max":
        return spmm_max(adj, x, use_values)[0]
    row = adj.row_index()
    msg = x[adj.col]
    if val is not None:
        msg = msg * val.to(x.dtype).unsqueeze(-1)
    out = torch.zeros(adj.n_rows, x.shape[1], dtype=x.dtype).index_add(0, row, msg)
    if reduce == "mean":
        out = out / adj.degree().clamp(min=1).to(x.dtype).unsqueeze(-1)
    elif reduce != "sum":
        raise ValueError(reduce)
    return out


def spmm_max(adj: CSR, x: torch.Tensor, use_values: bool = True):
    """[3P] torch_sparse spmm_max (matmul(adj_t, x, reduce='max')): out[i,f] = max over the stored
    entries of row i of (val *) x[col, f]; the FIRST maximal entry in CSR order wins (strict `>` in
    the row loop), its position is the arg; rows without entries give 0 and arg = -1; autograd sends
    the gradient to the arg-max entry only.  Returns (out, arg) with arg the row-relative position."""
    val = adj.val if use_values else None
    row = adj.row_index()
    msg = x[adj.col]
    if val is not None:
        msg = msg * val.to(x.dtype).unsqueeze(-1)
    nnz, feat = msg.shape
    idx = row.unsqueeze(-1).expand(-1, feat)
    top = torch.full((adj.n_rows, feat), float("-inf"), dtype=x.dtype).scatter_reduce(
        0, idx, msg.detach(), "amax", include_self=True)
    eidx = torch.arange(nnz).unsqueeze(-1).expand(-1, feat)
    cand = torch.where(msg.detach() == top[row], eidx, torch.full_like(eidx, nnz))
    arg = torch.full((adj.n_rows, feat), nnz, dtype=torch.int64).scatter_reduce(0, idx, cand, "amin",
                                                                                include_self=True)
    has = arg < nnz
    picked = msg.gather(0, arg.clamp(max=max(nnz - 1, 0))) if nnz else torch.zeros_like(top)
    out = torch.where(has, picked, torch.zeros_like(picked))
    rel = torch.where(has, arg - adj.rowptr[:-1].unsqueeze(-1), torch.full_like(arg, -1))
    return out, rel


def spmm_dense_check(adj: CSR, x: torch.Tensor, reduce="sum", use_values=True) -> torch.Tensor:
    """Independent float64 dense-algebra formulation (cross-check only)."""
    A = adj.to_dense() if use_values else CSR(adj.rowptr, adj.col, None, adj.n_cols).to_dense()
    out = A @ x.to(torch.float64)
    if reduce == "mean":
        out = out / adj.degree().clamp(min=1).to(torch.float64).unsqueeze(-1)
    return out


def gcn_norm_csr(adj: CSR) -> CSR:
    """plnlp/utils.py:83-89 gcn_normalization: set_diag() (diagonal := 1,
    existing diagonal entries replaced, [3P] Appendix A.2), deg = rowsum,
    deg^-1/2 with inf -> 0, row- and column-scale."""
    r, c, v = adj.coo()
    v = torch.ones(c.numel()) if v is None else v.to(torch.float32)
    off = r != c
    n = adj.n_rows
    ar = torch.arange(n)
    r2 = torch.cat([r[off], ar])
    c2 = torch.cat([c[off], ar])
    v2 = torch.cat([v[off], torch.ones(n)])
    a = CSR.from_coo(r2, c2, v2, n, adj.n_cols)
    deg = a.rowsum().to(torch.float32)
    dis = deg.pow(-0.5)
    dis[dis == float("inf")] = 0
    rr = a.row_index()
    a.val = dis[rr] * a.val * dis[a.col]
    return a


# --------------------------------------------------------------------------
# dropout: counter-based keep mask shared bit-for-bit with the HIP epilogues.
# The reference uses F.dropout (layer.py:22,26,85) whose Philox/MT stream
# cannot be reproduced across devices (SURVEY 7 "hard parts"); the law
# (Bernoulli(1-p) keep, 1/(1-p) scale) is the same.
# --------------------------------------------------------------------------
_M32 = np.uint64(0xFFFFFFFF)


def _lowbias32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x


def dropout_keep_mask(seed: int, n_rows: int, n_cols: int, p: float,
                      row0: int = 0) -> np.ndarray:
    """keep[r,c] for logical element index (row0+r)*n_cols + c under `seed`
    (a 64-bit per-call value chosen by the host).  Must match
    plnlp_amd/csrc/common.hip.h::dropout_keep exactly."""
    idx = (np.arange(row0, row0 + n_rows, dtype=np.uint64)[:, None] * np.uint64(n_cols)
           + np.arange(n_cols, dtype=np.uint64)[None, :])
    # elements are hashed in groups of four: two rounds over (group, seed) give word a, a third
    # round word b; element idx takes 16-bit field (idx & 3) of (a, b); P(drop) = thresh / 65536
    group = idx >> np.uint64(2)
    lo, hi = group & _M32, group >> np.uint64(32)
    s_lo, s_hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    a = _lowbias32(lo ^ s_lo)
    a = _lowbias32((a + ((hi * np.uint64(0x9E3779B9)) & _M32) + s_hi) & _M32)
    b = _lowbias32(a ^ np.uint64(0x85EBCA6B))
    word = np.where((idx & np.uint64(2)) != 0, b, a)
    field = (word >> ((idx & np.uint64(1)) * np.uint64(16))) & np.uint64(0xFFFF)
    thresh = np.uint64(max(0, min(int(p * 65536.0 + 0.5), 65535)))
    return field >= thresh


def random_walk_ref(adj: "CSR", start: torch.Tensor, walk_length: int, seed: int) -> torch.Tensor:
    """[3P] torch_cluster.random_walk semantics (Appendix A.6; main.py:242): column 0 = start, then a
    uniformly chosen neighbour per step, staying put on isolated nodes.  The uniform variate is the
    counter hash shared with plnlp_amd/csrc/incidence.hip::random_walk_kernel (floor(u * deg) as a
    32x32 -> high-32 multiply), so walks are bit-exact against the HIP kernel."""
    rowptr, col = adj.rowptr.numpy(), adj.col.numpy()
    cur = start.numpy().astype(np.int64).copy()
    s = cur.size
    out = np.empty((s, walk_length + 1), dtype=np.int64)
    out[:, 0] = cur
    s_lo, s_hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    w = np.arange(s, dtype=np.uint64)
    for l in range(walk_length):
        idx = w * np.uint64(walk_length) + np.uint64(l)
        h = _lowbias32((idx & _M32) ^ s_lo)
        h = _lowbias32((h + (((idx >> np.uint64(32)) * np.uint64(0x9E3779B9)) & _M32) + s_hi) & _M32)
        beg = rowptr[cur]
        deg = (rowptr[cur + 1] - beg).astype(np.uint64)
        off = ((h * deg) >> np.uint64(32)).astype(np.int64)
        nxt = col[np.minimum(beg + off, max(col.size - 1, 0))] if col.size else cur
        cur = np.where(deg > 0, nxt, cur)
        out[:, l + 1] = cur
    return torch.from_numpy(out)


def rmat_thresholds(probs=(0.57, 0.19, 0.19, 0.05)):
    """32-bit integer thresholds of the R-MAT quadrant draw: a, a + b, a + b + c as fractions of 2^32"""
    a, b, c, _ = probs
    return tuple(min(int(round(v * 4294967296.0)), 0xFFFFFFFF) for v in (a, a + b, a + b + c))


def _rmat_relabel(x: np.ndarray, scale: int, seed: int) -> np.ndarray:
    """the seeded bijection of [0, 2^scale) of plnlp_amd/csrc/incidence.hip::rmat_relabel"""
    m64 = (1 << 64) - 1
    M = np.uint64((1 << scale) - 1)
    sh = np.uint64((scale + 1) // 2)
    c1 = np.uint64(((seed * 0x9E3779B97F4A7C15) & m64) >> 7)
    c2 = np.uint64(((seed ^ 0xD6E8FEB86659FD93) * 0xBF58476D1CE4E5B9) & m64)
    with np.errstate(over="ignore"):
        x = (x * np.uint64(0x9E3779B97F4A7C15) + c1) & M
        x ^= x >> sh
        x = (x * np.uint64(0xBF58476D1CE4E5B9)) & M
        x ^= x >> sh
        x = (x * np.uint64(0x94D049BB133111EB) + c2) & M
        x ^= x >> sh
    return x


def rmat_edges_ref(scale: int, n_nodes: int, edge_lo: int, n_edges: int, seed: int,
                   probs=(0.57, 0.19, 0.19, 0.05), relabel: bool = True):
    """(rows, cols) int64 numpy arrays: edges [edge_lo, edge_lo + n_edges) of the R-MAT stream of BASELINE.json
    config 5 (the reference has no generator of its own -- it loads OGB files, main.py:74-95).
    Restates plnlp_amd/csrc/incidence.hip::rmat_edges_kernel bit for bit: `scale` quadrant draws per edge from
    the counter hash of (seed, edge * 64 + level) against integer thresholds, a seeded relabelling bijection,
    ids folded mod n_nodes."""
    t_a, t_ab, t_abc = (np.uint64(v) for v in rmat_thresholds(probs))
    s_lo, s_hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    e = np.arange(edge_lo, edge_lo + n_edges, dtype=np.uint64)
    r = np.zeros(n_edges, dtype=np.uint64)
    c = np.zeros(n_edges, dtype=np.uint64)
    for l in range(scale):
        idx = e * np.uint64(64) + np.uint64(l)
        h = _lowbias32((idx & _M32) ^ s_lo)
        h = _lowbias32((h + (((idx >> np.uint64(32)) * np.uint64(0x9E3779B9)) & _M32) + s_hi) & _M32)
        right = ((h >= t_a) & (h < t_ab)) | (h >= t_abc)
        down = h >= t_ab
        r = (r << np.uint64(1)) | down.astype(np.uint64)
        c = (c << np.uint64(1)) | right.astype(np.uint64)
    if relabel:
        r, c = _rmat_relabel(r, scale, seed), _rmat_relabel(c, scale, seed)
    n = np.uint64(n_nodes)
    return (r % n).astype(np.int64), (c % n).astype(np.int64)


def counter_dropout(x: torch.Tensor, p: float, seed: int, training: bool = True) -> torch.Tensor:
    if not training or p <= 0.0:
        return x
    keep = torch.from_numpy(dropout_keep_mask(seed, x.shape[0], x.shape[1], p))
    scale = torch.tensor(1.0 / (1.0 - p), dtype=torch.float32).to(x.dtype)
    return x * keep.to(x.dtype) * scale


# --------------------------------------------------------------------------
# encoders (plnlp/layer.py:7-45) and predictors (layer.py:66-87,167-176)
# --------------------------------------------------------------------------
class SAGEConvRef(torch.nn.Module):
    """[3P] PyG 2.0.1 SAGEConv(in,out) defaults (Appendix A.1):
    out = lin_l(mean_{j in N(i)} x_j) + lin_r(x_i); edge values dropped."""

    def __init__(self, cin, cout):
        super().__init__()
        self.lin_l = torch.nn.Linear(cin, cout, bias=True)
        self.lin_r = torch.nn.Linear(cin, cout, bias=False)

    def reset_parameters(self):
        self.lin_l.reset_parameters()
        self.lin_r.reset_parameters()

    def forward(self, x, adj: CSR, impl="index_add"):
        agg = spmm(adj, x, "mean", use_values=False, impl=impl)
        return self.lin_l(agg) + self.lin_r(x)


class GCNConvRef(torch.nn.Module):
    """[3P] PyG 2.0.1 GCNConv(in,out,normalize=False) (Appendix A.2):
    out = A_hat (x W^T) + b, glorot weight, zero bias, stored values used."""

    def __init__(self, cin, cout):
        super().__init__()
        self.bias = torch.nn.Parameter(torch.zeros(cout))
        self.lin = torch.nn.Linear(cin, cout, bias=False)
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.xavier_uniform_(self.lin.weight)
        torch.nn.init.zeros_(self.bias)

    def forward(self, x, adj: CSR, impl="index_add"):
        return spmm(adj, self.lin(x), "sum", use_values=True, impl=impl) + self.bias


class GNNRef(torch.nn.Module):
    """BaseGNN.forward control flow, layer.py:18-27: conv->relu->dropout for
    all but the last conv; last conv bare, EXCEPT relu+dropout are applied
    when num_layers == 1 (layer.py:24-26)."""

    def __init__(self, kind, cin, hidden, cout, num_layers, dropout, spmm_impl="index_add"):
        super().__init__()
        conv = {"SAGE": SAGEConvRef, "GCN": GCNConvRef}[kind.upper()]
        self.convs = torch.nn.ModuleList()
        for i in range(num_layers):
            a = cin if i == 0 else hidden
            b = cout if i == num_layers - 1 else hidden
            self.convs.append(conv(a, b))
        self.dropout, self.num_layers, self.spmm_impl = dropout, num_layers, spmm_impl
        # dropout hook: (x, layer_index) -> x ; default torch dropout
        self.dropout_fn: Optional[Callable] = None

    def reset_parameters(self):
        for c in self.convs:
            c.reset_parameters()

    def _drop(self, x, i):
        if self.dropout_fn is not None:
            return self.dropout_fn(x, i)
        return F.dropout(x, p=self.dropout, training=self.training)

    def forward(self, x, adj):
        n = len(self.convs)
        for i, conv in enumerate(self.convs):
            x = conv(x, adj, impl=self.spmm_impl)
            if i < n - 1 or self.num_layers == 1:
                x = self._drop(F.relu(x), i)
        return x


class MLPPredictorRef(torch.nn.Module):
    """layer.py:66-87: Hadamard, then Linear->relu->dropout ... Linear."""

    def __init__(self, cin, hidden, cout, num_layers, dropout):
        super().__init__()
        self.lins = torch.nn.ModuleList()
        for i in range(num_layers):
            a = cin if i == 0 else hidden
            b = cout if i == num_layers - 1 else hidden
            self.lins.append(torch.nn.Linear(a, b))
        self.dropout = dropout
        self.dropout_fn: Optional[Callable] = None

    def reset_parameters(self):
        for lin in self.lins:
            lin.reset_parameters()

    def forward(self, x_i, x_j):
        x = x_i * x_j
        for i, lin in enumerate(self.lins):
            x = lin(x)
            if i < len(self.lins) - 1:
                x = F.relu(x)
                x = (self.dropout_fn(x, i) if self.dropout_fn is not None
                     else F.dropout(x, p=self.dropout, training=self.training))
        return x


class DotPredictorRef(torch.nn.Module):
    """layer.py:167-176."""

    def reset_parameters(self):
        return

    def forward(self, x_i, x_j):
        return (x_i * x_j).sum(-1)


# --------------------------------------------------------------------------
# losses (plnlp/loss.py:5-62) -- table driven
# --------------------------------------------------------------------------
def pairwise_loss(kind: str, pos, neg, num_neg: int, w=None):
    """d = pos[b] - neg[b,n] over the [B,1] x [B,k] grid (loss.py reshape rule:
    negatives b*k .. b*k+k-1 belong to positive b)."""
    p = pos.reshape(-1, 1)
    n = neg.reshape(-1, num_neg)
    d = p - n
    if w is not None:
        w = w.reshape(-1, 1)
    if kind == "auc":                       # loss.py:5-8
        return torch.square(1 - d).sum()
    if kind == "hinge_auc":                 # loss.py:11-14
        return torch.square(torch.clamp(1 - d, min=0)).sum()
    if kind == "weighted_auc":              # loss.py:17-21
        return (w * torch.square(1 - d)).sum()
    if kind == "adaptive_auc":              # loss.py:24-28
        return torch.square(w - d).sum()
    if kind == "weighted_hinge_auc":        # loss.py:31-35 (weight doubles as margin)
        return (w * torch.square(torch.clamp(w - d, min=0))).sum()
    if kind == "adaptive_hinge_auc":        # loss.py:38-42
        return torch.square(torch.clamp(w - d, min=0)).sum()
    if kind == "log_rank":                  # loss.py:45-48 (mean)
        return -torch.log(torch.sigmoid(d) + 1e-15).mean()
    if kind == "info_nce":                  # loss.py:57-62 (mean)
        pe = torch.exp(p)
        ne = torch.exp(n).sum(1, keepdim=True)
        return -torch.log(pe / (pe + ne) + 1e-15).mean()
    raise KeyError(kind)


def _ce(pos, neg):                          # loss.py:51-54
    return (-torch.log(torch.sigmoid(pos) + 1e-15).mean()
            - torch.log(1 - torch.sigmoid(neg) + 1e-15).mean())


LOSSES: Dict[str, Callable] = {
    k: (lambda kind: (lambda pos, neg, num_neg, w=None: pairwise_loss(kind, pos, neg, num_neg, w)))(k)
    for k in ["auc", "hinge_auc", "weighted_auc", "adaptive_auc", "weighted_hinge_auc",
              "adaptive_hinge_auc", "log_rank", "info_nce"]
}
LOSSES["ce"] = lambda pos, neg, num_neg=None, w=None: _ce(pos, neg)

_DISPATCH = {  # model.py:107-126: CLI name -> (kind, needs_margin)
    "CE": ("ce", False), "InfoNCE": ("info_nce", False), "LogRank": ("log_rank", False),
    "HingeAUC": ("hinge_auc", False), "AdaAUC": ("adaptive_auc", True),
    "WeightedAUC": ("weighted_auc", True), "AdaHingeAUC": ("adaptive_hinge_auc", True),
    "WeightedHingeAUC": ("weighted_hinge_auc", True),
}


def select_loss(name: str, margin_available: bool) -> str:
    """model.py:107-126: weighted/adaptive names silently fall back to plain
    auc_loss when the split carries no weight; unknown names -> auc_loss."""
    kind, needs = _DISPATCH.get(name, ("auc", False))
    if needs and not margin_available:
        return "auc"
    return kind


# --------------------------------------------------------------------------
# index streams: batches (model.py:147) and negatives (negative_sample.py)
# --------------------------------------------------------------------------
def batch_permutation(n: int, batch_size: int, shuffle: bool) -> List[torch.Tensor]:
    """DataLoader(range(n), batch_size, shuffle) (model.py:147,178), torch 2.10
    behaviour probed in SURVEY 8a-10: the iterator draws `base_seed` from the
    default CPU generator, the RandomSampler draws its own seed from it, then
    randperm(n) under a fresh Generator seeded with that value."""
    if shuffle:
        _base_seed = torch.empty((), dtype=torch.int64).random_().item()
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        g = torch.Generator()
        g.manual_seed(seed)
        order = torch.randperm(n, generator=g)
    else:
        # the non-shuffling iterator still draws base_seed
        _base_seed = torch.empty((), dtype=torch.int64).random_().item()
        order = torch.arange(n)
    return list(torch.split(order, batch_size))


def local_neg_sample_ref(pos_edges, num_nodes, num_neg):
    """negative_sample.py:31-43 (random_src=False): keep the source, one
    torch.randint draw of k*E destinations on the default CPU generator."""
    e = pos_edges.size(0)
    src = pos_edges[:, 0].reshape(-1, 1).repeat(1, num_neg).reshape(-1)
    dst = torch.randint(0, num_nodes, (num_neg * e,), dtype=torch.long)
    return torch.stack((src, dst), dim=-1).reshape(-1, num_neg, 2)


def pad_negatives_ref(neg_edge, want):
    """negative_sample.py:11-18: if the structured sampler came back short,
    pad with rows picked by torch.randperm(M)[:short]."""
    src, dst = neg_edge[0], neg_edge[1]
    m = neg_edge.size(1)
    if m < want:
        pick = torch.randperm(m)[: want - m]
        src = torch.cat((src, src[pick]))
        dst = torch.cat((dst, dst[pick]))
    return src, dst


def perm_copy_ref(edge_index, target, copies):
    """negative_sample.py:61-76 sample_perm_copy."""
    src, dst = pad_negatives_ref(edge_index, target)
    s0, d0 = src, dst
    for _ in range(copies - 1):
        pick = torch.randperm(target)
        src = torch.cat((src, s0[pick]))
        dst = torch.cat((dst, d0[pick]))
    return torch.stack((src, dst), dim=-1).reshape(-1, copies, 2)


def structured_negative_sampling_ref(edge_index, num_nodes, num_neg_samples,
                                     rng: Optional[np.random.Generator] = None):
    """[3P, UNPINNED] torch_geometric.utils.negative_sampling(method='sparse')
    semantics (Appendix A.4): linearise existing edges, oversample candidate
    ids, drop hits on existing ids, up to 3 rounds, truncate.  The reference
    stream (Python `random.sample`) cannot be reproduced; the contract is
    distributional: no returned pair is an existing entry of `edge_index`
    (which already contains self loops, negative_sample.py:8), count <= asked.
    """
    rng = np.random.default_rng() if rng is None else rng
    n = int(num_nodes)
    size = n * n
    existing = (edge_index[0].to(torch.int64) * n + edge_index[1].to(torch.int64)).numpy()
    existing = np.unique(existing)
    want = min(int(num_neg_samples), size - existing.size)
    density = existing.size / float(size)
    alpha = abs(1.0 / (1.0 - 1.1 * density)) if density < 0.9 else 10.0
    got = np.empty(0, dtype=np.int64)
    for _ in range(3):
        k = min(int(alpha * want) + 1, size)
        cand = rng.choice(size, size=k, replace=False) if k * 4 > size else np.unique(
            rng.integers(0, size, size=int(k * 1.05) + 8))[:k]
        rng.shuffle(cand)
        cand = cand[~np.isin(cand, existing)]
        got = cand if got.size == 0 else np.concatenate([got, cand[~np.isin(cand, got)]])
        if got.size >= want:
            got = got[:want]
            break
    got = torch.from_numpy(got.astype(np.int64))
    return torch.stack([got // n, got % n], dim=0)


def global_neg_sample_ref(edge_index, num_nodes, num_samples, num_neg, rng=None):
    """negative_sample.py:6-20: add_self_loops ([3P] A.5: N inferred as max+1),
    structured negative sampling of E*k pairs, pad, reshape [E,k,2]."""
    n_inferred = int(edge_index.max()) + 1
    loops = torch.arange(n_inferred).repeat(2, 1)
    ei = torch.cat([edge_index, loops], dim=1)
    neg = structured_negative_sampling_ref(ei, num_nodes, num_samples * num_neg, rng)
    src, dst = pad_negatives_ref(neg, num_samples * num_neg)
    return torch.stack((src, dst), dim=-1).reshape(-1, num_neg, 2)


def pos_neg_edges_ref(split, split_edge, edge_index=None, num_nodes=None,
                      neg_sampler_name=None, num_neg=None, rng=None):
    """plnlp/utils.py:7-41 get_pos_neg_edges."""
    tr = split_edge["train"]
    if "edge" in tr:
        pos = split_edge[split]["edge"]
    else:                                                  # citation2, utils.py:10-13
        source = split_edge[split]["source_node"]
        pos = torch.stack([source, split_edge[split]["target_node"]]).t()
    if split == "train":
        if neg_sampler_name == "local":
            neg = local_neg_sample_ref(pos, num_nodes, num_neg)
        elif neg_sampler_name == "global":
            neg = global_neg_sample_ref(edge_index, num_nodes, pos.size(0), num_neg, rng)
        else:
            raise NotImplementedError("global_perm sampler is out of scope (SURVEY 2.1 #6)")
    elif "edge" in tr:
        neg = split_edge[split]["edge_neg"]
    else:                                                  # utils.py:36-40
        tneg = split_edge[split]["target_node_neg"]
        neg = torch.stack([source.repeat_interleave(tneg.size(1)), tneg.reshape(-1)]).t()
    return pos, neg


# --------------------------------------------------------------------------
# evaluation ([3P, UNPINNED] ogb 1.3.2 Evaluator, Appendix A.7; utils.py:44-80)
# --------------------------------------------------------------------------
def hits_at_k(pos_pred, neg_pred, k: int) -> float:
    if neg_pred.numel() < k:
        return 1.0
    kth = torch.topk(neg_pred, k).values[-1]
    return float((pos_pred > kth).sum().item()) / pos_pred.numel()


def mrr_list(pos_pred, neg_pred) -> torch.Tensor:
    """rank of the positive inside [pos | negs] by descending score.  ogb 1.3.2 takes
    `torch.argsort(y, dim=1, descending=True)` and leaves the order of EQUAL scores to the sort
    implementation (it differs between torch's CPU and device sorts, and between torch versions); this
    restatement pins it with a stable sort: the positive (column 0) stays ahead of negatives it ties
    with.  Without ties the two are the same function."""
    y = torch.cat([pos_pred.reshape(-1, 1), neg_pred], dim=1)
    order = torch.argsort(y, dim=1, descending=True, stable=True)
    rank = (order == 0).nonzero(as_tuple=False)[:, 1] + 1
    return 1.0 / rank.to(torch.float)


def evaluate_hits_ref(pos_val, neg_val, pos_test, neg_test):
    return {f"Hits@{k}": (hits_at_k(pos_val, neg_val, k), hits_at_k(pos_test, neg_test, k))
            for k in (20, 50, 100)}


def evaluate_mrr_ref(pos_val, neg_val, pos_test, neg_test):
    nv = neg_val.view(pos_val.shape[0], -1)
    nt = neg_test.view(pos_test.shape[0], -1)
    return {"MRR": (mrr_list(pos_val, nv).mean().item(), mrr_list(pos_test, nt).mean().item())}


# --------------------------------------------------------------------------
# trainer (plnlp/model.py:45-226)
# --------------------------------------------------------------------------
def clip_grad_norm_ref(params: Sequence[torch.Tensor], max_norm: float) -> torch.Tensor:
    """[3P] torch.nn.utils.clip_grad_norm_ (Appendix A.8): coef = c/(||g||+1e-6)
    clamped to <= 1, always multiplied."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:                      # DOT predictor: torch warns and returns 0
        return torch.tensor(0.0)
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def adjust_lr_ref(optimizer, decay_ratio, lr):
    """model.py:279-286: linear decay with floor 1e-4*lr."""
    new = max(lr * (1 - decay_ratio), lr * 0.0001)
    for group in optimizer.param_groups:
        group["lr"] = new
    return new


class TrainerRef:
    """BaseModel restated (model.py:45-226) around injected modules.

    `emb` is an nn.Embedding (or None), `x` optional raw node features
    (create_input_feat, model.py:98-105)."""

    def __init__(self, encoder, predictor, emb, adj: CSR, *, x=None, loss_name="AUC",
                 lr=1e-3, clip_norm=2.0, optimizer="Adam"):
        self.encoder, self.predictor, self.emb, self.adj, self.x = encoder, predictor, emb, adj, x
        self.loss_name, self.clip_norm = loss_name, clip_norm
        self.params = list(encoder.parameters()) + list(predictor.parameters())
        if emb is not None:
            self.params += list(emb.parameters())
        if optimizer == "AdamW":
            self.optimizer = torch.optim.AdamW(self.params, lr=lr)
        elif optimizer == "SGD":
            self.optimizer = torch.optim.SGD(self.params, lr=lr, momentum=0.9,
                                             weight_decay=1e-5, nesterov=True)
        else:
            self.optimizer = torch.optim.Adam(self.params, lr=lr)

    def param_init(self):                                   # model.py:92-96
        self.encoder.reset_parameters()
        self.predictor.reset_parameters()
        if self.emb is not None:
            torch.nn.init.xavier_uniform_(self.emb.weight)

    def input_feat(self):                                   # model.py:98-105
        if self.x is not None:
            return torch.cat([self.emb.weight, self.x], dim=-1) if self.emb is not None else self.x
        return self.emb.weight

    def step(self, pos_edge, neg_edge, num_neg, weight=None):
        """one iteration of model.py:147-171 on given [B,2] / [B,k,2] indices."""
        self.optimizer.zero_grad()
        h = self.encoder(self.input_feat(), self.adj)
        pe = pos_edge.t()
        ne = neg_edge.reshape(-1, 2).t()
        pos_out = self.predictor(h[pe[0]], h[pe[1]])
        neg_out = self.predictor(h[ne[0]], h[ne[1]])
        kind = select_loss(self.loss_name, weight is not None)
        loss = LOSSES[kind](pos_out, neg_out, num_neg, weight)
        loss.backward()
        if self.clip_norm >= 0:                             # model.py:163-165
            clip_grad_norm_ref(list(self.encoder.parameters()), self.clip_norm)
            clip_grad_norm_ref(list(self.predictor.parameters()), self.clip_norm)
        self.optimizer.step()
        return loss.detach(), pos_out.detach(), neg_out.detach()

    def train_epoch(self, pos_train_edge, neg_train_edge, batch_size, num_neg, weight=None):
        """model.py:128-173 after the negatives are drawn; returns
        sum(loss_b * B_b) / sum(B_b) (model.py:169-173)."""
        self.encoder.train()
        self.predictor.train()
        tot = 0.0
        cnt = 0
        for perm in batch_permutation(pos_train_edge.size(0), batch_size, True):
            w = weight[perm] if weight is not None else None
            loss, pos_out, _ = self.step(pos_train_edge[perm], neg_train_edge[perm], num_neg, w)
            tot += loss.item() * pos_out.size(0)
            cnt += pos_out.size(0)
        return tot / cnt

    @torch.no_grad()
    def score(self, h, edges, batch_size):                  # model.py:175-182
        out = []
        for perm in batch_permutation(edges.size(0), batch_size, False):
            e = edges[perm].t()
            out.append(self.predictor(h[e[0]], h[e[1]]).reshape(-1))
        return torch.cat(out)

    @torch.no_grad()
    def embed_for_eval(self):                               # model.py:189-194
        self.encoder.eval()
        self.predictor.eval()
        h = self.encoder(self.input_feat(), self.adj)
        return torch.cat([h, h.mean(0, keepdim=True)], 0)


# --------------------------------------------------------------------------
# driver-side graph preparation (main.py:109-150), small cases only
# --------------------------------------------------------------------------
def collab_graph_prep_ref(train_edge, train_w, train_year, valid_edge, valid_w, num_nodes, year=-1,
                          use_valedges_as_input=False, use_coalesce=False):
    """main.py:112-150 restated with plain loops over a dense float64 adjacency (pure-Python: toy sizes).
    [3P] to_undirected(edge_index, w, reduce='add') = both directions, duplicates summed;
    SparseTensor(row, col, value) puts value at A[row, col]; coalesce sorts by (row, col) and sums.
    Returns dict(adj=dense A or None, edge_index_keys=set of (r, c) or None, train_edge [E,2],
    train_weight [E], train_year)."""
    te, tw, ty = train_edge.clone(), train_w.clone(), None if train_year is None else train_year.clone()
    adj = keys = None

    def undirected(pairs, weights):
        a = torch.zeros(num_nodes, num_nodes, dtype=torch.float64)
        ks = set()
        for (u, v), w in zip(pairs.tolist(), weights.tolist()):
            a[u, v] += w
            a[v, u] += w
            ks.update({(u, v), (v, u)})
        return a.to(torch.float32).double(), ks          # values pass through float32 (main.py:124,137)

    if year > 0 and ty is not None:                       # main.py:114-126
        keep = [i for i, y in enumerate(ty.tolist()) if y >= year]
        te, tw, ty = te[keep], tw[keep], ty[keep]
        adj, keys = undirected(te, tw)
    if use_valedges_as_input:                             # main.py:129-150
        pairs = torch.cat([valid_edge, te], dim=0)        # edges  [valid, train]   (main.py:131)
        weights = torch.cat([tw, valid_w], dim=0)         # weights [train, valid]  (main.py:132) -- the quirk
        adj, keys = undirected(pairs, weights)
        if use_coalesce:                                  # main.py:139-140
            merged = {}
            for (u, v), w in zip(pairs.tolist(), weights.tolist()):
                merged[(u, v)] = merged.get((u, v), 0.0) + w
            order = sorted(merged)
            pairs = torch.tensor(order, dtype=torch.int64).reshape(-1, 2)
            weights = torch.tensor([merged[k] for k in order], dtype=weights.dtype)
        deg = adj.sum(dim=1).to(torch.float32)
        dis = deg.pow(-0.5)
        dis[dis == float("inf")] = 0
        te = pairs
        tw = dis[pairs[:, 0]] * weights * dis[pairs[:, 1]]
    return {"adj": adj, "edge_index_keys": keys, "train_edge": te, "train_weight": tw, "train_year": ty}


# --------------------------------------------------------------------------
# the driver's run loop (main.py:228-305)
# --------------------------------------------------------------------------
def random_walk_pairs_ref(walk: torch.Tensor, walk_length: int):
    """main.py:243-253: training pairs (start, j-th hop), j = 1 .. L, concatenated HOP-MAJOR (all walks' first hop,
    then all second hops, ...), weight 1 / j, pairs whose two ends coincide removed -- the order matters: it is the
    order the epoch's batch permutation indexes."""
    pairs, weights = [], []
    for j in range(walk_length):
        pairs.append(walk[:, [0, j + 1]])
        weights.append(torch.ones((walk.size(0),), dtype=torch.float) / (j + 1))
    pairs = torch.cat(pairs, dim=0)
    weights = torch.cat(weights, dim=0)
    mask = (pairs[:, 0] - pairs[:, 1]) != 0
    return torch.masked_select(pairs, mask.view(-1, 1)).view(-1, 2), torch.masked_select(weights, mask)


class LoggerRef:
    """plnlp/logger.py:6-50 restated (pinned by fixture G9 through tests/test_oracle.py): per run the test score at
    the best-validation evaluation point, over runs mean and unbiased std of those picks; returns the text."""

    def __init__(self, runs):
        self.results = [[] for _ in range(runs)]

    def add_result(self, run, result):
        assert len(result) == 2 and 0 <= run < len(self.results)
        self.results[run].append(result)

    @staticmethod
    def _best(r, last_best):
        v = r[:, 0]
        at = v.size(0) - int(v.flip(dims=[0]).argmax()) - 1 if last_best else int(v.argmax())
        return at

    def statistics(self, run=None, last_best=False) -> List[str]:
        if run is not None:
            r = 100 * torch.tensor(self.results[run])
            at = self._best(r, last_best)
            return [f"Run {run + 1:02d}:", f"Highest Valid: {r[:, 0].max():.2f}", f"Highest Eval Point: {at + 1}",
                    f"   Final Test: {r[at, 1]:.2f}"]
        best = []
        for r in 100 * torch.tensor(self.results):
            at = self._best(r, last_best)
            best.append((r[:, 0].max().item(), r[at, 1].item()))
        b = torch.tensor(best)
        return ["All runs:", f"Highest Valid: {b[:, 0].mean():.2f}  {b[:, 0].std():.2f}",
                f"   Final Test: {b[:, 1].mean():.2f}  {b[:, 1].std():.2f}"]


def run_loop_ref(trainer: "TrainerRef", split_edge: dict, *, num_nodes: int, runs: int, epochs: int, batch_size: int,
                 neg_sampler: str, num_neg: int, lr: float, eval_steps: int = 1, log_steps: int = 1,
                 use_lr_decay: bool = False, eval_metric: str = "hits", eval_last_best: bool = False,
                 random_walk_augment: bool = False, walk_length: int = 5, rw_adj: Optional["CSR"] = None,
                 rw_start: Optional[torch.Tensor] = None, rw_seed: int = 0, edge_index=None,
                 on_run_start: Optional[Callable] = None, on_epoch: Optional[Callable] = None) -> dict:
    """main.py:235-305, the run / epoch / eval / logging loop, around a TrainerRef.

    Per epoch: (random_walk_augment) a fresh set of walks -> pairs + weights (main.py:241-253; the walks come from
    random_walk_ref under seed rw_seed + epoch count, the counter stream the HIP kernel shares -- torch_cluster's own
    stream is third-party and unpinned), `model.train` (negatives drawn, then the DataLoader permutation: both from the
    default CPU generator, in that order), every eval_steps epochs `model.test` + Logger.add_result + the epoch line
    (main.py:260-286), then adjust_lr (main.py:288-291: AFTER the epoch, so epoch 1 trains at the full rate).
    on_run_start(run, trainer): called where main.py calls model.param_init() (main.py:236) -- the caller installs the
    run's initial weights (and RNG state) there.  Returns dict(lines, losses, lrs, results, logger_text)."""
    keys = ["Hits@20", "Hits@50", "Hits@100"] if eval_metric == "hits" else ["MRR"]
    loggers = {k: LoggerRef(runs) for k in keys}
    out = {"lines": [], "losses": [], "lrs": [], "results": [], "logger_text": []}
    split_edge = {k: dict(v) for k, v in split_edge.items()}
    walks_drawn = 0
    for run in range(runs):
        if on_run_start is not None:
            on_run_start(run, trainer)
        else:
            trainer.param_init()
        # main.py:236-239: param_init() re-draws the weights only -- the optimiser (Adam moments, step count, and the
        # learning rate adjust_lr left behind) carries over from the previous run, while the PRINTED rate restarts
        cur_lr = lr
        for epoch in range(1, 1 + epochs):
            if random_walk_augment:
                walks_drawn += 1
                walk = random_walk_ref(rw_adj, rw_start, walk_length, rw_seed + walks_drawn)
                split_edge["train"]["edge"], split_edge["train"]["weight"] = random_walk_pairs_ref(walk, walk_length)
            pos, neg = pos_neg_edges_ref("train", split_edge, edge_index=edge_index, num_nodes=num_nodes,
                                         neg_sampler_name=neg_sampler, num_neg=num_neg)
            weight = split_edge["train"].get("weight")
            loss = trainer.train_epoch(pos, neg.view(-1, num_neg, 2), batch_size, num_neg, weight)
            out["losses"].append(loss)
            out["lrs"].append(cur_lr)
            if epoch % eval_steps == 0:
                hh = trainer.embed_for_eval()
                pv, nv = pos_neg_edges_ref("valid", split_edge)
                pt, nt = pos_neg_edges_ref("test", split_edge)
                spv, snv = trainer.score(hh, pv, batch_size), trainer.score(hh, nv, batch_size)
                spt, snt = trainer.score(hh, pt, batch_size), trainer.score(hh, nt, batch_size)
                results = (evaluate_hits_ref if eval_metric == "hits" else evaluate_mrr_ref)(spv, snv, spt, snt)
                out["results"].append(results)
                for key, result in results.items():
                    loggers[key].add_result(run, result)
                if epoch % log_steps == 0:
                    for key, (valid_res, test_res) in results.items():
                        out["lines"] += [key, (f"Run: {run + 1:02d}, Epoch: {epoch:02d}, Loss: {loss:.4f}, "
                                               f"Learning Rate: {cur_lr:.4f}, Valid: {100 * valid_res:.2f}%, "
                                               f"Test: {100 * test_res:.2f}%")]
            if on_epoch is not None:
                on_epoch(run, epoch, trainer)
            if use_lr_decay:
                cur_lr = adjust_lr_ref(trainer.optimizer, epoch / epochs, lr)
        for key in loggers:
            out["logger_text"] += [key] + loggers[key].statistics(run, last_best=eval_last_best)
    for key in loggers:
        out["logger_text"] += [key] + loggers[key].statistics(last_best=eval_last_best)
    return out
