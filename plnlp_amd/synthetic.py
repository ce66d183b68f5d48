"""Synthetic OGB-shaped inputs.  No dataset files exist on the GPU
box and there is no network, so benchmarks and parity runs use seeded graphs
with the node / edge counts and degree skew of the OGB datasets the reference's
recipes name (README.md:24,31,35,40).  Pure torch; runs on CPU or GPU."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch

from .graph import Graph

SHAPES = {
    # name: nodes, undirected train edges, skew exponent of the Chung-Lu weights
    "ddi": dict(num_nodes=4267, num_edges=1_067_911, skew=0.5),
    "collab": dict(num_nodes=235_868, num_edges=1_179_052, skew=0.9),
    "citation2": dict(num_nodes=2_927_963, num_edges=30_387_995, skew=0.8),
}


@dataclass
class SyntheticData:
    """duck-type of the `data` object main.py hands to BaseModel.train/test"""
    adj_t: Graph
    edge_index: torch.Tensor          # [2, nnz] int64, CPU (negative sampler input)
    num_nodes: int
    x: Optional[torch.Tensor] = None


def _chung_lu_edges(n: int, m: int, skew: float, gen: torch.Generator, device) -> torch.Tensor:
    """~m distinct undirected non-loop edges with P(node i) ~ (i+1)^-skew"""
    w = torch.arange(1, n + 1, dtype=torch.float64, device=device).pow(-skew)
    cdf = torch.cumsum(w / w.sum(), 0).to(torch.float32)
    keys = torch.empty(0, dtype=torch.int64, device=device)
    want = m
    for _ in range(8):
        need = int((want - keys.numel()) * 1.15) + 1024
        u = torch.rand(2, need, generator=gen, device=device)
        e = torch.searchsorted(cdf, u).clamp_(max=n - 1)
        a, b = torch.minimum(e[0], e[1]), torch.maximum(e[0], e[1])
        k = (a * n + b)[a != b]
        keys = torch.unique(torch.cat([keys, k]))
        if keys.numel() >= want:
            break
    if keys.numel() > want:       # drop a random subset, keep determinism under the generator
        pick = torch.randperm(keys.numel(), generator=gen, device=device)[:want]
        keys = keys[pick.sort().values]
    return torch.stack([keys // n, keys % n])


def make_graph(shape: str = "collab", seed: int = 0, device="cpu", scale: float = 1.0,
               weighted: bool = False, num_nodes: Optional[int] = None,
               num_edges: Optional[int] = None) -> Dict:
    """returns dict(edges=[2,m] undirected pairs (lo,hi), adj_t=Graph (symmetric), weight, data)"""
    cfg = dict(SHAPES[shape])
    n = int(num_nodes if num_nodes is not None else max(16, round(cfg["num_nodes"] * scale)))
    m = int(num_edges if num_edges is not None else max(16, round(cfg["num_edges"] * scale)))
    m = min(m, n * (n - 1) // 2)
    gen = torch.Generator(device=device).manual_seed(20240101 + seed)
    # shuffle node ids so that degree is not monotone in the id
    e = _chung_lu_edges(n, m, cfg["skew"], gen, device)
    relabel = torch.randperm(n, generator=gen, device=device)
    e = relabel[e]
    lo, hi = torch.minimum(e[0], e[1]), torch.maximum(e[0], e[1])
    w = None
    if weighted:
        w = torch.randint(1, 6, (lo.numel(),), generator=gen, device=device).to(torch.float32)
    row, col = torch.cat([lo, hi]), torch.cat([hi, lo])
    adj = Graph.from_coo(row, col, None if w is None else torch.cat([w, w]), n, n)
    r, c, _ = adj.coo()
    data = SyntheticData(adj_t=adj, edge_index=torch.stack([c, r]).cpu(), num_nodes=n)
    return dict(edges=torch.stack([lo, hi], 1), weight=w, adj_t=adj, data=data, num_nodes=n)


def random_walk_pairs(adj: Graph, start: torch.Tensor, walk_length: int, gen: torch.Generator):
    """main.py:241-253 with torch_cluster.random_walk semantics (uniform neighbour,
    stay put on isolated nodes): pairs (start, j-th hop) with weight 1/(j+1),
    self pairs removed."""
    rowptr, col = adj.rowptr, adj.col.to(torch.int64)
    deg = (rowptr[1:] - rowptr[:-1])
    cur = start
    pairs, weights = [], []
    for j in range(walk_length):
        d = deg[cur]
        u = torch.rand(cur.numel(), generator=gen, device=cur.device)
        off = (u * d.to(torch.float32)).to(torch.int64).clamp_(max=(d - 1).clamp(min=0))
        nxt = torch.where(d > 0, col[(rowptr[cur] + off).clamp_(max=col.numel() - 1)], cur)
        pairs.append(torch.stack([start, nxt], 1))
        weights.append(torch.full((start.numel(),), 1.0 / (j + 1), device=start.device))
        cur = nxt
    pairs, weights = torch.cat(pairs), torch.cat(weights)
    keep = pairs[:, 0] != pairs[:, 1]
    return pairs[keep], weights[keep]


def uniform_graph(num_nodes: int, num_edges: int, device, seed: int = 0) -> Graph:
    """symmetric Erdos-Renyi-like graph (no degree skew, no locality): the cache-hostile case
    used for the HBM roofline of the aggregation kernel"""
    gen = torch.Generator(device=device).manual_seed(seed)
    a = torch.randint(0, num_nodes, (num_edges,), generator=gen, device=device)
    b = torch.randint(0, num_nodes, (num_edges,), generator=gen, device=device)
    return Graph.from_coo(torch.cat([a, b]), torch.cat([b, a]), None, num_nodes, num_nodes)


def rmat_graph(scale: int, num_edges: int, device, seed: int = 0, probs=(0.57, 0.19, 0.19, 0.05),
               num_nodes: Optional[int] = None, symmetric: bool = False, chunk: int = 1 << 26,
               permute: bool = True) -> Graph:
    """R-MAT (a, b, c, d) multigraph on the device (BASELINE.json config 5: the skewed, cache-hostile stress for
    the aggregation kernel).  2**scale raw ids, relabelled by a seeded bijection (permute) and folded mod
    num_nodes; duplicates kept (a multigraph row just lists a source twice, like torch_sparse).  The edge stream
    is the counter-hash stream of plnlp_rmat_edges (ops.rmat_edges): a function of (scale, seed, probs, edge id)
    alone, restated bit for bit by the CPU oracle and replayable per row block (rmat_row_block)."""
    from . import ops
    n = int(num_nodes) if num_nodes is not None else 1 << scale
    rows, cols = [], []
    for e0 in range(0, num_edges, chunk):
        r, c = ops.rmat_edges(scale, n, e0, min(chunk, num_edges - e0), seed, device, probs, relabel=permute)
        rows.append(r)
        cols.append(c)
    r, cc = torch.cat(rows), torch.cat(cols)
    if symmetric:
        r, cc = torch.cat([r, cc]), torch.cat([cc, r])
    return Graph.from_coo(r, cc, None, n, n)


def rmat_row_block(scale: int, num_edges: int, num_nodes: int, row_lo: int, n_rows: int, n_cols: int, device,
                   seed: int = 0, probs=(0.57, 0.19, 0.19, 0.05), chunk: int = 1 << 26) -> Graph:
    """destination rows [row_lo, row_lo + n_rows) of the SAME R-MAT multigraph rmat_graph(scale, num_edges,
    num_nodes=..., seed=...) builds -- every rank of a row-sharded run replays the identical edge stream (it
    is a function of the edge id) chunk by chunk and keeps only the entries of its own rows, so no rank ever
    holds the whole edge list (BASELINE.json config 5: 1 B edges over 8 ranks)."""
    from . import ops
    n = int(num_nodes)
    rows, cols = [], []
    for e0 in range(0, num_edges, chunk):
        r, c = ops.rmat_edges(scale, n, e0, min(chunk, num_edges - e0), seed, device, probs)
        keep = (r >= row_lo) & (r < row_lo + n_rows)
        rows.append(r[keep] - row_lo)
        cols.append(c[keep])
        del r, c, keep
    return Graph.from_coo(torch.cat(rows), torch.cat(cols), None, n_rows, n_cols)
