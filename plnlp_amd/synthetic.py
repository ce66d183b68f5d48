"""Synthetic OGB-shaped inputs.  No dataset files exist on the GPU
box and there is no network, so benchmarks and parity runs use seeded graphs
with the node / edge counts and degree skew of the OGB datasets the reference's
recipes name (README.md:24,31,35,40).  Pure torch; runs on CPU or GPU."""
from __future__ import annotations

import math
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


def community_graph(num_nodes: int = 20000, community: int = 25, p_in: float = 0.5, cross_per_node: float = 1.0,
                    seed: int = 0, device="cpu", holdout: float = 0.1, unlearnable: float = 0.0) -> Dict:
    """A LEARNABLE link-prediction problem (stochastic block model): nodes sit in communities of `community`
    nodes, every intra-community pair is an edge with probability p_in, plus cross_per_node random cross edges per
    node.  A fraction `holdout` of the intra-community edges is held out (valid / test positives); negatives are
    uniform random pairs.  An encoder + predictor trained on the remaining edges recovers the communities and ranks
    the held-out edges far above random pairs: Hits@K lands in the 60-95 % range the reference reports on OGB
    (README.md:7-10) instead of the few percent of a random graph -- the regime in which "Hits@K within 0.3 points"
    is a statement about the arithmetic and not about which of 10 000 negatives happens to be 20th.
    unlearnable: that fraction of the valid / test positives is replaced by uniform random non-edges -- pairs no model
    can tell from the negatives -- so a CONVERGED model plateaus at (1 - unlearnable) instead of 100 %: a stable,
    non-trivial ceiling set by the data (the reference's own numbers sit at such plateaus: 90.9 % on ddi).
    returns dict(num_nodes, train=[m,2], valid=[v,2], test=[v,2], valid_neg, test_neg, adj_t (train edges,
    symmetric), data)"""
    gen = torch.Generator(device=device).manual_seed(777_000 + seed)
    n, c = int(num_nodes), int(community)
    n_comm = n // c
    # all intra-community pairs (i < j), Bernoulli(p_in)
    iu = torch.triu_indices(c, c, offset=1, device=device)
    base = (torch.arange(n_comm, device=device) * c)[:, None]
    a = (base + iu[0][None, :]).reshape(-1)
    b = (base + iu[1][None, :]).reshape(-1)
    keep = torch.rand(a.numel(), generator=gen, device=device) < p_in
    a, b = a[keep], b[keep]
    # shuffle node ids so that community membership is not visible in the id
    relabel = torch.randperm(n, generator=gen, device=device)
    a, b = relabel[a], relabel[b]
    perm = torch.randperm(a.numel(), generator=gen, device=device)
    a, b = a[perm], b[perm]
    n_hold = int(a.numel() * holdout) // 2 * 2
    held = torch.stack([a[:n_hold], b[:n_hold]], 1)
    ta, tb = a[n_hold:], b[n_hold:]
    m_cross = int(n * cross_per_node)
    ca = torch.randint(0, n, (m_cross,), generator=gen, device=device)
    cb = torch.randint(0, n, (m_cross,), generator=gen, device=device)
    ok = ca != cb
    ta, tb = torch.cat([ta, ca[ok]]), torch.cat([tb, cb[ok]])
    lo, hi = torch.minimum(ta, tb), torch.maximum(ta, tb)
    key = torch.unique(lo * n + hi)
    lo, hi = key // n, key % n
    order = torch.randperm(lo.numel(), generator=gen, device=device)
    lo, hi = lo[order], hi[order]
    adj = Graph.from_coo(torch.cat([lo, hi]), torch.cat([hi, lo]), None, n, n)
    r, cc, _ = adj.coo()
    data = SyntheticData(adj_t=adj, edge_index=torch.stack([cc, r]).cpu(), num_nodes=n)
    # negatives: uniform random NON-edges (OGB's valid / test negatives are true non-edges; a random pair that is a
    # training edge would rank like a positive and cap Hits@K from above)
    n_neg = 10000
    cand = torch.randint(0, n, (3 * n_neg, 2), generator=gen, device=device)
    cl, ch = torch.minimum(cand[:, 0], cand[:, 1]), torch.maximum(cand[:, 0], cand[:, 1])
    known = torch.unique(torch.cat([key, torch.minimum(held[:, 0], held[:, 1]) * n + torch.maximum(held[:, 0], held[:, 1])]))
    ck = cl * n + ch
    pos_in = torch.searchsorted(known, ck).clamp_(max=known.numel() - 1)
    negs = cand[(known[pos_in] != ck) & (cl != ch)][: 2 * n_neg]
    assert negs.size(0) == 2 * n_neg
    valid, test = held[: n_hold // 2].clone(), held[n_hold // 2:].clone()
    if unlearnable > 0.0:
        extra = torch.randint(0, n, (4 * n_hold + 64, 2), generator=gen, device=device)
        el, eh = torch.minimum(extra[:, 0], extra[:, 1]), torch.maximum(extra[:, 0], extra[:, 1])
        ek = el * n + eh
        pe = torch.searchsorted(known, ek).clamp_(max=known.numel() - 1)
        extra = extra[(known[pe] != ek) & (el != eh)]
        for part in (valid, test):
            cnt = int(part.size(0) * unlearnable)
            part[:cnt] = extra[:cnt]
            extra = extra[cnt:]
    return dict(num_nodes=n, train=torch.stack([lo, hi], 1), valid=valid, test=test,
                valid_neg=negs[:n_neg], test_neg=negs[n_neg:], adj_t=adj, data=data)


def geometric_graph(num_nodes: int = 3000, avg_degree: float = 30.0, softness: float = 0.15, holdout: float = 0.15,
                    n_neg: int = 10000, seed: int = 0, dim: int = 2) -> Dict:
    """A link-prediction problem whose Hits@K does NOT saturate and DEPENDS ON RANKING QUALITY: a soft random geometric
    graph.  Nodes get latent positions on the unit torus; a pair at distance d is an edge with probability
    sigmoid((r - d) / (softness r)), r set for the wanted average degree.  Held-out edges (valid / test positives) are
    mostly close pairs, the negatives are uniform random NON-edges -- a few of which are close pairs too, and those are
    the ones the K-th-negative threshold of Hits@K lands on: how many positives clear it depends on how well the
    trained embedding has recovered the geometry (a stochastic block model cannot do this: inside a block held-out
    edges and non-edges are exchangeable, so Hits@K against same-block negatives is chance whatever the model).
    With the defaults a converged SAGE + DOT model sits near Hits@20 / @50 / @100 = 60 / 80 / 92 %, an MLP scorer
    higher -- the 60-90 % band the reference reports on OGB (README.md:7-10), with nothing pinned at a ceiling.
    Host tensors only (the all-pairs pass is O(n^2): a test-sized generator).
    returns dict(num_nodes, train [m,2], valid [v,2], test [v,2], valid_neg, test_neg [n_neg,2], adj_t (train edges,
    symmetric), data, pos (latent positions))"""
    gen = torch.Generator().manual_seed(888_000 + seed)
    n = int(num_nodes)
    z = torch.rand(n, dim, generator=gen, dtype=torch.float64)
    vol = {1: 2.0, 2: math.pi, 3: 4.0 / 3.0 * math.pi}[dim]
    r = (avg_degree / (vol * n)) ** (1.0 / dim)
    iu = torch.triu_indices(n, n, offset=1)
    d = (z[iu[0]] - z[iu[1]]).abs()
    d = torch.minimum(d, 1.0 - d).pow(2).sum(1).sqrt()
    p = torch.sigmoid((r - d) / (softness * r)) if softness > 0 else (d < r).double()
    keep = torch.rand(d.numel(), generator=gen, dtype=torch.float64) < p
    a, b = iu[0][keep], iu[1][keep]
    perm = torch.randperm(a.numel(), generator=gen)
    a, b = a[perm], b[perm]
    n_hold = int(a.numel() * holdout) // 2 * 2
    held = torch.stack([a[:n_hold], b[:n_hold]], 1)
    lo, hi = a[n_hold:], b[n_hold:]                        # (a < b by construction, pairs distinct)
    adj = Graph.from_coo(torch.cat([lo, hi]), torch.cat([hi, lo]), None, n, n)
    rr, cc, _ = adj.coo()
    data = SyntheticData(adj_t=adj, edge_index=torch.stack([cc, rr]).cpu(), num_nodes=n)
    cand = torch.randint(0, n, (3 * n_neg, 2), generator=gen)
    cl, ch = torch.minimum(cand[:, 0], cand[:, 1]), torch.maximum(cand[:, 0], cand[:, 1])
    known = torch.unique(a * n + b)                        # every edge, held-out ones included
    ck = cl * n + ch
    at = torch.searchsorted(known, ck).clamp_(max=known.numel() - 1)
    negs = cand[(known[at] != ck) & (cl != ch)][: 2 * n_neg]
    assert negs.size(0) == 2 * n_neg
    return dict(num_nodes=n, train=torch.stack([lo, hi], 1), valid=held[: n_hold // 2].clone(),
                test=held[n_hold // 2:].clone(), valid_neg=negs[:n_neg], test_neg=negs[n_neg:], adj_t=adj, data=data,
                pos=z)


def geometric_graph_blocked(num_nodes: int = 40000, avg_degree: float = 20.0, softness: float = 0.15, holdout: float = 0.15,
                            n_neg: int = 10000, seed: int = 0, n_hubs: int = 0, hub_degree: float = 1500.0,
                            block: int = 1000) -> Dict:
    """geometric_graph for tens of thousands of nodes (the all-pairs pass goes block of rows by block of rows, so memory
    is O(block * n) instead of O(n^2)), optionally with HUBS: `n_hubs` nodes whose connection radius is scaled so that
    their expected degree is `hub_degree` -- a pair's radius is the larger of its endpoints' -- which gives the
    aggregation kernel rows beyond its long-row threshold while every edge still follows the latent geometry.  Same
    problem statement and return value as geometric_graph; a different random stream (not the same graph at equal
    arguments).  2-d torus."""
    gen = torch.Generator().manual_seed(889_000 + seed)
    n = int(num_nodes)
    z = torch.rand(n, 2, generator=gen, dtype=torch.float64)
    r0 = (avg_degree / (math.pi * n)) ** 0.5
    rad = torch.full((n,), r0, dtype=torch.float64)
    if n_hubs > 0:
        hubs = torch.randperm(n, generator=gen)[:n_hubs]
        rad[hubs] = (hub_degree / (math.pi * n)) ** 0.5
    aa, bb = [], []
    for lo_i in range(0, n, block):
        hi_i = min(n, lo_i + block)
        d = (z[lo_i:hi_i, None, :] - z[None, :, :]).abs()
        d = torch.minimum(d, 1.0 - d).pow(2).sum(2).sqrt()                    # [block, n]
        rr = torch.maximum(rad[lo_i:hi_i, None], rad[None, :])
        p = torch.sigmoid((rr - d) / (softness * rr)) if softness > 0 else (d < rr).double()
        keep = torch.rand(d.shape, generator=gen, dtype=torch.float64) < p
        i, j = keep.nonzero(as_tuple=True)
        i = i + lo_i
        up = i < j                                                             # each unordered pair is decided once
        aa.append(i[up])
        bb.append(j[up])
    a, b = torch.cat(aa), torch.cat(bb)
    perm = torch.randperm(a.numel(), generator=gen)
    a, b = a[perm], b[perm]
    n_hold = int(a.numel() * holdout) // 2 * 2
    held = torch.stack([a[:n_hold], b[:n_hold]], 1)
    lo, hi = a[n_hold:], b[n_hold:]
    adj = Graph.from_coo(torch.cat([lo, hi]), torch.cat([hi, lo]), None, n, n)
    rr_, cc_, _ = adj.coo()
    data = SyntheticData(adj_t=adj, edge_index=torch.stack([cc_, rr_]).cpu(), num_nodes=n)
    cand = torch.randint(0, n, (3 * n_neg, 2), generator=gen)
    cl, ch = torch.minimum(cand[:, 0], cand[:, 1]), torch.maximum(cand[:, 0], cand[:, 1])
    known = torch.unique(a * n + b)
    ck = cl * n + ch
    at = torch.searchsorted(known, ck).clamp_(max=known.numel() - 1)
    negs = cand[(known[at] != ck) & (cl != ch)][: 2 * n_neg]
    assert negs.size(0) == 2 * n_neg
    return dict(num_nodes=n, train=torch.stack([lo, hi], 1), valid=held[: n_hold // 2].clone(),
                test=held[n_hold // 2:].clone(), valid_neg=negs[:n_neg], test_neg=negs[n_neg:], adj_t=adj, data=data,
                pos=z)
