"""Negative-edge index generation, plnlp/negative_sample.py restated.

Host-side, once per epoch, on the default CPU generator exactly as the
reference does, so that every data-parallel rank regenerates the identical
[E, k, 2] tensor from the same seed ("bit-exact per GPU").

`local` is pure torch and is bit-exact against the reference (fixtures G4/G7).
`global` depends on torch_geometric.utils.negative_sampling, whose stream
(Python `random.sample` + numpy) is not reproducible here (SURVEY.md Appendix
A.4): `structured_negative_sampling` keeps its contract -- no returned pair is
an existing edge or a self loop, no duplicates, at most the requested count --
and the reference's own padding / reshaping around it is bit-exact (fixture G5).
"""
from typing import Optional

import numpy as np
import torch


def add_self_loops(edge_index: torch.Tensor, num_nodes: Optional[int] = None):
    """torch_geometric.utils.add_self_loops: N inferred as max+1 when not given."""
    n = int(edge_index.max()) + 1 if num_nodes is None else int(num_nodes)
    loop = torch.arange(n, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index, loop.unsqueeze(0).expand(2, -1)], dim=1), None


def structured_negative_sampling(edge_index, num_nodes, num_neg_samples, method="sparse",
                                 generator: Optional[np.random.Generator] = None):
    """Stand-in for torch_geometric.utils.negative_sampling(method='sparse'):
    candidate ids drawn uniformly from the N*N grid, existing ids removed, up to
    three rounds, truncated to the requested count.  Returns int64 [2, <=n]."""
    rng = generator if generator is not None else np.random.default_rng(
        int(torch.empty((), dtype=torch.int64).random_().item()))
    n = int(num_nodes)
    population = n * n
    present = np.unique((edge_index[0].to(torch.int64) * n + edge_index[1].to(torch.int64)).cpu().numpy())
    want = int(min(num_neg_samples, population - present.size))
    density = present.size / float(population)
    over = 1.0 / max(1.0 - 1.1 * density, 0.05)
    picked = np.empty(0, dtype=np.int64)
    for _ in range(3):
        need = want - picked.size
        if need <= 0:
            break
        k = min(int(over * need * 1.1) + 16, population)
        if population <= (1 << 26):                      # distinct by construction, like random.sample
            draw = rng.choice(population, size=k, replace=False).astype(np.int64)
        else:
            draw = rng.integers(0, population, size=k, dtype=np.int64)
            _, first = np.unique(draw, return_index=True)
            draw = draw[np.sort(first)]                  # distinct, arrival order kept
        draw = draw[~np.isin(draw, present, assume_unique=False)]
        if picked.size:
            draw = draw[~np.isin(draw, picked)]
        picked = np.concatenate([picked, draw])[:want]
    out = torch.from_numpy(picked)
    return torch.stack([out // n, out % n], dim=0)


def _pad_short(neg_edge: torch.Tensor, want: int):
    """negative_sample.py:11-18 -- top up a short sample with randperm-picked repeats"""
    src, dst = neg_edge[0], neg_edge[1]
    have = neg_edge.size(1)
    if have < want:
        extra = torch.randperm(have)[: want - have]
        src, dst = torch.cat((src, src[extra])), torch.cat((dst, dst[extra]))
    return src, dst


def global_neg_sample(edge_index, num_nodes, num_samples, num_neg, method="sparse"):
    """negative_sample.py:6-20"""
    with_loops, _ = add_self_loops(edge_index)
    neg_edge = structured_negative_sampling(with_loops, num_nodes, num_samples * num_neg, method)
    src, dst = _pad_short(neg_edge, num_samples * num_neg)
    return torch.stack((src, dst), dim=-1).reshape(-1, num_neg, 2)


def sample_perm_copy(edge_index, target_num_sample, num_perm_copy):
    """negative_sample.py:61-76"""
    src, dst = _pad_short(edge_index, target_num_sample)
    base_src, base_dst = src, dst
    for _ in range(num_perm_copy - 1):
        shuffle = torch.randperm(target_num_sample)
        src, dst = torch.cat((src, base_src[shuffle])), torch.cat((dst, base_dst[shuffle]))
    return torch.stack((src, dst), dim=-1).reshape(-1, num_perm_copy, 2)


def global_perm_neg_sample(edge_index, num_nodes, num_samples, num_neg, method="sparse"):
    """negative_sample.py:23-28"""
    with_loops, _ = add_self_loops(edge_index)
    neg_edge = structured_negative_sampling(with_loops, num_nodes, num_samples, method)
    return sample_perm_copy(neg_edge, num_samples, num_neg)


def local_neg_sample(pos_edges, num_nodes, num_neg, random_src=False):
    """negative_sample.py:31-43: keep one endpoint, draw the other uniformly from
    all nodes with ONE torch.randint call (no filtering of true edges / self pairs)."""
    count = pos_edges.size(0)
    if random_src:
        side = torch.randint(0, 2, (count,), dtype=torch.long)
        anchor = pos_edges[torch.arange(count), side]
    else:
        anchor = pos_edges[:, 0]
    anchor = anchor.reshape(-1, 1).repeat(1, num_neg).reshape(-1)
    other = torch.randint(0, num_nodes, (num_neg * count,), dtype=torch.long)
    return torch.stack((anchor, other), dim=-1).reshape(-1, num_neg, 2)
