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
It runs on the device the edge list lives on (torch ops), seeded from the CPU
generator, so ranks still agree.
"""
from typing import Optional

import torch


def add_self_loops(edge_index: torch.Tensor, num_nodes: Optional[int] = None):
    """torch_geometric.utils.add_self_loops: N inferred as max+1 when not given."""
    n = int(edge_index.max()) + 1 if num_nodes is None else int(num_nodes)
    loop = torch.arange(n, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index, loop.unsqueeze(0).expand(2, -1)], dim=1), None


def structured_negative_sampling(edge_index, num_nodes, num_neg_samples, method="sparse",
                                 generator: Optional[torch.Generator] = None):
    """Stand-in for torch_geometric.utils.negative_sampling(method='sparse'):
    candidate ids drawn uniformly from the N*N grid, existing ids removed, up to
    three rounds, truncated to the requested count.  Returns int64 [2, <=n] on
    edge_index's device.

    Pure torch ops on whatever device the edge list lives on: for an epoch of the
    random-walk-augmented collab recipe (11.8 M negatives) the host numpy version
    took 5-10 s -- ten times the epoch's GPU time; on the device it is a sort and a
    binary search (milliseconds).  The private generator is seeded from ONE draw of
    the default CPU generator, so every data-parallel rank that seeded
    torch.manual_seed alike produces the identical tensor."""
    dev = edge_index.device
    gen = generator
    if gen is None:
        gen = torch.Generator(device=dev)
        gen.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
    n = int(num_nodes)
    population = n * n
    present = torch.unique(edge_index[0].to(torch.int64) * n + edge_index[1].to(torch.int64))     # sorted
    want = int(min(num_neg_samples, population - present.numel()))
    density = present.numel() / float(population)
    over = 1.0 / max(1.0 - 1.1 * density, 0.05)

    def absent(sorted_set, ids):
        if sorted_set.numel() == 0:
            return torch.ones_like(ids, dtype=torch.bool)
        pos = torch.searchsorted(sorted_set, ids).clamp_(max=sorted_set.numel() - 1)
        return sorted_set[pos] != ids

    picked = torch.empty(0, dtype=torch.int64, device=dev)
    for _ in range(3):
        need = want - picked.numel()
        if need <= 0:
            break
        k = min(int(over * need * 1.1) + 16, population)
        if population <= (1 << 26):                      # distinct by construction, like random.sample
            draw = torch.randperm(population, generator=gen, device=dev)[:k]
        else:
            draw = torch.unique(torch.randint(0, population, (k,), generator=gen, device=dev, dtype=torch.int64))
            draw = draw[torch.randperm(draw.numel(), generator=gen, device=dev)]       # back to a random order
        draw = draw[absent(present, draw)]
        if picked.numel():
            draw = draw[absent(torch.sort(picked).values, draw)]
        picked = torch.cat([picked, draw])[:want]
    return torch.stack([picked // n, picked % n], dim=0)


def _pad_short(neg_edge: torch.Tensor, want: int):
    """negative_sample.py:11-18 -- top up a short sample with randperm-picked repeats"""
    src, dst = neg_edge[0], neg_edge[1]
    have = neg_edge.size(1)
    if have < want:
        extra = torch.randperm(have)[: want - have].to(src.device)
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
        shuffle = torch.randperm(target_num_sample).to(base_src.device)
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
