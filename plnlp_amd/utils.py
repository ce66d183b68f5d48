"""Edge plumbing and evaluation around the hot path -- plnlp/utils.py restated,
plus the two third-party pieces it leans on that do not exist on the GPU box:
the DataLoader batch permutation and the OGB link-prediction Evaluator."""
from typing import Dict, List, Optional, Tuple

import torch

from .graph import gcn_normalization, adj_normalization  # noqa: F401  (utils.py:83-97)
from .negative_sample import global_neg_sample, global_perm_neg_sample, local_neg_sample


def get_pos_neg_edges(split, split_edge, edge_index=None, num_nodes=None, neg_sampler_name=None, num_neg=None):
    """utils.py:7-41.  'edge'-style splits (ddi, collab) or source/target style
    (citation2); training negatives are sampled, valid/test negatives are given."""
    citation_style = 'edge' not in split_edge['train']
    if citation_style:
        if 'source_node' not in split_edge['train']:
            raise KeyError("split_edge['train'] has neither 'edge' nor 'source_node'")
        source = split_edge[split]['source_node']
        pos_edge = torch.stack([source, split_edge[split]['target_node']]).t()
    else:
        pos_edge = split_edge[split]['edge']

    if split == 'train':
        if neg_sampler_name == 'local':
            neg_edge = local_neg_sample(pos_edge, num_nodes=num_nodes, num_neg=num_neg)
        else:
            sampler = global_neg_sample if neg_sampler_name == 'global' else global_perm_neg_sample
            neg_edge = sampler(edge_index, num_nodes=num_nodes, num_samples=pos_edge.size(0), num_neg=num_neg)
    elif citation_style:
        per_source = split_edge[split]['target_node_neg']
        neg_edge = torch.stack([source.repeat_interleave(per_source.size(1)), per_source.reshape(-1)]).t()
    else:
        neg_edge = split_edge[split]['edge_neg']
    return pos_edge, neg_edge


def batch_permutation(n: int, batch_size: int, shuffle: bool) -> List[torch.Tensor]:
    """The index batches `DataLoader(range(n), batch_size, shuffle)` yields
    (model.py:147,178), reproduced bit-for-bit (fixture G6) without the
    per-item Python collation: the loader's iterator draws a base seed from the
    default CPU generator; a shuffling sampler then draws its own seed from it
    and permutes under a private generator."""
    torch.empty((), dtype=torch.int64).random_()
    if shuffle:
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        order = torch.randperm(n, generator=torch.Generator().manual_seed(seed))
    else:
        order = torch.arange(n)
    return list(order.split(batch_size))


class Evaluator:
    """ogb.linkproppred.Evaluator (1.3.2) semantics for the two metrics PLNLP
    uses.  `eval({'y_pred_pos', 'y_pred_neg'})` -> {'hits@K': float} or
    {'mrr_list': tensor, ...}."""

    def __init__(self, name: str = 'ogbl-ddi', K: Optional[int] = None):
        self.name = name
        defaults = {'ogbl-ddi': ('hits', 20), 'ogbl-collab': ('hits', 50), 'ogbl-ppa': ('hits', 100),
                    'ogbl-citation2': ('mrr', None)}
        self.eval_metric, k = defaults.get(name, ('hits', 50))
        self.K = K if K is not None else k

    def eval(self, input_dict: Dict[str, torch.Tensor]) -> Dict:
        pos, neg = input_dict['y_pred_pos'], input_dict['y_pred_neg']
        if neg.dim() == 2:
            return self._mrr(pos, neg)
        return {f'hits@{self.K}': self._hits(pos, neg, self.K)}

    @staticmethod
    def _hits(pos, neg, k):
        if len(neg) < k:
            return 1.0
        kth = torch.topk(neg, k)[0][-1]
        return float(torch.sum(pos > kth).cpu()) / len(pos)

    @staticmethod
    def _mrr(pos, neg):
        scores = torch.cat([pos.view(-1, 1), neg], dim=1)
        # ogb 1.3.2: argsort(descending) then the position of column 0.  stable=True pins what a tie does
        # (the positive, column 0, stays ahead of equal negatives -- what the CPU sort the reference runs
        # does) so the device path ranks ties like the host path
        order = torch.argsort(scores, dim=1, descending=True, stable=True)
        rank = torch.nonzero(order == 0, as_tuple=False)[:, 1] + 1
        rr = 1.0 / rank.to(torch.float)
        return {'mrr_list': rr, 'hits@1_list': (rank <= 1).float(), 'hits@3_list': (rank <= 3).float(),
                'hits@10_list': (rank <= 10).float()}


def evaluate_hits(evaluator, pos_val_pred, neg_val_pred, pos_test_pred, neg_test_pred):
    """utils.py:44-60"""
    results = {}
    for K in (20, 50, 100):
        evaluator.K = K
        scores = []
        for pos, neg in ((pos_val_pred, neg_val_pred), (pos_test_pred, neg_test_pred)):
            scores.append(evaluator.eval({'y_pred_pos': pos, 'y_pred_neg': neg})[f'hits@{K}'])
        results[f'Hits@{K}'] = tuple(scores)
    return results


def evaluate_mrr(evaluator, pos_val_pred, neg_val_pred, pos_test_pred, neg_test_pred):
    """utils.py:63-80"""
    scores = []
    for pos, neg in ((pos_val_pred, neg_val_pred), (pos_test_pred, neg_test_pred)):
        out = evaluator.eval({'y_pred_pos': pos, 'y_pred_neg': neg.view(pos.shape[0], -1)})
        scores.append(out['mrr_list'].mean().item())
    return {'MRR': tuple(scores)}


def host_cpu_budget() -> int:
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota.  A container that
    shows 256 cores behind a 16-CPU quota freezes the whole process for the rest of every 100 ms period once its
    threads -- OpenMP workers spinning after a parallel region included -- have burnt the quota (measured on the
    MI355X boxes of this project: 128 torch threads turned a steady 41 ms collab epoch into 45-100 ms)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                                            # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:                                                        # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def limit_host_threads(share: int = 1) -> int:
    """cap torch's intra-op threads at this process's share of host_cpu_budget() (share = ranks on the node);
    returns the thread count in force.  Called by the drivers (train.py, bench.py), never by the library itself."""
    import torch
    t = max(1, min(torch.get_num_threads(), host_cpu_budget() // max(1, share)))
    torch.set_num_threads(t)
    return t
