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


_perm_host = {"buf": None, "owner": None}


class StreamedPermutation:
    """The same index batches as batch_permutation(n, batch_size, True) -- i.e. as the reference's
    `DataLoader(range(n), batch_size, shuffle=True)` (model.py:147), bit for bit -- delivered WHILE the shuffle is
    still running.

    torch.randperm on the CPU generator is a forward Fisher-Yates shuffle: entry i is final after iteration i.  The
    reference shuffles the whole epoch before its first step; at the collab recipe's 23 M random-walk pairs that is
    0.8 s of host time in front of a 0.6 s GPU epoch.  Here a host thread runs the shuffle in slices of a few batches
    through the library's host entry points (plnlp_host_randperm_*: same MT19937 stream, same swaps, a third of the
    time) into pinned memory and copies every finished slice to the device on its own stream; `batch(i, stream)`
    blocks only until slice i exists and makes `stream` wait for its copy.  The default CPU generator is consumed
    exactly as the loader consumes it (base seed, sampler seed), so everything drawn after it stays in step."""

    LIMIT = 0xFFFFFFFF // 20          # ATen's own range for this algorithm (randperm_cpu)

    def __init__(self, n: int, batch_size: int, device, chunk_batches: int = 4):
        import threading
        from . import _lib as L
        self._L, self._lib = L, L.load()
        self.n, self.batch_size = int(n), int(batch_size)
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        assert self.device.type == "cuda" and 0 < self.n < self.LIMIT
        torch.empty((), dtype=torch.int64).random_()                                   # the loader's base-seed draw
        seed = int(torch.empty((), dtype=torch.int64).random_().item())               # the sampler's seed
        self.sizes = [min(self.batch_size, self.n - lo) for lo in range(0, self.n, self.batch_size)]
        prev = _perm_host.get("owner")        # an epoch that ended in an exception left its thread shuffling into the
        if prev is not None:                  # pinned buffer this instance is about to re-initialise: stop it first
            prev.stop()
        _perm_host["owner"] = self
        buf = _perm_host["buf"]
        if buf is None or buf.numel() < self.n:
            buf = _perm_host["buf"] = torch.empty(max(self.n, 1 << 20), dtype=torch.int64, pin_memory=True)
        self._host = buf[: self.n]
        self._mt = torch.empty(625, dtype=torch.int32)
        self._copy = _copy_stream(self.device)
        self._copy.synchronize()              # a previous epoch's copies out of the shared pinned buffer are done
        L.check(self._lib.plnlp_host_randperm_init(seed & 0xFFFFFFFFFFFFFFFF, self.n, self._host.data_ptr(),
                                                   self._mt.data_ptr()), "plnlp_host_randperm_init")
        self.order = torch.empty(self.n, dtype=torch.int64, device=self.device)
        # `order` was allocated on the caller's stream and is first WRITTEN on the copy stream: whatever that stream still
        # has queued on a recycled allocator block must run before the copies do
        self._copy.wait_stream(torch.cuda.current_stream(self.device))
        self._stop = threading.Event()
        self._chunk = max(1, int(chunk_batches)) * self.batch_size
        self._cv = threading.Condition()
        self._chunks = []                     # (end index, event of its copy), in order
        self._error = None
        self._waited = 0                      # chunks whose event the consumer already waited for, per stream id
        self._seen = {}
        self._thread = threading.Thread(target=self._run, name="plnlp-permutation", daemon=True)
        self._thread.start()

    def _run(self):
        try:
            torch.cuda.set_device(self.device)
            pos = 0
            while pos < self.n and not self._stop.is_set():
                # the first slice is one batch (the first step can start at once), the rest a few batches each
                to = min(self.n, pos + (self.batch_size if pos == 0 else self._chunk))
                self._L.check(self._lib.plnlp_host_randperm_advance(self.n, self._host.data_ptr(), self._mt.data_ptr(),
                                                                    pos, to), "plnlp_host_randperm_advance")
                with torch.cuda.stream(self._copy):
                    self.order[pos:to].copy_(self._host[pos:to], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self._copy)
                with self._cv:
                    self._chunks.append((to, ev))
                    self._cv.notify_all()
                pos = to
        except BaseException as exc:          # surfaced by the consumer
            with self._cv:
                self._error = exc
                self._cv.notify_all()

    def batch(self, i: int, stream=None) -> torch.Tensor:
        """index batch i (a view of the device-resident order); `stream` (default: the current one) is made to wait
        for the copy that delivered it"""
        lo = i * self.batch_size
        hi = lo + self.sizes[i]
        with self._cv:
            while self._error is None and (not self._chunks or self._chunks[-1][0] < hi):
                self._cv.wait()
            if self._error is not None:
                raise self._error
            upto = len(self._chunks)
            need = next(j for j, (end, _) in enumerate(self._chunks) if end >= hi)
            ev = self._chunks[need][1]
        stream = torch.cuda.current_stream(self.device) if stream is None else stream
        key = stream.cuda_stream
        if self._seen.get(key, -1) < need:       # copies complete in order: the event of the covering slice suffices
            stream.wait_event(ev)
            self._seen[key] = need
        return self.order[lo:hi]

    def stop(self):
        """abandon the shuffle (checked per slice) and wait for the thread"""
        self._stop.set()
        self._thread.join()
        if _perm_host.get("owner") is self:
            _perm_host["owner"] = None

    def join(self):
        self._thread.join()
        if _perm_host.get("owner") is self:
            _perm_host["owner"] = None
        if self._error is not None:
            raise self._error


_copy_streams = {}


def _copy_stream(device):
    key = torch.device(device).index or 0
    if key not in _copy_streams:
        _copy_streams[key] = torch.cuda.Stream(device=device)
    return _copy_streams[key]


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
