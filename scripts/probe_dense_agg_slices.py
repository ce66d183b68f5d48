"""ddi's aggregation on the matrix cores (csrc/aggregate_dense.hip): per-launch time against the number of K slices
(plnlp_dense_aggregate_tuning; 0 = the library's rule)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import plnlp_amd as P
from plnlp_amd import synthetic, _lib
from bench import time_kernel

dev = torch.device("cuda")
g = synthetic.make_graph("ddi", seed=2, device=dev, weighted=False)
adj, n = g["adj_t"], g["num_nodes"]
x = torch.randn(n, 512, device=dev)
lib = _lib.load()
P.ops.DENSE_AGG["enabled"] = False
want = P.ops.csr_aggregate(adj, x, "mean", False)
P.ops.DENSE_AGG["enabled"] = True
for s in (0, 2, 3, 4, 5, 6, 7, 8, 0):
    lib.plnlp_dense_aggregate_tuning(s)
    got = P.ops.csr_aggregate(adj, x, "mean", False)
    t = time_kernel(lambda: P.ops.csr_aggregate(adj, x, "mean", False), iters=30)
    tt = time_kernel(lambda: P.ops.csr_aggregate(adj.t_mean(), x, "sum", True), iters=30)
    print(json.dumps({"slices": s or "rule", "forward_mean_ms": round(t * 1e3, 4), "transposed_ms": round(tt * 1e3, 4),
                      "rel_diff_vs_csr": float((got - want).abs().max() / want.abs().max())}), flush=True)
lib.plnlp_dense_aggregate_tuning(0)
