"""is the dense aggregation's error BIASED?  (the full-size ddi step's scorer-bias gradient moved by 2e-3 of its scale with it):
signed error statistics of the dense (MFMA) and the CSR aggregation against float64 on the ddi-shaped graph"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plnlp_amd as P
from plnlp_amd import synthetic
dev = torch.device("cuda")
g = synthetic.make_graph("ddi", seed=2, device=dev, weighted=False)
adj, n = g["adj_t"], g["num_nodes"]
r, c, _ = adj.coo()
a64 = torch.zeros(n, n, dtype=torch.float64, device=dev)
a64.view(-1).index_add_(0, r.long() * n + c.long(), torch.ones(r.numel(), dtype=torch.float64, device=dev))
deg = a64.sum(1).clamp_min(1.0)
for name, x in (("randn", torch.randn(n, 512, device=dev)), ("positive (relu-like)", torch.randn(n, 512, device=dev).abs()),
                ("randn + 3", torch.randn(n, 512, device=dev) + 3.0)):
    want = (a64 @ x.double()) / deg[:, None]
    for on in (True, False):
        P.ops.DENSE_AGG["enabled"] = on
        got = P.ops.csr_aggregate(adj, x, "mean", False).double()
        err = got - want
        print(json.dumps({"x": name, "dense": on, "mean_signed_err_over_mean_abs": float((err * want.sign()).mean() / want.abs().mean()),
                          "rms_err_over_rms": float(err.pow(2).mean().sqrt() / want.pow(2).mean().sqrt()),
                          "max_err_over_max": float(err.abs().max() / want.abs().max())}), flush=True)
