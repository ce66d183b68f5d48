"""VERDICT r5 #4, first look: ddi's aggregation as a dense product.  N = 4 267 at 11.7 % density: the mean aggregation
out = D^-1 A X is a [4267 x 4267] x [4267 x 512] product.  Before a bitmap kernel (A exact in bf16, three MFMAs per block) is
written: what do the EXISTING product kernels make of it, with A as a dense float32 matrix of 1 / deg values (six MFMAs per block,
73 MB of A per launch)?  A third to a half of that time is what a 3-product bitmap kernel could reach."""
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
F = 512
x = torch.randn(n, F, device=dev)
P.ops.tune_aggregation(adj, [F])
want = P.ops.csr_aggregate(adj, x, "mean", False)
t_csr = time_kernel(lambda: P.ops.csr_aggregate(adj, x, "mean", False), iters=20)
print(json.dumps({"what": "csr_aggregate mean (tuned form)", "ms": t_csr * 1e3, "form": P.ops.describe_form(adj._agg_tune.get(F, 0))}), flush=True)
r, c, _ = adj.coo()
a = torch.zeros(n, n, device=dev)
a.index_put_((r.long(), c.long()), torch.ones(r.numel(), device=dev), accumulate=True)
deg = a.sum(1).clamp_min(1.0)
a_norm = a / deg[:, None]
lib = _lib.load()
P.ops.GEMM_MATH["mode"] = "bf16x3"
flop = 2.0 * n * n * F
for name, setup in (("tile kernels (default for 4 267 rows)", lambda: lib.plnlp_gemm_stationary_tuning(0, 16384)),
                    ("gemm_x3s (row threshold lowered)", lambda: lib.plnlp_gemm_stationary_tuning(0, 1024)),
                    ("gemm_x3s, 128-column tiles", lambda: lib.plnlp_gemm_stationary_tuning(4, 1024))):
    setup()
    P.ops.GEMM_STATIONARY_B["min_rows"] = 1024
    got = P.ops.gemm([(a_norm, x)], False, False)
    err = float((got - want).abs().max() / want.abs().max())
    c0 = P.ops.launch_counts()
    t = time_kernel(lambda: P.ops.gemm([(a_norm, x)], False, False), iters=20)
    c1 = P.ops.launch_counts()
    print(json.dumps({"what": "dense product, " + name, "ms": t * 1e3, "rel_err_vs_csr": err, "bf16_TFLOPs_executed": 6 * flop / t / 1e12,
                      "launches": {k: (c1[k] - c0[k]) // 24 for k in c1 if c1[k] != c0[k]}}), flush=True)
lib.plnlp_gemm_stationary_tuning(0, 16384)
