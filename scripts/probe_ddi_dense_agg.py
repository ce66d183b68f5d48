"""ddi's aggregation as a dense product.  N = 4 267 at 11.7 % density: the mean aggregation out = D^-1 A X is a
[4267 x 4267] x [4267 x 512] product.  Timed here per launch: the CSR kernels (tuned form), the dense-graph kernel
(csrc/aggregate_dense.hip: bf16 counts x three-term split, three MFMAs per block) and -- the first look of the round -- the same
result on the generic product kernels with A as a float32 matrix (six MFMAs per block, 73 MB of A per launch)."""
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
P.ops.DENSE_AGG["enabled"] = False
P.ops.tune_aggregation(adj, [F])
for dense in (False, True):
    P.ops.DENSE_AGG["enabled"] = dense
    got = P.ops.csr_aggregate(adj, x, "mean", False)
    if not dense:
        want = got
    t = time_kernel(lambda: P.ops.csr_aggregate(adj, x, "mean", False), iters=20)
    tt = time_kernel(lambda: P.ops.csr_aggregate(adj.t_mean(), x, "sum", True), iters=20)
    print(json.dumps({"what": "dense-graph aggregation on the matrix cores (aggregate_dense.hip)" if dense else
                      "CSR kernels, the tuned form: " + P.ops.describe_form(adj._agg_tune.get(F, 0)),
                      "forward_mean_ms": t * 1e3, "transposed_ms": tt * 1e3,
                      "rel_diff_vs_csr": float((got - want).abs().max() / want.abs().max())}), flush=True)
r, c, _ = adj.coo()
a = torch.zeros(n, n, device=dev)
a.index_put_((r.long(), c.long()), torch.ones(r.numel(), device=dev), accumulate=True)
a_norm = a / a.sum(1).clamp_min(1.0)[:, None]
lib = _lib.load()
P.ops.GEMM_MATH["mode"] = "bf16x3"
flop = 2.0 * n * n * F
for name, setup in (("tile kernels (default for 4 267 rows)", lambda: lib.plnlp_gemm_stationary_tuning(0, 16384)),
                    ("gemm_x3s (row threshold lowered)", lambda: lib.plnlp_gemm_stationary_tuning(0, 1024))):
    setup()
    got = P.ops.gemm([(a_norm, x)], False, False)
    t = time_kernel(lambda: P.ops.gemm([(a_norm, x)], False, False), iters=20)
    print(json.dumps({"what": "the same result as ONE generic split-bf16 product, A = float32 matrix of 1 / deg, " + name, "ms": t * 1e3,
                      "rel_diff_vs_csr": float((got - want).abs().max() / want.abs().max()), "bf16_TFLOPs_executed": 6 * flop / t / 1e12}), flush=True)
lib.plnlp_gemm_stationary_tuning(0, 16384)
