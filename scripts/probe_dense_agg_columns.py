"""column-coherent error of the dense aggregation (a bias gradient downstream sums every ROW of a column): per forced slice count,
max over columns of |sum_r err[r, c]| / sum_r |want[r, c]|, forward mean and the transposed form, against float64"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plnlp_amd as P
from plnlp_amd import synthetic, _lib
dev = torch.device("cuda")
g = synthetic.make_graph("ddi", seed=2, device=dev, weighted=False)
adj, n = g["adj_t"], g["num_nodes"]
r, c, _ = adj.coo()
a64 = torch.zeros(n, n, dtype=torch.float64, device=dev)
a64.view(-1).index_add_(0, r.long() * n + c.long(), torch.ones(r.numel(), dtype=torch.float64, device=dev))
deg = a64.sum(1).clamp_min(1.0)
x = torch.randn(n, 512, device=dev).abs()
want = (a64 @ x.double()) / deg[:, None]
want_t = a64.t() @ (x.double() / deg[:, None])
lib = _lib.load()
for s in [int(v) for v in os.environ.get("PROBE_SLICES", "4,3,5,7,0").split(",")]:
    lib.plnlp_dense_aggregate_tuning(s)
    out = {"slices": s}
    for name, fn, w in (("forward", lambda: P.ops.csr_aggregate(adj, x, "mean", False), want),
                        ("transposed", lambda: P.ops.csr_aggregate(adj.t_mean(), x, "sum", True), want_t)):
        for dense in (True, False):
            P.ops.DENSE_AGG["enabled"] = dense
            err = fn().double() - w
            col = (err.sum(0).abs() / w.abs().sum(0))
            row = (err.sum(1).abs() / w.abs().sum(1))
            out["%s_%s" % (name, "dense" if dense else "csr")] = {"col_max": float(col.max()), "col_mean": float(col.mean()),
                                                                   "row_max": float(row.max()), "max_abs": float(err.abs().max() / w.abs().max())}
    P.ops.DENSE_AGG["enabled"] = True
    print(json.dumps(out), flush=True)
lib.plnlp_dense_aggregate_tuning(0)
