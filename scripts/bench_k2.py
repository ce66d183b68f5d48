"""Micro-benchmark of the transposed (backward) aggregation variants on the collab-shaped graph: what the
weights, the source map and the addend epilogue each cost next to the plain forward gather."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plnlp_amd as P
from plnlp_amd import synthetic, _lib as L
from bench import time_kernel

dev = torch.device("cuda")
g = synthetic.make_graph("collab", seed=2, device=dev, weighted=False)["adj_t"]
n, F = g.n_rows, 256
gen = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(n, F, device=dev, generator=gen)
out = torch.empty(n, F, device=dev)
gt, gtm = g.t(), g.t_mean()
for frac in (0.55, 1.0):
    keep = torch.rand(n, device=dev, generator=gen) < frac
    rows = torch.nonzero(keep).reshape(-1)
    nmap = torch.full((n,), -1, dtype=torch.int32, device=dev)
    nmap[rows] = torch.arange(rows.numel(), dtype=torch.int32, device=dev)
    xc = x[rows].contiguous()
    add = torch.randn(rows.numel(), F, device=dev, generator=gen)
    epi = L.make_epilogue(addend=add, addend_index=nmap)
    cases = {
        "forward mean (unweighted)": lambda: P.ops.csr_aggregate(g, x, "mean", False, out=out),
        "transposed, weighted (A^T D^-1), dense source": lambda: P.ops.csr_aggregate(gtm, x, "sum", True, out=out),
        "  + source map": lambda: P.ops.csr_aggregate(gtm, xc, "sum", True, out=out, src_map=nmap),
        "  + source map + addend": lambda: P.ops.csr_aggregate(gtm, xc, "sum", True, out=out, src_map=nmap, epilogue=epi),
        "transposed, unweighted + source map": lambda: P.ops.csr_aggregate(gt, xc, "sum", False, out=out, src_map=nmap),
    }
    for name, fn in cases.items():
        t = time_kernel(fn, iters=20)
        print(json.dumps({"mapped_fraction": frac, "case": name, "ms": round(t * 1e3, 4)}), flush=True)
