"""The aggregation tuner's measurements on the collab-shaped graph, repeated: how close are the forms, how often does the pick flip?
usage: probe_agg_tuner.py [repeats]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plnlp_amd as P
from plnlp_amd import ops, synthetic

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = synthetic.make_graph("collab", seed=2, device=torch.device("cuda"), weighted=True)["adj_t"]
graph = ops.as_graph(g) if hasattr(ops, "as_graph") else g
feat = 256
x = torch.randn(graph.n_cols, feat, device="cuda")
out = torch.empty(graph.n_rows, feat, device="cuda")
cands = list(ops.AGG_AUTOTUNE["candidates"])
for r in range(reps):
    t = {}
    for c in cands:
        ops.csr_aggregate(graph, x, "mean", False, out=out, tune=c)
    for c in cands:
        best = float("inf")
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); ops.csr_aggregate(graph, x, "mean", False, out=out, tune=c); e.record(); e.synchronize()
            best = min(best, s.elapsed_time(e))
        t[c] = round(best * 1e3, 1)
    pick = ops._time_agg_forms(graph, x, out, "mean", False, None, None)
    print(json.dumps({"rep": r, "us_by_form": t, "pick": pick}), flush=True)
