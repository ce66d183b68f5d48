"""The collab step's three dominant launches AS THE STEP MAKES THEM (bench.py::measure_step_launches), alone in a
process -- what scripts/refresh_profiles_r03.sh runs under `rocprofv3 --pmc` to get their HBM / fabric traffic."""
import json, os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import plnlp_amd as P
from plnlp_amd import synthetic

dev = torch.device("cuda", 0)
cfg = bench.WORKLOADS["collab"]
torch.manual_seed(1234); P.manual_seed(1234)
g = synthetic.make_graph("collab", seed=2, device=dev, weighted=True)
n, B, k = g["num_nodes"], cfg["batch"], 1
if os.environ.get("PLNLP_PROBE_RELABEL") == "degree":
    # VERDICT r5 #6, one locality experiment: nodes renumbered by descending degree, so that the hub source rows -- the rows the
    # aggregation gathers most often -- are the FIRST rows of the table / gradient (contiguous, a few MiB) instead of scattered
    # over its 241 MB.  Same graph up to the names of its nodes; everything below is built from the renumbered one.
    from plnlp_amd.graph import Graph
    deg = g["adj_t"].degree()
    new_id = torch.empty(n, dtype=torch.int64, device=dev)
    new_id[torch.argsort(deg, descending=True, stable=True)] = torch.arange(n, device=dev)
    r_, c_, v_ = g["adj_t"].coo()
    adj = Graph.from_coo(new_id[r_.long()], new_id[c_.long()], v_, n, n)
    g["adj_t"] = adj
    g["edges"] = new_id[g["edges"]]
    g["data"].adj_t = adj
    rr, cc, _ = adj.coo()
    g["data"].edge_index = torch.stack([cc, rr]).cpu()
pairs, weights = P.ops.random_walk_pairs(g["adj_t"], g["edges"].reshape(-1), 10, 777)
sel = torch.randperm(pairs.size(0), device=dev)[:4 * B]
pos_all, w_all = pairs[sel], weights[sel]
row, col, _ = g["adj_t"].coo()
neg_all = P.negative_sample.global_neg_sample(torch.stack([col, row]), n, 4 * B, k)
m = P.BaseModel(lr=1e-3, dropout=0.3, grad_clip_norm=1.0, gnn_num_layers=1, mlp_num_layers=2, emb_hidden_channels=256,
                gnn_hidden_channels=256, mlp_hidden_channels=256, num_nodes=n, num_node_feats=0, gnn_encoder_name="SAGE",
                predictor_name="DOT", loss_func="WeightedHingeAUC", optimizer_name="Adam", device=dev, use_node_feats=False,
                train_node_emb=True)
m.param_init(); m.encoder.train()
data = g["data"]
for i in range(3):      # warm: row-split tables, transposed graph, kernel choice
    sl = slice(i * B, (i + 1) * B)
    m.train_step(data, pos_all[sl], neg_all[sl], k, w_all[sl], edges_ready=True)
torch.cuda.synchronize()
out = bench.measure_step_launches(P, m, data, pos_all[:B], neg_all[:B], cfg, dev)
print(json.dumps({k_: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if not isinstance(b, dict)}) for k_, v in out.items()}))
