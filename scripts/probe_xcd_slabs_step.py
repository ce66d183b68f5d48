"""The collab step's two aggregation launches (bench.measure_step_launches) with the eight XCD-pinned column slabs (PLNLP_AGG_SLABS_XCD)
forced on the transposed mapped launch / on both, against the pinned forms -- an experiment, nothing here is a product path."""
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
for i in range(3):
    sl = slice(i * B, (i + 1) * B)
    m.train_step(data, pos_all[sl], neg_all[sl], k, w_all[sl], edges_ready=True)
torch.cuda.synchronize()
mapped0, tune0 = P.ops.mapped_form, P.ops._agg_tune
for name, mp, tn in (("pinned forms", mapped0, tune0), ("transposed launch in XCD slabs", lambda t: 64, tune0),
                     ("both launches in XCD slabs", lambda t: 64, lambda *a, **kw: 64), ("pinned forms again", mapped0, tune0)):
    P.ops.mapped_form, P.ops._agg_tune = mp, tn
    c0 = P.ops.launch_counts()
    out = bench.measure_step_launches(P, m, data, pos_all[:B], neg_all[:B], cfg, dev)
    c1 = P.ops.launch_counts()
    print(json.dumps({"arm": name, "forward_agg_ms": round(out["roofline_workload_agg"]["kernel_ms"], 4),
                      "transposed_adam_ms": round(out["roofline_agg_adam"]["kernel_ms"], 4),
                      "kinds": {k_: c1[k_] - c0[k_] for k_ in c1 if k_.startswith("agg") and c1[k_] != c0[k_]}}))
P.ops.mapped_form, P.ops._agg_tune = mapped0, tune0
