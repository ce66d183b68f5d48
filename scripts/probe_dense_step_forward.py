"""the ddi model's encoder output (2 x SAGE, h = 512) per forced slice count of the dense aggregation against the CSR kernels:
element-wise and column-coherent differences (what a bias gradient downstream sums)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plnlp_amd as P
from plnlp_amd import synthetic, _lib
dev = torch.device("cuda")
g = synthetic.make_graph("ddi", seed=2, device="cpu")
n, h = g["num_nodes"], 512
torch.manual_seed(31)
m = P.BaseModel(lr=1e-3, dropout=0.0, grad_clip_norm=2.0, gnn_num_layers=2, mlp_num_layers=2, emb_hidden_channels=h,
                gnn_hidden_channels=h, mlp_hidden_channels=h, num_nodes=n, num_node_feats=0, gnn_encoder_name="SAGE",
                predictor_name="MLP", loss_func="AUC", optimizer_name="Adam", device="cuda", use_node_feats=False, train_node_emb=True)
m.param_init()


class Data:
    pass


data = Data()
data.adj_t = g["adj_t"].to("cuda")
m.encoder.train()
lib = _lib.load()
res = {}
with torch.no_grad():
    P.ops.DENSE_AGG["enabled"] = False
    ref = m.encoder(m.create_input_feat(data), data.adj_t).double()
    P.ops.DENSE_AGG["enabled"] = True
    for s in (4, 3, 5, 7, 4):
        lib.plnlp_dense_aggregate_tuning(s)
        c0 = P.ops.launch_counts()
        hh = m.encoder(m.create_input_feat(data), data.adj_t).double()
        c1 = P.ops.launch_counts()
        d = hh - ref
        print(json.dumps({"slices": s, "dense_launches": c1["agg_dense"] - c0["agg_dense"], "max_abs_diff_over_max": float(d.abs().max() / ref.abs().max()),
                          "col_coherent_max": float((d.sum(0).abs() / ref.abs().sum(0)).max()),
                          "mean_signed": float(d.mean() / ref.abs().mean()), "nan": bool(torch.isnan(hh).any())}), flush=True)
lib.plnlp_dense_aggregate_tuning(0)
