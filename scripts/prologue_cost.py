"""What the next batch's index preparation costs a training step: the preparation alone on an idle GPU, the step with
every batch prepared BEFORE the timed region (no side-stream work at all), and the step with the usual one batch of
look-ahead.  usage: prologue_cost.py [workload] [steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import plnlp_amd as P
from plnlp_amd import synthetic

name = sys.argv[1] if len(sys.argv) > 1 else "collab"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = bench.WORKLOADS[name]
dev = torch.device("cuda")
torch.manual_seed(1); P.manual_seed(1)
g = synthetic.make_graph(cfg["shape"], seed=2, device=dev, weighted=cfg["weighted"])
n, data = g["num_nodes"], g["data"]
if cfg["encoder"] == "GCN":
    g["adj_t"] = data.adj_t = P.gcn_normalization(g["adj_t"])
feats = cfg.get("feats", 0)
if feats:
    data.x = torch.randn(n, feats, device=dev)
m = P.BaseModel(lr=1e-3, dropout=cfg["dropout"], grad_clip_norm=cfg["clip"], gnn_num_layers=cfg["gnn_layers"],
                mlp_num_layers=cfg["mlp_layers"], emb_hidden_channels=cfg.get("emb", cfg["hidden"]),
                gnn_hidden_channels=cfg["hidden"], mlp_hidden_channels=cfg["hidden"], num_nodes=n, num_node_feats=feats,
                gnn_encoder_name=cfg["encoder"], predictor_name=cfg["predictor"], loss_func=cfg["loss"],
                optimizer_name="Adam", device=dev, use_node_feats=feats > 0, train_node_emb=True)
m.param_init(); m.encoder.train(); m.predictor.train()
B, k = cfg["batch"], cfg["num_neg"]
W = 8
gen = torch.Generator(device=dev).manual_seed(7)
pos = g["edges"][torch.randint(0, g["edges"].size(0), ((W + K) * B,), generator=gen, device=dev)]
neg = torch.randint(0, n, ((W + K) * B, k, 2), generator=gen, device=dev)
w = torch.rand((W + K) * B, device=dev) if cfg["weighted"] else None
sl = lambda i: slice(i * B, (i + 1) * B)

def steps(lo, hi, prepared=None):
    plans = {}
    for i in range(lo, hi):
        if prepared is not None:
            prep = prepared[i]
        else:
            prep = plans.pop(i, None) or m.prepare_edges(pos[sl(i)], neg[sl(i)])
            if i + 1 < hi:
                plans[i + 1] = m.prepare_edges(pos[sl(i + 1)], neg[sl(i + 1)])
        m.train_step(data, pos[sl(i)], neg[sl(i)], k, None if w is None else w[sl(i)], edges_ready=True, prepared=prep)

def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3

steps(0, W)
out = {"workload": name, "steps": K}
for rep in range(2):
    out[f"look_ahead_ms_{rep}"] = timed(lambda: steps(W, W + K)) / K
    prepared = {i: m.prepare_edges(pos[sl(i)], neg[sl(i)]) for i in range(W, W + K)}
    for b in prepared.values():
        _ = b.incidence.count if b.incidence is not None and hasattr(b.incidence, "count") else None
    out[f"prepared_before_ms_{rep}"] = timed(lambda: steps(W, W + K, prepared)) / K
    del prepared
    # the preparation alone, back to back on an idle GPU (side stream, as in the step)
    def preps():
        for i in range(W, W + K):
            b = m.prepare_edges(pos[sl(i)], neg[sl(i)])
        P.ops.side_stream(dev).synchronize()
    out[f"preparation_alone_ms_{rep}"] = timed(preps) / K
print(json.dumps(out))
