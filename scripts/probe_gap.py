"""diagnostic: does the steady-state collab step call the device allocator, and where does the host block?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plnlp_amd as P
from plnlp_amd import synthetic

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.manual_seed(1234)
P.manual_seed(1234)
g = synthetic.make_graph("collab", seed=2, device=dev, weighted=True)
n, data = g["num_nodes"], g["data"]
B, k, K = 65536, 1, 30
gen = torch.Generator(device=dev).manual_seed(777)
sel = torch.randint(0, g["edges"].size(0), ((K + 5) * B,), generator=gen, device=dev)
pos_all = g["edges"][sel]
w_all = torch.rand(pos_all.size(0), device=dev)
neg_all = torch.randint(0, n, (pos_all.size(0), k, 2), device=dev, generator=gen)
m = P.BaseModel(lr=1e-3, dropout=0.3, grad_clip_norm=1.0, gnn_num_layers=1, mlp_num_layers=2, emb_hidden_channels=256,
                gnn_hidden_channels=256, mlp_hidden_channels=256, num_nodes=n, num_node_feats=0,
                gnn_encoder_name="SAGE", predictor_name="DOT", loss_func="WeightedHingeAUC", optimizer_name="Adam",
                device=dev, use_node_feats=False, train_node_emb=True)
m.param_init(); m.encoder.train()
plans = {}


def step(i, timing=None):
    sl = slice(i * B, (i + 1) * B)
    prep = plans.pop(i, None) or m.prepare_edges(pos_all[sl], neg_all[sl])
    nx = slice((i + 1) * B, (i + 2) * B)
    t0 = time.perf_counter()
    plans[i + 1] = m.prepare_edges(pos_all[nx], neg_all[nx])
    t1 = time.perf_counter()
    out = m.train_step(data, pos_all[sl], neg_all[sl], k, w_all[sl], edges_ready=True, prepared=prep)
    t2 = time.perf_counter()
    if timing is not None:
        timing.append((t1 - t0, t2 - t1))
    return out


for i in range(5):
    step(i)
torch.cuda.synchronize()
st0 = torch.cuda.memory_stats()
tm = []
t0 = time.perf_counter()
for i in range(5, 5 + K - 1):
    step(i, tm)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
st1 = torch.cuda.memory_stats()
print("ms/step", t_all / (K - 1) * 1e3, "host enqueue ms/step", t_host / (K - 1) * 1e3)
for key in ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams", "allocation.all.allocated"):
    print(key, st1.get(key, 0) - st0.get(key, 0))
import statistics
print("host: prepare_edges ms", statistics.median(a for a, _ in tm) * 1e3, "train_step ms", statistics.median(b for _, b in tm) * 1e3)
print("EdgeBatch.join:", P.ops.JOIN_STATS)
