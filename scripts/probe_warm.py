"""diagnostic: per-step wall time and device-allocator calls of the first collab steps"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plnlp_amd as P
from plnlp_amd import synthetic
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
torch.manual_seed(1234); P.manual_seed(1234)
g = synthetic.make_graph("collab", seed=2, device=dev, weighted=True)
n, data = g["num_nodes"], g["data"]
B, k, K = 65536, 1, 30
gen = torch.Generator(device=dev).manual_seed(777)
sel = torch.randint(0, g["edges"].size(0), ((K + 2) * B,), generator=gen, device=dev)
pos_all = g["edges"][sel]; w_all = torch.rand(pos_all.size(0), device=dev)
neg_all = torch.randint(0, n, (pos_all.size(0), k, 2), device=dev, generator=gen)
m = P.BaseModel(lr=1e-3, dropout=0.3, grad_clip_norm=1.0, gnn_num_layers=1, mlp_num_layers=2, emb_hidden_channels=256,
                gnn_hidden_channels=256, mlp_hidden_channels=256, num_nodes=n, num_node_feats=0,
                gnn_encoder_name="SAGE", predictor_name="DOT", loss_func="WeightedHingeAUC", optimizer_name="Adam",
                device=dev, use_node_feats=False, train_node_emb=True)
m.param_init(); m.encoder.train()
plans = {}
def step(i):
    sl = slice(i * B, (i + 1) * B)
    prep = plans.pop(i, None) or m.prepare_edges(pos_all[sl], neg_all[sl])
    nx = slice((i + 1) * B, (i + 2) * B)
    plans[i + 1] = m.prepare_edges(pos_all[nx], neg_all[nx])
    return m.train_step(data, pos_all[sl], neg_all[sl], k, w_all[sl], edges_ready=True, prepared=prep)
torch.cuda.synchronize()
prev = torch.cuda.memory_stats().get("num_device_alloc", 0)
for i in range(K):
    t0 = time.perf_counter(); step(i); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    cur = torch.cuda.memory_stats().get("num_device_alloc", 0)
    print(i, round(dt * 1e3, 3), "ms (synchronised)", "hipMalloc", cur - prev, "reserved MB", torch.cuda.memory_reserved() >> 20)
    prev = cur
