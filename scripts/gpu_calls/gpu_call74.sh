#!/bin/bash
# long run: 400 steps -- step time and allocator state must stay flat
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import plnlp_amd as P
from plnlp_amd import synthetic
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
torch.manual_seed(1); P.manual_seed(1)
g = synthetic.make_graph("collab", seed=2, device=dev, weighted=True)
n, data = g["num_nodes"], g["data"]
m = P.BaseModel(lr=1e-3, dropout=0.3, grad_clip_norm=1.0, gnn_num_layers=1, mlp_num_layers=2, emb_hidden_channels=256,
                gnn_hidden_channels=256, mlp_hidden_channels=256, num_nodes=n, num_node_feats=0,
                gnn_encoder_name="SAGE", predictor_name="DOT", loss_func="WeightedHingeAUC", optimizer_name="Adam",
                device=dev, use_node_feats=False, train_node_emb=True)
m.param_init()
split = {"train": {"edge": g["edges"], "weight": g["weight"] / 5.0}}
for ep in range(22):      # 18 batches per epoch at B = 65536 -> ~400 steps
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = m.train(data, split, 65536, "global", 1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = torch.cuda.memory_stats()
    if ep % 3 == 0 or ep == 21:
        print(f"epoch {ep}: loss {loss:.4f}  {dt*1e3:.1f} ms  reserved {torch.cuda.memory_reserved()>>20} MB  hipMalloc calls {st.get('num_device_alloc',0)}", flush=True)
PY
