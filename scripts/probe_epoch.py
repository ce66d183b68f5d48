import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import plnlp_amd as P
from plnlp_amd import synthetic, model as M, utils as U
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
orig = M.get_pos_neg_edges
acc = {}
def timed(*a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = orig(*a, **k)
    torch.cuda.synchronize(); acc["sampler"] = time.perf_counter() - t0
    return out
M.get_pos_neg_edges = timed
orig_bp = M.batch_permutation
def timed_bp(*a, **k):
    t0 = time.perf_counter(); out = orig_bp(*a, **k); acc["perm"] = time.perf_counter() - t0; return out
M.batch_permutation = timed_bp
orig_ts = m.train_step
marks = []
def timed_ts(*a, **k):
    marks.append(time.perf_counter()); out = orig_ts(*a, **k); marks.append(time.perf_counter()); return out
m.train_step = timed_ts
import gc
if os.environ.get("PROBE_GC_OFF") == "1":
    gc.disable()
gc_t = []
def _cb(phase, info):
    if phase == "start": gc_t.append([time.perf_counter(), info["generation"]])
    else: gc_t[-1].append(time.perf_counter())
gc.callbacks.append(_cb)
for ep in range(12):
    marks.clear(); gc_t.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = m.train(data, split, 65536, "global", 1)
    t_ret = time.perf_counter()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    steps = [(marks[i + 1] - marks[i]) * 1e3 for i in range(0, len(marks), 2)]
    gaps = [(marks[i + 2] - marks[i + 1]) * 1e3 for i in range(0, len(marks) - 2, 2)]
    print(f"epoch {ep}: {dt*1e3:.1f} ms | sampler {acc['sampler']*1e3:.1f} perm {acc['perm']*1e3:.1f} | before first step "
          f"{(marks[0]-t0)*1e3:.1f} | in train_step calls {sum(steps):.1f} (max {max(steps):.1f}) | between calls {sum(gaps):.1f} "
          f"(max {max(gaps):.1f}) | after last step until return {(t_ret-marks[-1])*1e3:.1f} | gc: "
          + ", ".join(f"gen{g[1]} {(g[2]-g[0])*1e3:.1f}ms" for g in gc_t if len(g) == 3 and g[2]-g[0] > 1e-3), flush=True)
