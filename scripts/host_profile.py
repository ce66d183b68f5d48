"""Where the host's ~0.9 ms per collab step goes: cProfile over the loop BaseModel.train / bench.py run
(StepPipeline.prepare one batch ahead + StepPipeline.step), eager.  Diagnostic."""
import cProfile, os, pstats, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plnlp_amd as P
from plnlp_amd import synthetic

dev = torch.device("cuda", 0)
g = synthetic.make_graph("collab", seed=2, device=dev, weighted=True)
n, B, k = g["num_nodes"], 65536, 1
gen = torch.Generator(device=dev).manual_seed(1)
S = 70
pos = torch.randint(0, n, (S * B, 2), device=dev, generator=gen)
neg = torch.randint(0, n, (S * B, k, 2), device=dev, generator=gen)
w = torch.rand(S * B, device=dev, generator=gen)
m = P.BaseModel(lr=1e-3, dropout=0.3, grad_clip_norm=1.0, gnn_num_layers=1, mlp_num_layers=2, emb_hidden_channels=256,
                gnn_hidden_channels=256, mlp_hidden_channels=256, num_nodes=n, num_node_feats=0, gnn_encoder_name="SAGE",
                predictor_name="DOT", loss_func="WeightedHingeAUC", optimizer_name="Adam", device=dev, use_node_feats=False,
                train_node_emb=True)
m.param_init(); m.encoder.train()
pipe = m.pipeline(g["data"], k, B, True, capture=False)


def prep(i):
    sl = slice(i * B, (i + 1) * B)
    return pipe.prepare(pos[sl], neg[sl], w[sl])


def loop(lo, hi):
    h = prep(lo)
    for i in range(lo, hi):
        nxt = prep(i + 1) if i + 1 < hi else None
        pipe.step(h, global_count=B)
        h = nxt


loop(0, 8)
torch.cuda.synchronize()
w0 = P.ops.StepThrottle.waited_s
t0 = time.perf_counter()
loop(8, 38)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("plain: enqueue %.3f ms/step, of which waiting for the GPU %.3f" % ((t1 - t0) / 30 * 1e3, (P.ops.StepThrottle.waited_s - w0) / 30 * 1e3))
pr = cProfile.Profile()
pr.enable()
loop(38, 68)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(30)
