"""Host-side cost of a training step (enqueue time, time blocked on the count read-back, how far the
host runs ahead of the GPU), single process vs a 1-rank RCCL group.  Diagnostic, not a benchmark."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", 0))
import plnlp_amd as P
from plnlp_amd import synthetic
dev = torch.device("cuda", 0)
g = synthetic.make_graph("collab", seed=2, device=dev, weighted=True)
n = g["num_nodes"]; B = 65536
gen = torch.Generator(device=dev).manual_seed(1)
pos = torch.randint(0, n, (40 * B, 2), device=dev, generator=gen)
neg = torch.randint(0, n, (40 * B, 1, 2), device=dev, generator=gen)
w = torch.rand(40 * B, device=dev, generator=gen)
for dist in (False, True):
    m = P.BaseModel(lr=1e-3, dropout=0.3, grad_clip_norm=1.0, gnn_num_layers=1, mlp_num_layers=2, emb_hidden_channels=256,
                    gnn_hidden_channels=256, mlp_hidden_channels=256, num_nodes=n, num_node_feats=0, gnn_encoder_name="SAGE",
                    predictor_name="DOT", loss_func="WeightedHingeAUC", optimizer_name="Adam", device=dev, use_node_feats=False,
                    train_node_emb=True, process_group=torch.distributed.group.WORLD if dist else None)
    m.param_init(); m.encoder.train()
    fn = m.train_step_global if dist else m.train_step
    for i in range(5):
        sl = slice(i * B, (i + 1) * B); fn(g["data"], pos[sl], neg[sl], 1, w[sl], edges_ready=True)
    torch.cuda.synchronize()
    import cProfile, pstats
    pr = cProfile.Profile()
    t0 = time.perf_counter(); pr.enable()
    for i in range(5, 35):
        sl = slice(i * B, (i + 1) * B); fn(g["data"], pos[sl], neg[sl], 1, w[sl], edges_ready=True)
    pr.disable(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("dist" if dist else "single", "host enqueue ms/step", (t1 - t0) / 30 * 1e3, "drain ms", (t2 - t1) * 1e3)
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(8)
torch.distributed.destroy_process_group()
