"""GPU debugging aid (round 2): (1) the reference-style loop vs BaseModel's fused step on fixture G8,
step by step; (2) Hits@20 of the ddi recipe over seeds on GPU / oracle fp32 / oracle fp64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import oracle as O
import plnlp_amd as P
from gpu_util import to_graph

what = sys.argv[1] if len(sys.argv) > 1 else "loop"
if what == "loop":
    from tests.test_oracle import _toy_adj, build_trainer_from_g8
    from tests.test_hip_parity import _g8_model
    g = np.load("tests/golden/g8_train_trajectory.npz", allow_pickle=False)
    N, lo, hi, w, adj = _toy_adj(g)
    name = sys.argv[2] if len(sys.argv) > 2 else "sage_mlp_whinge_noweight"
    m, c = _g8_model(P, g, name, N)              # surface modules
    m2, _ = _g8_model(P, g, name, N)             # fused trainer
    (enc, pred, emb), _ = build_trainer_from_g8(g, name, adj, N)
    a = O.gcn_norm_csr(adj) if c["enc"] == "GCN" else adj
    ref = O.TrainerRef(enc.double(), pred.double(), emb.double(),
                       O.CSR(a.rowptr, a.col, None if a.val is None else a.val.double(), a.n_cols),
                       loss_name=c["loss"], lr=0.01, clip_norm=c["clip"])
    adj_t = to_graph(P, a)
    class D: pass
    data = D(); data.adj_t = adj_t
    params = list(m.encoder.parameters()) + list(m.predictor.parameters()) + list(m.emb.parameters())
    opt = torch.optim.Adam(params, lr=0.01)
    pos_all = torch.stack([lo, hi], 1)
    fn, weighted = P.loss.BY_NAME.get(c["loss"], (P.loss.auc_loss, False))
    m.encoder.train(); m.predictor.train(); m2.encoder.train(); m2.predictor.train()
    torch.manual_seed(4242)
    for epoch in range(2):
        neg_all = P.negative_sample.local_neg_sample(pos_all, N, c["k"])
        batches = P.utils.batch_permutation(pos_all.size(0), c["B"], True)
        for bi, perm in enumerate(batches):
            opt.zero_grad()
            h = m.encoder(m.emb.weight, adj_t)
            pe = pos_all.cuda()[perm.cuda()].t(); ne = neg_all.cuda()[perm.cuda()].reshape(-1, 2).t()
            po = m.predictor(h[pe[0]], h[pe[1]]); no = m.predictor(h[ne[0]], h[ne[1]])
            loss = P.loss.auc_loss(po, no, c["k"]) if (not weighted or True) else None
            loss.backward()
            g_surface = [p.grad.detach().clone() for p in params]
            torch.nn.utils.clip_grad_norm_(m.encoder.parameters(), c["clip"])
            torch.nn.utils.clip_grad_norm_(m.predictor.parameters(), c["clip"])
            opt.step()
            l2 = m2.train_step(data, pos_all.cuda()[perm.cuda()], neg_all.cuda()[perm.cuda()], c["k"], None)
            lr_, _, _ = ref.step(pos_all[perm], neg_all[perm], c["k"], None)
            p2 = list(m2.encoder.parameters()) + list(m2.predictor.parameters()) + list(m2.emb.parameters())
            pr = list(ref.encoder.parameters()) + list(ref.predictor.parameters()) + list(ref.emb.parameters())
            dw = max(float((a_.detach().cpu().double() - b_.detach()).abs().max()) for a_, b_ in zip(params, pr))
            dw2 = max(float((a_.detach().cpu().double() - b_.detach()).abs().max()) for a_, b_ in zip(p2, pr))
            print(f"epoch {epoch} step {bi}: loss surface {float(loss):.6f} fused {float(l2):.6f} f64 {float(lr_):.6f} | "
                  f"max|w - w64| surface {dw:.3e} fused {dw2:.3e}")
else:
    import bench
    rows = []
    for s in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
        r = bench.hits_parity(P, torch.device("cuda"), epochs=5, recipe="ddi", with_f64=True, seed=s + 1)
        rows.append((r["gpu_valid"], r["gpu_test"], r["cpu_valid"], r["cpu_test"], r["cpu64_valid"], r["cpu64_test"]))
        lo_ = r["epoch_losses"]
        print(s + 1, rows[-1], "loss gpu/cpu/cpu64 last", lo_["gpu"][-1], lo_["cpu"][-1], lo_["cpu64"][-1], flush=True)
    a = np.array(rows)
    print("means", a.mean(0), "std", a.std(0, ddof=1))
