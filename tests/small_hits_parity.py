"""The small same-seed Hits@K comparison of rounds 1-5's bench line (a 3 000-node collab-shaped graph at h = 64, trained on the HIP
path and on the CPU oracle side by side): kept for the two tests that use it.  bench.py's `hits50_parity` now runs the collab
recipe at its own width against the oracle's fixture curves instead.  Test infrastructure: imports the oracle."""
import torch


def hits_parity(P, device, epochs=12, recipe="collab", with_f64=False, seed=0):
    """Hits@K parity (BASELINE.json metric): train the SAME small problem on the GPU path and on the CPU
    oracle -- same initial weights, same negatives, same batch permutations, dropout 0 so both are
    deterministic -- and compare Hits@K on held-out edges after every epoch.
    recipe 'collab': SAGE x1 + DOT, WeightedHingeAUC, k=1, Hits@50 (the bench workload's recipe);
    recipe 'ddi'   : SAGE x2 + MLP predictor, AUC loss, k=3, Hits@20 (BASELINE config 2's recipe).
    seed: another initialisation and another stream of negatives / batch permutations (same graph and
    held-out edges) -- the reference reports mean +- std over 10 such runs (main.py:43)."""
    import oracle as O
    from plnlp_amd import synthetic
    from plnlp_amd.utils import Evaluator, evaluate_hits
    ddi = recipe == "ddi"
    g = synthetic.make_graph("collab", seed=11, device="cpu", num_nodes=3000, num_edges=24000, weighted=True)
    n, h, B, k = g["num_nodes"], 64, 4096, (3 if ddi else 1)
    layers, pred_name, loss_name, hk = (2, "MLP", "AUC", "Hits@20") if ddi else (1, "DOT", "WeightedHingeAUC", "Hits@50")
    edges, w = g["edges"], g["weight"] / 5.0
    gen = torch.Generator().manual_seed(5)
    perm = torch.randperm(edges.size(0), generator=gen)
    held, train = edges[perm[:2000]], edges[perm[2000:]]
    wtrain = w[perm[2000:]]
    negs = torch.randint(0, n, (20000, 2), generator=gen)
    adj = P.Graph.from_coo(torch.cat([train[:, 0], train[:, 1]]), torch.cat([train[:, 1], train[:, 0]]), None, n, n)
    model = P.BaseModel(lr=0.01, dropout=0.0, grad_clip_norm=1.0, gnn_num_layers=layers, mlp_num_layers=2,
                        emb_hidden_channels=h, gnn_hidden_channels=h, mlp_hidden_channels=h, num_nodes=n,
                        num_node_feats=0, gnn_encoder_name="SAGE", predictor_name=pred_name,
                        loss_func=loss_name, optimizer_name="Adam", device=device,
                        use_node_feats=False, train_node_emb=True)
    torch.manual_seed(21 + 7919 * seed)
    model.param_init()
    enc = O.GNNRef("SAGE", h, h, h, layers, 0.0)
    enc.load_state_dict({k_: v.cpu() for k_, v in model.encoder.state_dict().items()})
    if ddi:
        pred = O.MLPPredictorRef(h, h, 1, 2, 0.0)
        pred.load_state_dict({k_: v.cpu() for k_, v in model.predictor.state_dict().items()})
    else:
        pred = O.DotPredictorRef()
    emb = torch.nn.Embedding(n, h)
    emb.weight.data.copy_(model.emb.weight.detach().cpu())
    csr = O.CSR(adj.rowptr, adj.col.to(torch.int64), None, n)
    ref = O.TrainerRef(enc, pred, emb, csr, loss_name=loss_name, lr=0.01, clip_norm=1.0)
    ref64 = None
    if with_f64:        # the same oracle in float64: |fp32 - fp64| of the REFERENCE arithmetic is the yardstick
        import copy
        ref64 = O.TrainerRef(copy.deepcopy(enc).double(), copy.deepcopy(pred).double(), copy.deepcopy(emb).double(),
                             O.CSR(adj.rowptr, adj.col.to(torch.int64), None, n), loss_name=loss_name, lr=0.01,
                             clip_norm=1.0)

    class D:
        pass
    data = D()
    data.adj_t = adj.to(device)
    data.edge_index = torch.stack([adj.coo()[1], adj.coo()[0]])
    tr_split = {"edge": train} if ddi else {"edge": train, "weight": wtrain}
    split = {"train": tr_split,
             "valid": {"edge": held[:1000], "edge_neg": negs[:10000]},
             "test": {"edge": held[1000:], "edge_neg": negs[10000:]}}
    ev = Evaluator("ogbl-ddi" if ddi else "ogbl-collab")
    rows = []
    losses = {"gpu": [], "cpu": [], "cpu64": []}
    for epoch in range(epochs):
        torch.manual_seed(1000 + epoch + 100003 * seed)
        losses["gpu"].append(float(model.train(data, split, B, "local", k)))
        torch.manual_seed(1000 + epoch + 100003 * seed)
        _, neg = O.pos_neg_edges_ref("train", {"train": {"edge": train}}, num_nodes=n, neg_sampler_name="local",
                                     num_neg=k)
        losses["cpu"].append(float(ref.train_epoch(train, neg, B, k, None if ddi else wtrain)))
        res = model.test(data, split, B, ev, "hits")[hk]

        def oracle_hits(t):
            hh = t.embed_for_eval()
            pv = [t.score(hh, split[s_]["edge"], B) for s_ in ("valid", "test")]
            nv = [t.score(hh, split[s_]["edge_neg"], B) for s_ in ("valid", "test")]
            return O.evaluate_hits_ref(pv[0], nv[0], pv[1], nv[1])[hk]
        rres = oracle_hits(ref)
        row = [100 * res[0], 100 * res[1], 100 * rres[0], 100 * rres[1]]
        if ref64 is not None:
            torch.manual_seed(1000 + epoch + 100003 * seed)          # same negatives, same permutation stream
            O.pos_neg_edges_ref("train", {"train": {"edge": train}}, num_nodes=n, neg_sampler_name="local", num_neg=k)
            losses["cpu64"].append(float(ref64.train_epoch(train, neg, B, k, None if ddi else wtrain.double())))
            r64 = oracle_hits(ref64)
            row += [100 * r64[0], 100 * r64[1]]
        rows.append(tuple(row))
    last = rows[-1]
    extra = {}
    if ref64 is not None:
        extra = {"cpu64_valid": last[4], "cpu64_test": last[5], "epoch_losses": losses,
                 "gpu_vs_f64_points": max(max(abs(r[0] - r[4]), abs(r[1] - r[5])) for r in rows),
                 "cpu32_vs_f64_points": max(max(abs(r[2] - r[4]), abs(r[3] - r[5])) for r in rows)}
    return {"epochs": epochs, "metric": hk, "gpu_valid": last[0], "gpu_test": last[1], "cpu_valid": last[2],
            "cpu_test": last[3], **extra,
            "max_abs_diff_points": max(max(abs(r[0] - r[2]), abs(r[1] - r[3])) for r in rows),
            "note": "%s (percent) on 1000+1000 held-out edges vs 10000 negatives each, small collab-shaped "
                    "graph (N=3000), SAGE x%d h=64 + %s, same seeds on the HIP path and the CPU oracle"
                    % (hk, layers, pred_name)}
