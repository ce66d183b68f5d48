"""GPU parity, second batch: max aggregation, the on-device `global` sampler, MRR on the device path,
a reference-style training loop written from the module surface only, statistical Hits@K parity of
the ddi recipe, and the driver's graph preparation against an oracle restatement.  Same rules as
tests/test_hip_parity.py: through the C ABI, fp32 tolerance 1e-5 relative, integer outputs bit-exact."""
import numpy as np
import pytest
import torch

import oracle as O
from gpu_util import close, dev, rand_csr, to_graph

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import plnlp_amd
    from plnlp_amd import _lib
    _lib.load()                      # no library -> the GPU suite must fail, not skip
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return plnlp_amd


@pytest.fixture(params=["f32", "bf16x3"])
def math(request, P):
    """both ways the dense products are formed (include/plnlp_hip.h PLNLP_GEMM_MATH_*)"""
    old = P.ops.GEMM_MATH["mode"]
    P.ops.GEMM_MATH["mode"] = request.param
    yield request.param
    P.ops.GEMM_MATH["mode"] = old


# ------------------------------------------------------------ max aggregation ----
@pytest.mark.parametrize("feat", [4, 32, 64, 100, 128, 200, 256, 512, 1024, 178, 7])
@pytest.mark.parametrize("weighted", [False, True])
def test_csr_aggregate_max_matches_oracle(P, feat, weighted):
    """values AND arg positions bit-exact (max of fp32 products is exact), incl. a hub row that takes
    the chunk / finalize passes, an empty row, and ties (small-integer features): first entry wins"""
    csr = rand_csr(300, 3000, feat + 17, weighted=weighted, hub=700)
    g = to_graph(P, csr)
    gen = torch.Generator().manual_seed(2)
    for x in (torch.randn(300, feat, generator=gen),
              torch.randint(-2, 3, (300, feat), generator=gen).float()):       # many ties
        ref, ref_arg = O.spmm_max(csr if weighted else O.CSR(csr.rowptr, csr.col, None, csr.n_cols), x, True)
        out, arg = P.ops.csr_aggregate_max(g, dev(x), use_values=weighted)
        assert torch.equal(out.cpu(), ref), f"feat={feat}"
        assert torch.equal(arg.cpu().long(), ref_arg), f"feat={feat}"
        out2, arg2 = P.ops.csr_aggregate_max(g, dev(x), use_values=weighted, split=None)   # hub row on one wave
        assert torch.equal(out2, out) and torch.equal(arg2, arg)


@pytest.mark.parametrize("feat", [64, 256, 178])
@pytest.mark.parametrize("weighted", [False, True])
def test_csr_aggregate_max_backward_matches_oracle(P, feat, weighted):
    csr = rand_csr(200, 2500, feat, weighted=weighted, hub=300)
    ocsr = csr if weighted else O.CSR(csr.rowptr, csr.col, None, csr.n_cols)
    g = to_graph(P, csr)
    gen = torch.Generator().manual_seed(3)
    x = torch.randint(-2, 3, (200, feat), generator=gen).float() + 0.25 * torch.randn(200, feat, generator=gen).round()
    gy = torch.randn(200, feat, generator=gen)
    xr = x.double().requires_grad_(True)
    O.spmm(ocsr, xr, "max", True).backward(gy.double())
    xd = dev(x).requires_grad_(True)
    out = P.ops.AggregateFn.apply(xd, g, "max", weighted)
    out.backward(dev(gy))
    close(xd.grad, xr.grad)
    xd2 = dev(x).requires_grad_(True)                       # deterministic: same bits on a second run
    P.ops.AggregateFn.apply(xd2, g, "max", weighted).backward(dev(gy))
    assert torch.equal(xd.grad, xd2.grad)


def test_sage_conv_max_and_add_reductions(P):
    """SAGEConv(aggr=...) over the max / sum kernels vs lin_l(reduce(x)) + lin_r(x) in float64"""
    csr = rand_csr(150, 1200, 9, weighted=False)
    g = to_graph(P, csr)
    x = torch.randn(150, 64, generator=torch.Generator().manual_seed(4))
    for aggr, red in (("max", "max"), ("add", "sum")):
        torch.manual_seed(5)
        conv = P.SAGEConv(64, 32, aggr=aggr).cuda()
        xd = dev(x).requires_grad_(True)
        y = conv(xd, g)
        y.square().sum().backward()
        xr = x.double().requires_grad_(True)
        wl, bl, wr = (t.detach().cpu().double().requires_grad_(True) for t in
                      (conv.lin_l.weight, conv.lin_l.bias, conv.lin_r.weight))
        yr = O.spmm(csr, xr, red, False) @ wl.t() + bl + xr @ wr.t()
        yr.square().sum().backward()
        close(y, yr, rtol=2e-5)
        close(xd.grad, xr.grad, rtol=5e-5)
        close(conv.lin_l.weight.grad, wl.grad, rtol=5e-5)
        close(conv.lin_r.weight.grad, wr.grad, rtol=5e-5)
        close(conv.lin_l.bias.grad, bl.grad, rtol=5e-5)


# ------------------------------------------------- `global` sampler on the device ----
def _sym_edge_index(n, e, seed):
    g = torch.Generator().manual_seed(seed)
    a, b = torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)
    keep = a != b
    key = torch.unique(torch.minimum(a, b)[keep] * n + torch.maximum(a, b)[keep])
    lo, hi = key // n, key % n
    return torch.stack([torch.cat([lo, hi]), torch.cat([hi, lo])])


def test_global_sampler_on_device_contract(P):
    """negative_sample.py:6-20 with the structured sampler running on a CUDA edge list (what
    BaseModel.train does for ddi / collab): right shape and device, no existing edge, no self loop, no
    duplicate, and the SAME tensor for the same CPU seed on a second call (ranks must agree)."""
    from plnlp_amd import negative_sample as NS
    n, k = 700, 3
    ei = _sym_edge_index(n, 6000, 1).cuda()
    e = ei.size(1) // 2
    torch.manual_seed(77)
    out = NS.global_neg_sample(ei, n, e, k)
    torch.manual_seed(77)
    again = NS.global_neg_sample(ei, n, e, k)
    assert out.is_cuda and out.dtype == torch.int64 and out.shape == (e, k, 2)
    assert torch.equal(out, again)
    torch.manual_seed(78)
    assert not torch.equal(out, NS.global_neg_sample(ei, n, e, k))
    flat = out.reshape(-1, 2).cpu()
    keys = flat[:, 0] * n + flat[:, 1]
    assert bool((flat[:, 0] != flat[:, 1]).all())
    assert not set((ei[0] * n + ei[1]).cpu().tolist()).intersection(keys.tolist())
    assert torch.unique(keys).numel() == keys.numel()        # sparse graph: nothing to pad, so no repeats
    assert 0 <= int(flat.min()) and int(flat.max()) < n
    counts = np.bincount(flat[:, 0].numpy(), minlength=n)     # roughly uniform over the free cells
    assert counts.std() / counts.mean() < 0.45


def test_global_sampler_padding_branch_on_device(P, monkeypatch):
    """a dense graph leaves fewer free cells than negatives asked for: the reference tops the sample
    up with randperm-picked repeats (negative_sample.py:11-18) -- same picks as the oracle's
    restatement from the same CPU seed, on device tensors"""
    from plnlp_amd import negative_sample as NS
    n, k = 9, 4
    full = torch.ones(n, n).nonzero().t()
    ei = full[:, (full[0] + full[1]) % 4 != 0].cuda()         # 3/4 of all cells are edges
    e = 6                                                     # 24 wanted, 16 free cells: padded with 8 repeats
    torch.manual_seed(5)
    out = NS.global_neg_sample(ei, n, e, k)                   # natural short sample: only checks the contract
    assert out.shape == (e, k, 2) and out.is_cuda
    flat = out.reshape(-1, 2).cpu()
    assert bool(((flat[:, 0] + flat[:, 1]) % 4 == 0).all()) and bool((flat[:, 0] != flat[:, 1]).all())
    assert torch.unique(flat[:, 0] * n + flat[:, 1]).numel() < flat.size(0)      # it did have to pad
    short = torch.tensor([[0, 1, 2, 3, 5, 6, 7], [4, 3, 6, 1, 3, 2, 1]]).cuda()
    monkeypatch.setattr(NS, "structured_negative_sampling", lambda *a, **kw: short)
    torch.manual_seed(23)
    got = NS.global_neg_sample(ei, n, 3, 4)
    torch.manual_seed(23)
    src, dst = O.pad_negatives_ref(short.cpu(), 12)
    assert torch.equal(got.cpu(), torch.stack((src, dst), dim=-1).reshape(-1, 4, 2))


def test_train_epoch_with_global_sampler_on_device(P):
    """the ddi / collab recipes' default sampler inside BaseModel.train on the GPU: one epoch runs,
    the loss is finite, and the same seeds give the same loss bits"""
    n, h = 600, 32
    ei = _sym_edge_index(n, 5000, 2)
    adj = P.Graph.from_edge_index(ei, None, n).to("cuda")

    class D:
        pass
    data = D()
    data.adj_t, data.edge_index = adj, ei
    half = ei.size(1) // 2
    split = {"train": {"edge": ei[:, :half].t().contiguous()}}
    losses = []
    for _ in range(2):
        torch.manual_seed(9)
        P.manual_seed(9)
        m = P.BaseModel(lr=0.01, dropout=0.0, grad_clip_norm=1.0, gnn_num_layers=2, mlp_num_layers=2,
                        emb_hidden_channels=h, gnn_hidden_channels=h, mlp_hidden_channels=h, num_nodes=n,
                        num_node_feats=0, gnn_encoder_name="SAGE", predictor_name="MLP", loss_func="AUC",
                        optimizer_name="Adam", device="cuda", use_node_feats=False, train_node_emb=True)
        m.param_init()
        losses.append([m.train(data, split, 1024, "global", 3) for _ in range(2)])
    assert np.isfinite(losses[0]).all() and losses[0] == losses[1]


# ------------------------------------------------------------ MRR on the device ----
@pytest.mark.parametrize("predictor", ["MLP", "DOT"])
def test_eval_path_mrr_parity_including_ties(P, predictor):
    """model.test(..., 'mrr') (model.py:184-226 + utils.py:63-80, citation2's metric) vs
    oracle.evaluate_mrr_ref on the same weights.  With the DOT scorer some negatives repeat the positive
    target, so their scores tie with the positive's bit for bit and the rank depends on the tie rule
    (the MLP case has no engineered ties: a CPU sgemm need not give the same row the same bits in two
    batches, which would turn an intended tie into a coin flip on the ORACLE side)."""
    torch.manual_seed(6)
    N, h, S, M = 500, 64, 120, 40
    ties = predictor == "DOT"
    csr = rand_csr(N, 6000, 41, weighted=False)
    m = P.BaseModel(lr=0.01, dropout=0.0, grad_clip_norm=1.0, gnn_num_layers=2, mlp_num_layers=2,
                    emb_hidden_channels=h, gnn_hidden_channels=h, mlp_hidden_channels=h, num_nodes=N,
                    num_node_feats=0, gnn_encoder_name="SAGE", predictor_name=predictor, loss_func="AUC",
                    optimizer_name="Adam", device="cuda", use_node_feats=False, train_node_emb=True)
    m.param_init()
    g = torch.Generator().manual_seed(8)
    split = {"train": {"source_node": torch.randint(0, N, (50,), generator=g),
                       "target_node": torch.randint(0, N, (50,), generator=g)}}
    for s in ("valid", "test"):
        src, dst = torch.randint(0, N, (S,), generator=g), torch.randint(0, N, (S,), generator=g)
        neg = torch.randint(0, N, (S, M), generator=g)
        if ties:
            neg[::3, 5] = dst[::3]                              # tie with the positive
            neg[::7, 9] = neg[::7, 8]                           # tie between negatives
        split[s] = {"source_node": src, "target_node": dst, "target_node_neg": neg}

    class D:
        pass
    data = D()
    data.adj_t = to_graph(P, csr)
    res = m.test(data, split, 1000, P.utils.Evaluator("ogbl-citation2"), "mrr")
    enc = O.GNNRef("SAGE", h, h, h, 2, 0.0)
    enc.load_state_dict({k: v.cpu() for k, v in m.encoder.state_dict().items()})
    if predictor == "MLP":
        pred = O.MLPPredictorRef(h, h, 1, 2, 0.0)
        pred.load_state_dict({k: v.cpu() for k, v in m.predictor.state_dict().items()})
    else:
        pred = O.DotPredictorRef()
    emb = torch.nn.Embedding(N, h)
    emb.weight.data.copy_(m.emb.weight.detach().cpu())
    tr = O.TrainerRef(enc, pred, emb, csr)
    hh = tr.embed_for_eval()
    sc = {}
    for s in ("valid", "test"):
        pos, neg = O.pos_neg_edges_ref(s, split)
        sc[s] = (tr.score(hh, pos, 1000), tr.score(hh, neg, 1000))
    ref = O.evaluate_mrr_ref(sc["valid"][0], sc["valid"][1], sc["test"][0], sc["test"][1])
    # per-edge ranks can only differ where two DIFFERENT edges score within fp32 round-off of each other
    assert abs(res["MRR"][0] - ref["MRR"][0]) <= 2e-3 and abs(res["MRR"][1] - ref["MRR"][1]) <= 2e-3, (res, ref)
    if not ties:
        return
    # the tie rule itself on the DEVICE scores: ranking on the device == ranking of the same numbers on
    # the host == the oracle's, with the positive ahead of the negatives it ties with
    with torch.no_grad():
        m.encoder.eval()
        hd = m.encoder(m.create_input_feat(data), data.adj_t)
        hd = torch.cat([hd, hd.mean(dim=0, keepdim=True)], dim=0)
        pos_e, neg_e = P.utils.get_pos_neg_edges("valid", split)
        pos_d = m.batch_predict(hd, pos_e.cuda(), 1000, to_cpu=False)
        neg_d = m.batch_predict(hd, neg_e.cuda(), 1000, to_cpu=False).view(S, M)
    assert bool((pos_d[::3] == neg_d[::3, 5]).all()) and bool((neg_d[::7, 9] == neg_d[::7, 8]).all())   # real ties
    ev = P.utils.Evaluator("ogbl-citation2")
    on_dev = ev.eval({"y_pred_pos": pos_d, "y_pred_neg": neg_d})["mrr_list"].cpu()
    on_host = ev.eval({"y_pred_pos": pos_d.cpu(), "y_pred_neg": neg_d.cpu()})["mrr_list"]
    assert torch.equal(on_dev, on_host) and torch.equal(on_host, O.mrr_list(pos_d.cpu(), neg_d.cpu()))
    strictly_above = (neg_d[::3] > pos_d[::3, None]).sum(dim=1).cpu()
    assert torch.equal(on_dev[::3], 1.0 / (strictly_above + 1).float())      # ties do not push the positive down


# ---------------------------------- the reference's loop, from the surface only ----
def test_reference_style_loop_over_the_module_surface(P, golden, math):
    """a caller that keeps the reference's own loop (model.py:147-171) -- torch
    DataLoader, `h[idx]` indexing, two predictor() calls, the loss on the two score tensors,
    torch.nn.utils.clip_grad_norm_ per module, torch.optim.Adam -- over plnlp_amd modules on the GPU,
    against the trajectory the reference itself produced (fixture G8).  Nothing of
    plnlp_amd.BaseModel's fused step is used."""
    from torch.utils.data import DataLoader
    from tests.test_oracle import _toy_adj
    from tests.test_hip_parity import _g8_model, _oracle_f64_losses
    g = golden("g8_train_trajectory")
    N, lo, hi, w, adj = _toy_adj(g)
    for name in g["config_names"].tolist():
        m, c = _g8_model(P, g, name, N)                         # only used to build + load the modules
        encoder, predictor, emb = m.encoder, m.predictor, m.emb
        adj_t = to_graph(P, O.gcn_norm_csr(adj) if c["enc"] == "GCN" else adj)
        params = list(encoder.parameters()) + list(predictor.parameters()) + list(emb.parameters())
        optimizer = torch.optim.Adam(params, lr=0.01)
        pos_all = torch.stack([lo, hi], 1)
        weight_all = (w / w.max()).to(torch.float32).cuda() if c["weighted"] else None
        fn, weighted = P.loss.BY_NAME.get(c["loss"], (P.loss.auc_loss, False))
        encoder.train()
        predictor.train()
        torch.manual_seed(4242)
        losses = []
        for _ in range(3):
            neg_all = P.negative_sample.local_neg_sample(pos_all, N, c["k"]).cuda()
            pos_dev = pos_all.cuda()
            total = count = 0
            for perm in DataLoader(range(pos_all.size(0)), c["B"], shuffle=True):
                optimizer.zero_grad()
                h = encoder(emb.weight, adj_t)
                pos_edge = pos_dev[perm].t()
                neg_edge = torch.reshape(neg_all[perm], (-1, 2)).t()
                pos_out = predictor(h[pos_edge[0]], h[pos_edge[1]])
                neg_out = predictor(h[neg_edge[0]], h[neg_edge[1]])
                if weighted and weight_all is not None:
                    loss = fn(pos_out, neg_out, c["k"], weight_all[perm])
                elif weighted:
                    loss = P.loss.auc_loss(pos_out, neg_out, c["k"])
                else:
                    loss = fn(pos_out, neg_out, c["k"])
                loss.backward()
                if c["clip"] >= 0:
                    torch.nn.utils.clip_grad_norm_(encoder.parameters(), c["clip"])
                    torch.nn.utils.clip_grad_norm_(predictor.parameters(), c["clip"])
                optimizer.step()
                total += loss.item() * pos_out.size(0)
                count += pos_out.size(0)
            losses.append(total / count)
        losses = np.array(losses)
        ref32 = g[f"{name}_losses"]
        ref64 = _oracle_f64_losses(g, name, adj, N, lo, hi, w)
        # free-running: the first epoch agrees to fp32 round-off; afterwards Adam's 1/sqrt(v) turns round-off
        # in near-zero gradients into O(lr) weight moves and any two fp32 realisations drift apart
        # (measured: max |w - w64| ~ 4e-2 after 6 steps for this loop AND for BaseModel's fused step)
        close(losses[0], ref32[0], rtol=2e-5, msg=name)
        assert (np.abs(losses - ref64) <= 1e-3 * np.abs(ref64)).all(), (name, losses, ref32, ref64)
        # teacher-forced: along the float64 oracle's trajectory every step of the loop reproduces the
        # oracle's loss from the oracle's weights (no drift to hide behind)
        from tests.test_oracle import build_trainer_from_g8
        (enc_r, pred_r, emb_r), _ = build_trainer_from_g8(g, name, adj, N)
        a64 = O.gcn_norm_csr(adj) if c["enc"] == "GCN" else adj
        a64 = O.CSR(a64.rowptr, a64.col, None if a64.val is None else a64.val.double(), a64.n_cols)
        ref = O.TrainerRef(enc_r.double(), pred_r.double(), emb_r.double(), a64, loss_name=c["loss"], lr=0.01,
                           clip_norm=c["clip"])
        w64 = (w / w.max()).double() if c["weighted"] else None
        torch.manual_seed(4242)
        worst = 0.0
        for _ in range(2):
            _, neg_cpu = O.pos_neg_edges_ref("train", {"train": {"edge": pos_all}}, num_nodes=N,
                                             neg_sampler_name="local", num_neg=c["k"])
            for perm in O.batch_permutation(pos_all.size(0), c["B"], True):
                with torch.no_grad():
                    for dst_m, src_m in ((encoder, ref.encoder), (predictor, ref.predictor), (emb, ref.emb)):
                        for pd, ps in zip(dst_m.parameters(), src_m.parameters()):
                            pd.copy_(ps.detach().float())
                    h = encoder(emb.weight, adj_t)
                    pe, ne = pos_all[perm].t().cuda(), neg_cpu[perm].reshape(-1, 2).t().cuda()
                    pos_out, neg_out = predictor(h[pe[0]], h[pe[1]]), predictor(h[ne[0]], h[ne[1]])
                    if weighted and weight_all is not None:
                        got = fn(pos_out, neg_out, c["k"], weight_all[perm.cuda()])
                    elif weighted:
                        got = P.loss.auc_loss(pos_out, neg_out, c["k"])
                    else:
                        got = fn(pos_out, neg_out, c["k"])
                want, _, _ = ref.step(pos_all[perm], neg_cpu[perm], c["k"], None if w64 is None else w64[perm])
                worst = max(worst, abs(float(got) - float(want)) / abs(float(want)))
        print(f"{name}: teacher-forced worst relative loss deviation over the steps {worst:.3e}")
        assert worst <= 1e-6, (name, worst)


# (the 12-seed Hits@20 comparison on the N = 3000 random toy that stood here was statistically toothless -- at
# Hits@20 ~ 8 % the 20th of 10 000 negatives decides everything; it is replaced by
# tests/test_hip_round3.py::test_trained_regime_hits_parity_over_seeds on a learnable graph, 0.3 points asserted outright)


# ------------------------------------------------------- row-sharded encoder ----
@pytest.mark.parametrize("feat,out", [(64, 32), (256, 256)])
def test_sage_block_conv_equals_full_conv(P, feat, out):
    """ops.SAGEConvBlockFn over the 3 destination-row blocks of a graph == SAGEConvFn on the whole
    graph: outputs equal (same per-row arithmetic), the partial input gradients of the blocks
    add up to the full one, the weight gradients too (what the reduce-scatter / all-reduce of
    plnlp_amd/shard.py sum over ranks)."""
    from plnlp_amd import shard
    from plnlp_amd.ops import SAGEConvBlockFn, SAGEConvFn, _Act
    n, W = 301, 3
    csr = rand_csr(n, 4000, 23, weighted=False, hub=400)
    g = to_graph(P, csr)
    part = [shard.RowPartition(n, W, r) for r in range(W)]
    S, npad = part[0].rows, part[0].padded
    gen = torch.Generator().manual_seed(1)
    x = torch.zeros(npad, feat)
    x[:n] = torch.randn(n, feat, generator=gen)
    wl, wr = torch.randn(out, feat, generator=gen) * 0.1, torch.randn(out, feat, generator=gen) * 0.1
    bl = torch.randn(out, generator=gen)
    gy = torch.randn(npad, out, generator=gen)

    def leaves():
        return [dev(t).requires_grad_(True) for t in (x, wl, bl, wr)]

    xf, wlf, blf, wrf = leaves()
    gpad = P.Graph(torch.cat([g.rowptr, g.rowptr[-1:].expand(npad - n)]), g.col, None, npad, npad)
    y_full = SAGEConvFn.apply(xf, wlf, blf, wrf, gpad, _Act(True, 0.0, True), None, None, None)
    y_full.backward(dev(gy))
    acc = None
    ys = []
    for r in range(W):
        xb, wlb, blb, wrb = leaves()
        blk = g.row_block(part[r].lo, S, npad)
        yb = SAGEConvBlockFn.apply(xb, wlb, blb, wrb, blk, _Act(True, 0.0, True), part[r].lo)
        yb.backward(dev(gy)[part[r].lo:part[r].lo + S])
        ys.append(yb.detach())
        grads = [xb.grad, wlb.grad, blb.grad, wrb.grad]
        acc = grads if acc is None else [a + b for a, b in zip(acc, grads)]
    close(torch.cat(ys), y_full.detach(), rtol=1e-6)          # same kernels, same per-row sums (split-K may regroup)
    for got, want, name in zip(acc, (xf.grad, wlf.grad, blf.grad, wrf.grad), ("x", "wl", "bl", "wr")):
        close(got, want, rtol=3e-5, msg=name)


@pytest.mark.parametrize("feat,out", [(178, 64), (256, 256)])
def test_gcn_block_conv_equals_full_conv(P, feat, out):
    """ops.GCNConvBlockFn (aggregate first) over 3 destination-row blocks == GCNConvFn (transform first)
    on the whole normalised graph, to fp32 reassociation; input width 178 = the citation2 [emb | x] case
    (not a multiple of 4: padded operand)."""
    from plnlp_amd import shard
    from plnlp_amd.ops import GCNConvBlockFn, GCNConvFn, _Act
    n, W = 301, 3
    csr = rand_csr(n, 4000, 29, weighted=True, hub=400)
    g = to_graph(P, csr)
    part = [shard.RowPartition(n, W, r) for r in range(W)]
    S, npad = part[0].rows, part[0].padded
    gen = torch.Generator().manual_seed(2)
    x = torch.zeros(npad, feat)
    x[:n] = torch.randn(n, feat, generator=gen)
    w = torch.randn(out, feat, generator=gen) * 0.1
    b = torch.randn(out, generator=gen)
    gy = torch.randn(npad, out, generator=gen)

    def leaves():
        return [dev(t).requires_grad_(True) for t in (x, w, b)]

    xf, wf, bf = leaves()
    gpad = P.Graph(torch.cat([g.rowptr, g.rowptr[-1:].expand(npad - n)]), g.col, g.val, npad, npad)
    y_full = GCNConvFn.apply(xf, wf, bf, gpad, _Act(True, 0.0, True), None, None, None)
    y_full.backward(dev(gy))
    want = torch.relu(O.spmm(csr, x[:n].double(), "sum", True) @ w.double().T + b.double())
    acc, ys = None, []
    for r in range(W):
        xb, wb, bb = leaves()
        blk = g.row_block(part[r].lo, S, npad)
        yb = GCNConvBlockFn.apply(xb, wb, bb, blk, _Act(True, 0.0, True))
        yb.backward(dev(gy)[part[r].lo:part[r].lo + S])
        ys.append(yb.detach())
        grads = [xb.grad, wb.grad, bb.grad]
        assert xb.grad.shape == x.shape and wb.grad.shape == w.shape
        acc = grads if acc is None else [a + c for a, c in zip(acc, grads)]
    close(torch.cat(ys)[:n], want, msg="vs oracle")
    close(torch.cat(ys)[:n], y_full.detach()[:n], rtol=3e-5, msg="vs transform-first")
    for got, ref, name in zip(acc, (xf.grad, wf.grad, bf.grad), ("x", "w", "b")):
        close(got, ref, rtol=5e-5, msg=name)


def test_sharded_gcn_feature_step_on_one_rank_rccl_group_matches_plain_step(P):
    """the citation2 recipe's shape through dp_exchange='shard' on a 1-rank RCCL group: GCN x2 on
    [embedding | features] (width 40 + 18 = 58, padded operand), MLP predictor, AUC loss"""
    import torch.distributed as dist
    from test_sharded_encoder import _free_port
    from plnlp_amd import synthetic
    n, h, B, k = 2000, 64, 1024, 1
    g = synthetic.make_graph("collab", seed=5, device="cpu", num_nodes=n, num_edges=12000, weighted=False)
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                                device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        def make(pg, exchange):
            m = P.BaseModel(lr=0.01, dropout=0.0, grad_clip_norm=1.0, gnn_num_layers=2, mlp_num_layers=2,
                            emb_hidden_channels=40, gnn_hidden_channels=h, mlp_hidden_channels=h, num_nodes=n,
                            num_node_feats=18, gnn_encoder_name="GCN", predictor_name="MLP",
                            loss_func="AUC", optimizer_name="Adam", device="cuda",
                            use_node_feats=True, train_node_emb=True, process_group=pg, dp_exchange=exchange)
            torch.manual_seed(37)
            m.param_init()
            return m
        plain, sharded = make(None, "auto"), make(dist.group.WORLD, "shard")
        data = g["data"]
        data.adj_t = P.gcn_normalization(g["adj_t"].to("cuda"))
        data.x = torch.randn(n, 18, generator=torch.Generator().manual_seed(8)).cuda()
        split = {"train": {"edge": g["edges"]}}
        la, lb = [], []
        for epoch in range(2):
            torch.manual_seed(60 + epoch)
            la.append(plain.train(data, split, B, "global", k))
            torch.manual_seed(60 + epoch)
            lb.append(sharded.train(data, split, B, "global", k))
        close(np.array(lb), np.array(la), rtol=3e-4)
        close(sharded.emb.weight, plain.emb.weight, rtol=1e-3, atol=2e-2)
        assert sharded.check_replicas()
    finally:
        dist.destroy_process_group()


def test_sharded_step_on_one_rank_rccl_group_matches_plain_step(P):
    """BaseModel(dp_exchange='shard') driven through a 1-rank RCCL group on the GPU: every collective of
    the sharded step runs (all-gather, all-to-all, reduce-scatter, all-reduce), on the HIP kernels, and
    the losses track the plain single-process trainer from the same weights and batches."""
    import torch.distributed as dist
    from test_sharded_encoder import _free_port
    n, h, B, k = 2000, 64, 1024, 1
    from plnlp_amd import synthetic
    g = synthetic.make_graph("collab", seed=4, device="cpu", num_nodes=n, num_edges=12000, weighted=True)
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                                device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        def make(pg, exchange):
            m = P.BaseModel(lr=0.01, dropout=0.0, grad_clip_norm=1.0, gnn_num_layers=2, mlp_num_layers=2,
                            emb_hidden_channels=h, gnn_hidden_channels=h, mlp_hidden_channels=h, num_nodes=n,
                            num_node_feats=0, gnn_encoder_name="SAGE", predictor_name="DOT",
                            loss_func="WeightedHingeAUC", optimizer_name="Adam", device="cuda",
                            use_node_feats=False, train_node_emb=True, process_group=pg, dp_exchange=exchange)
            torch.manual_seed(31)
            m.param_init()
            return m
        plain, sharded = make(None, "auto"), make(dist.group.WORLD, "shard")
        assert sharded.dp_mode() == "shard"
        data = g["data"]
        data.adj_t = g["adj_t"].to("cuda")
        split = {"train": {"edge": g["edges"], "weight": g["weight"] / 5.0}}
        la, lb = [], []
        for epoch in range(2):
            torch.manual_seed(50 + epoch)
            la.append(plain.train(data, split, B, "local", k))
            torch.manual_seed(50 + epoch)
            lb.append(sharded.train(data, split, B, "local", k))
        close(np.array(lb), np.array(la), rtol=2e-4)
        close(sharded.emb.weight, plain.emb.weight, rtol=1e-3, atol=2e-2)       # Adam: O(lr) on round-off-zero grads
        assert sharded.check_replicas()
    finally:
        dist.destroy_process_group()


# ------------------------------------------- dense products: f32 MFMA vs split-bf16 ----
def _gemm_err(P, a, b, at, bt, mode):
    """max and rms of |C - C_fp64| / sum_k |a||b| (the scale every fp32 product-sum error is relative to)"""
    old = P.ops.GEMM_MATH["mode"]
    P.ops.GEMM_MATH["mode"] = mode
    try:
        got = P.ops.gemm([(dev(a), dev(b))], at, bt).double().cpu()
    finally:
        P.ops.GEMM_MATH["mode"] = old
    a64 = (a.T if at else a).double()
    b64 = (b.T if bt else b).double()
    ref, mag = a64 @ b64, a64.abs() @ b64.abs()
    e = (got - ref).abs() / mag.clamp_min(1e-300)
    return float(e.max()), float((e ** 2).mean().sqrt()), got


@pytest.mark.parametrize("at,bt,m,n,k", [(False, True, 1500, 256, 512), (False, False, 1500, 384, 256),
                                         (True, False, 256, 200, 30000), (False, True, 700, 200, 180)])
@pytest.mark.parametrize("spread", [0, 12])
def test_split_bf16_products_are_fp32_grade(P, at, bt, m, n, k, spread):
    """PLNLP_GEMM_MATH_BF16X3 (three bf16 terms per operand element, six bf16 MFMAs per block) against fp64,
    beside the f32-input MFMA on the same operands: the error -- relative to sum_k |a||b| -- stays at f32
    round-off (<= 2^-22 worst element) and within 1.5x of the f32 MFMA's own error.  spread = 12: every element
    scaled by 10^U(-6, 6), so neighbouring elements of a tile differ by up to 12 decades (the split is per
    element: no shared exponent)."""
    gen = torch.Generator().manual_seed(at * 7 + bt * 3 + k + spread)

    def operand(r, c):
        x = torch.randn(r, c, generator=gen)
        if spread:
            x = x * 10.0 ** ((torch.rand(r, c, generator=gen) - 0.5) * spread)
        return x
    a = operand(k, m) if at else operand(m, k)
    b = operand(n, k) if bt else operand(k, n)
    mx3, rms3, _ = _gemm_err(P, a, b, at, bt, "bf16x3")
    mx1, rms1, _ = _gemm_err(P, a, b, at, bt, "f32")
    print(f"a_trans={at} b_trans={bt} {m}x{n}x{k} spread={spread}: bf16x3 max {mx3:.2e} rms {rms3:.2e}; "
          f"f32 MFMA max {mx1:.2e} rms {rms1:.2e}")
    bound = 2.0 ** -22 if spread == 0 else 2.0 ** -20      # (a few huge terms dominate sum |a||b| when spread out)
    assert mx3 <= bound and mx1 <= bound
    assert rms3 <= 1.5 * rms1 + 1e-9 and mx3 <= 1.5 * mx1 + 2e-8


def test_split_bf16_products_exact_cases(P):
    """what must not depend on the split: zeros stay zeros (a zero row of A -> an exactly zero row of C),
    products of small integers are exact, and two launches give identical bits"""
    gen = torch.Generator().manual_seed(5)
    a = torch.randint(-8, 9, (300, 96), generator=gen).float()
    b = torch.randint(-8, 9, (130, 96), generator=gen).float()
    a[7] = 0.0
    a *= 2.0 ** -20                  # an exact scaling: still exact products and sums
    for mode in ("bf16x3", "f32"):
        _, _, got = _gemm_err(P, a, b, False, True, mode)
        assert torch.equal(got, a.double() @ b.double().T), mode
        assert (got[7] == 0).all()
    x, w = torch.randn(777, 300, generator=gen), torch.randn(130, 300, generator=gen)
    assert torch.equal(_gemm_err(P, x, w, False, True, "bf16x3")[2], _gemm_err(P, x, w, False, True, "bf16x3")[2])


# ------------------------------- Hadamard of the endpoint rows inside the GEMM loaders ----
@pytest.mark.parametrize("n,k,e,nout", [(500, 64, 3000, 64), (4267, 512, 20000, 512), (300, 200, 1000, 200), (64, 36, 129, 50)])
def test_gemm_pair_gathered_operands_equal_materialised_hadamard(P, n, k, e, nout):
    """a_index/a_index2 (forward: (h[src] * h[dst]) W^T) and b_index/b_index2 (weight gradient:
    dz^T (h[src] * h[dst])) against the same products on the materialised Hadamard -- the same f32 product
    enters the same split, so the bits agree -- and against fp64."""
    gen = torch.Generator().manual_seed(n + e)
    h = torch.randn(n, k, generator=gen)
    w = torch.randn(nout, k, generator=gen) * 0.1
    src = torch.randint(0, n, (e,), generator=gen)
    dst = torch.randint(0, n, (e,), generator=gen)
    dz = torch.randn(e, nout, generator=gen)
    had = dev(h)[dev(src)] * dev(h)[dev(dst)]
    s32, d32 = dev(src).to(torch.int32), dev(dst).to(torch.int32)
    old, old_st = P.ops.GEMM_MATH["mode"], P.ops.GEMM_STATIONARY_B["enabled"]
    P.ops.GEMM_MATH["mode"] = "bf16x3"
    # (the pair loaders live in the 128 x 128 kernels: the bit-for-bit reference is the SAME kernel on the materialised
    #  operand -- the stationary-weights kernel sums the same products in another order, tests/test_hip_round4.py)
    P.ops.GEMM_STATIONARY_B["enabled"] = False
    try:
        y = P.ops.gemm([(dev(h), dev(w))], False, True, a_index=[s32], a_index2=d32)
        y_ref = P.ops.gemm([(had, dev(w))], False, True)
        gw = P.ops.gemm([(dev(dz), dev(h))], True, False, b_index=s32, b_index2=d32)
        gw_ref = P.ops.gemm([(dev(dz), had)], True, False)
        P.ops.GEMM_MATH["mode"] = "f32"
        with pytest.raises(Exception):          # the f32-MFMA form has no pair loader: it must refuse, not ignore
            P.ops.gemm([(dev(h), dev(w))], False, True, a_index=[s32], a_index2=d32)
    finally:
        P.ops.GEMM_MATH["mode"], P.ops.GEMM_STATIONARY_B["enabled"] = old, old_st
    assert torch.equal(y, y_ref) and torch.equal(gw, gw_ref)
    had64 = h.double()[src] * h.double()[dst]
    close(y, had64 @ w.double().T, atol=2e-5 * np.sqrt(k))
    close(gw, dz.double().T @ had64, atol=3e-5 * np.sqrt(e))


@pytest.mark.parametrize("layers,hidden", [(2, 64), (3, 128)])
def test_fused_edge_mlp_equals_hadamard_then_stack(P, layers, hidden):
    """MLPPredictor.score_edges with ops.FUSE_EDGE_MLP on (ops.EdgeMLPFn) and off (EdgeHadamardFn +
    MLPStackFn): same scores bit for bit in eval mode, every gradient (h and all weights) to fp32 round-off;
    and against the oracle predictor in fp64."""
    from plnlp_amd import ops
    n, e = 700, 5000
    gen = torch.Generator().manual_seed(layers)
    h0 = torch.randn(n, hidden, generator=gen)
    src, dst = torch.randint(0, n, (e,), generator=gen), torch.randint(0, n, (e,), generator=gen)
    gy = torch.randn(e, 1, generator=gen)
    pred = P.layer.MLPPredictor(hidden, hidden, 1, layers, 0.0).cuda()
    torch.manual_seed(3)
    pred.reset_parameters()
    outs = {}
    for fused in (True, False):
        ops.FUSE_EDGE_MLP["enabled"] = fused
        try:
            h = dev(h0).requires_grad_(True)
            pred.zero_grad()
            y = pred.score_edges(h, dev(src), dev(dst))
            assert (type(y.grad_fn).__name__ == "EdgeMLPFnBackward") == fused
            y.backward(dev(gy))
            outs[fused] = [y.detach(), h.grad] + [p.grad.clone() for p in pred.parameters()]
        finally:
            ops.FUSE_EDGE_MLP["enabled"] = False
    assert torch.equal(outs[True][0], outs[False][0])
    for a, b in zip(outs[True][1:], outs[False][1:]):
        close(a, b, rtol=2e-6)
    ref = O.MLPPredictorRef(hidden, hidden, 1, layers, 0.0).double()
    ref.load_state_dict({k_: v.detach().cpu().double() for k_, v in pred.state_dict().items()})
    h64 = h0.double().requires_grad_(True)
    y64 = ref(h64[src], h64[dst])
    y64.backward(gy.double())
    close(outs[True][0], y64.detach())
    close(outs[True][1], h64.grad, rtol=3e-5)
    for got, p64 in zip(outs[True][2:], ref.parameters()):
        close(got, p64.grad, rtol=3e-5)


def test_fused_edge_mlp_in_the_training_step(P):
    """the ddi recipe's shape (SAGE x2 + MLP predictor, row-sparse backward, touched-rows forward) trained
    3 epochs with the fused predictor and with the unfused one from the same weights: same losses"""
    from plnlp_amd import ops, synthetic
    n, h, B, k = 3000, 64, 2048, 3
    g = synthetic.make_graph("ddi", seed=6, device="cpu", num_nodes=n, num_edges=30000)
    data = g["data"]
    data.adj_t = g["adj_t"].to("cuda")
    split = {"train": {"edge": g["edges"]}}
    losses = {}
    for fused in (True, False):
        ops.FUSE_EDGE_MLP["enabled"] = fused
        try:
            m = P.BaseModel(lr=0.005, dropout=0.3, grad_clip_norm=2.0, gnn_num_layers=2, mlp_num_layers=3,
                            emb_hidden_channels=h, gnn_hidden_channels=h, mlp_hidden_channels=h, num_nodes=n,
                            num_node_feats=0, gnn_encoder_name="SAGE", predictor_name="MLP", loss_func="AUC",
                            optimizer_name="Adam", device="cuda", use_node_feats=False, train_node_emb=True)
            torch.manual_seed(11)
            P.manual_seed(11)
            m.param_init()
            out = []
            for ep in range(3):
                torch.manual_seed(100 + ep)
                out.append(m.train(data, split, B, "global", k))
            losses[fused] = np.array(out)
        finally:
            ops.FUSE_EDGE_MLP["enabled"] = False
    close(losses[True], losses[False], rtol=1e-4)
    close(losses[True][:1], losses[False][:1], rtol=2e-6)


# ------------------------------- embedding update in the gradient kernel's epilogue ----
@pytest.mark.parametrize("pred,layers,feat", [("DOT", 1, 64), ("MLP", 1, 256), ("DOT", 2, 64)])
def test_embedding_adam_in_the_aggregation_epilogue_gives_the_same_bits(P, pred, layers, feat):
    """model.FUSE_EMBEDDING_ADAM: the table's Adam step applied by the transposed aggregation that finishes its
    gradient (PLNLP_EPI_ADAM; hub rows through the finalize pass) instead of a gradient tensor + the optimiser
    kernel -- the same arithmetic on the same values: parameters, Adam moments and losses after 3 epochs are
    bit-identical.  (2 layers: the first conv's backward is dense, the path falls back to the gradient tensor.)"""
    from plnlp_amd import model as M, synthetic
    n, B, k = 3000, 1024, 1
    g = synthetic.make_graph("collab", seed=9, device="cpu", num_nodes=n, num_edges=20000, weighted=True)
    data = g["data"]
    data.adj_t = g["adj_t"].to("cuda")
    assert int((data.adj_t.rowptr[1:] - data.adj_t.rowptr[:-1]).max()) > 256       # a hub row: chunk + finalize passes
    split = {"train": {"edge": g["edges"], "weight": g["weight"] / 5.0}}
    res = {}
    for fused in (True, False):
        M.FUSE_EMBEDDING_ADAM["enabled"] = fused
        try:
            m = P.BaseModel(lr=0.01, dropout=0.3, grad_clip_norm=1.0, gnn_num_layers=layers, mlp_num_layers=2,
                            emb_hidden_channels=feat, gnn_hidden_channels=feat, mlp_hidden_channels=feat, num_nodes=n,
                            num_node_feats=0, gnn_encoder_name="SAGE", predictor_name=pred,
                            loss_func="WeightedHingeAUC" if pred == "DOT" else "AUC", optimizer_name="Adam",
                            device="cuda", use_node_feats=False, train_node_emb=True)
            torch.manual_seed(21)
            P.manual_seed(21)
            m.param_init()
            losses = []
            for ep in range(3):
                torch.manual_seed(70 + ep)
                losses.append(m.train(data, split, B, "local", k))
            torch.cuda.synchronize()
            st = m.optimizer.state[m.emb.weight]
            res[fused] = (losses, [p.detach().clone() for p in m.para_list], st["exp_avg"].clone(),
                          st["exp_avg_sq"].clone(), st["step"])
        finally:
            M.FUSE_EMBEDDING_ADAM["enabled"] = True
    assert res[True][0] == res[False][0]
    assert res[True][4] == res[False][4]
    for a, b in zip(res[True][1], res[False][1]):
        assert torch.equal(a, b)
    assert torch.equal(res[True][2], res[False][2]) and torch.equal(res[True][3], res[False][3])


# ------------------------------------------------- aggregation: feature slabs ----
@pytest.mark.parametrize("feat", [256, 512, 384])
@pytest.mark.parametrize("tune", [16, 32])
def test_csr_aggregate_feature_slab_forms_match_oracle(P, feat, tune):
    """one wave per (row, 128- or 256-column slab) -- the HBM-bound form -- against the oracle, with the
    epilogues the training path uses on it (bias, accumulate, indexed addend + gate), hub rows included"""
    from plnlp_amd import _lib
    csr = rand_csr(300, 3000, feat + tune, weighted=True, hub=700)
    g = to_graph(P, csr)
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(300, feat, generator=gen)
    for reduce, use_values in (("mean", False), ("sum", True)):
        ref = O.spmm(csr, x.double(), reduce, use_values)
        out = P.ops.csr_aggregate(g, dev(x), reduce, use_values, tune=tune)
        close(out, ref, msg=f"{reduce} feat={feat} tune={tune}")
    bias = torch.randn(feat, generator=gen)
    base = torch.randn(300, feat, generator=gen)
    out = dev(base.clone())
    P.ops.csr_aggregate(g, dev(x), "sum", True, out=out, tune=tune,
                        epilogue=_lib.make_epilogue(bias=dev(bias), relu=True, accumulate=True))
    close(out, base.double() + torch.relu(O.spmm(csr, x.double(), "sum", True) + bias.double()))
    gate = torch.randn(300, feat, generator=gen)
    idx = torch.full((300,), -1, dtype=torch.int32)
    idx[::3] = torch.arange(100, dtype=torch.int32)
    add = torch.randn(100, feat, generator=gen)
    out = P.ops.csr_aggregate(g, dev(x), "sum", True, tune=tune,
                              epilogue=_lib.make_epilogue(addend=dev(add), addend_index=dev(idx), gate=dev(gate),
                                                          gate_scale=1.5))
    want = O.spmm(csr, x.double(), "sum", True)
    want[::3] += add.double()
    want = torch.where(gate > 0, want * 1.5, torch.zeros_like(want))
    close(out, want)


# ------------------------------------------- last conv at the touched rows only ----
@pytest.mark.parametrize("enc,layers,pred,in_feats", [("SAGE", 1, "DOT", 0), ("SAGE", 1, "MLP", 0), ("SAGE", 2, "DOT", 0),
                                                     ("GCN", 1, "DOT", 0), ("GCN", 2, "MLP", 50), ("GCN", 3, "MLP", 0)])
def test_row_restricted_last_conv_equals_full_forward(P, enc, layers, pred, in_feats):
    """ops.SPARSE_FORWARD: the last conv produces only the rows the batch's edges touch (row-indexed
    aggregation, gathered root operand, dropout drawn at the original row positions).  Against the full
    forward with the same row-sparse backward: identical loss bits over several steps (every produced
    row is the same arithmetic), gradients and updated weights equal to fp32 round-off."""
    from plnlp_amd import ops
    from gpu_util import rand_csr
    n, feat, batch, k = 3000, 64, 512, 2
    csr = rand_csr(n, 8 * n, 31, weighted=False, hub=900)
    r, c, _ = csr.coo()
    adj = P.Graph.from_coo(torch.cat([r, c]), torch.cat([c, r]), None, n, n).to("cuda")
    if enc == "GCN":
        adj = P.gcn_normalization(adj)

    class D:
        pass
    data = D()
    data.adj_t = adj
    if in_feats:
        data.x = torch.randn(n, in_feats, generator=torch.Generator().manual_seed(8)).cuda()
    gen = torch.Generator().manual_seed(9)
    steps = 3
    pos = torch.randint(0, n, (steps * batch, 2), generator=gen).cuda()
    pos[:40, 0] = 3                                   # a hub of the graph is a hot node of the batch too
    neg = torch.randint(0, n, (steps * batch, k, 2), generator=gen).cuda()
    w = torch.rand(steps * batch, generator=gen).cuda()
    res = {}
    old_f, old_b = dict(ops.SPARSE_FORWARD), dict(ops.SPARSE_BACKWARD)
    from plnlp_amd import model as model_mod
    fuse_old = model_mod.FUSE_EMBEDDING_ADAM["enabled"]
    model_mod.FUSE_EMBEDDING_ADAM["enabled"] = False          # this comparison reads the embedding's gradient tensor
    try:
        ops.SPARSE_BACKWARD["max_expected_fraction"] = 1.0
        for mode in (False, True):
            ops.SPARSE_FORWARD["enabled"] = mode
            torch.manual_seed(77)
            P.manual_seed(77)
            m = P.BaseModel(lr=1e-3, dropout=0.3, grad_clip_norm=1.0, gnn_num_layers=layers, mlp_num_layers=2,
                            emb_hidden_channels=feat, gnn_hidden_channels=feat, mlp_hidden_channels=feat,
                            num_nodes=n, num_node_feats=in_feats, gnn_encoder_name=enc, predictor_name=pred,
                            loss_func="WeightedHingeAUC", optimizer_name="Adam", device="cuda",
                            use_node_feats=in_feats > 0, train_node_emb=True)
            m.param_init()
            m.encoder.train()
            m.predictor.train()
            losses, first_grads = [], None
            for i in range(steps):
                sl = slice(i * batch, (i + 1) * batch)
                losses.append(float(m.train_step(data, pos[sl], neg[sl], k, w[sl])))
                if i == 0:
                    first_grads = [p.grad.clone() for p in m.para_list]
            res[mode] = (losses, first_grads, [p.detach().clone() for p in m.para_list])
    finally:
        ops.SPARSE_FORWARD.update(old_f)
        ops.SPARSE_BACKWARD.update(old_b)
        model_mod.FUSE_EMBEDDING_ADAM["enabled"] = fuse_old
    (lf, gf, wf), (lr_, gr, wr) = res[False], res[True]
    assert lf[0] == lr_[0], (lf, lr_)
    close(np.array(lr_), np.array(lf), rtol=1e-6)
    for a, b in zip(gf, gr):
        scale = max(1e-6, float(a.abs().max()))
        assert float((a - b).abs().max()) <= 2e-6 * scale + 1e-7
    for a, b in zip(wf, wr):
        # Adam: a coordinate whose gradient is round-off around zero may move by lr in either direction
        assert float((a - b).abs().max()) <= 3 * 1e-3 + 1e-6
        assert float(((a - b).abs() <= 1e-6).float().mean()) >= 0.999
