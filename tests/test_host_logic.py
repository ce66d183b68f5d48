"""CPU tests of the host-side mirror of the reference interface (plnlp_amd's samplers,
edge plumbing, batch permutation, evaluator, logger, factories, Graph) against the fixtures
captured from the reference and against the oracle.  No kernels are launched here."""
import io

import os

import numpy as np
import pytest
import torch

import oracle as O
import plnlp_amd as P
from plnlp_amd import negative_sample as NS, utils as U

T = torch.from_numpy


def test_local_sampler_bit_exact_vs_reference(golden):
    g = golden("g4_g5_samplers")
    torch.manual_seed(0)
    np.testing.assert_array_equal(NS.local_neg_sample(torch.tensor([[0, 1], [2, 3]]), 10, 3).numpy(), g["local_s0"])
    for seed in (11, 12, 13):
        torch.manual_seed(seed)
        out = NS.local_neg_sample(T(g[f"local_{seed}_pos"]), int(g[f"local_{seed}_N"]), int(g[f"local_{seed}_k"]))
        np.testing.assert_array_equal(out.numpy(), g[f"local_{seed}_out"])


def test_perm_copy_and_global_padding_bit_exact(golden, monkeypatch):
    g = golden("g4_g5_samplers")
    ei = T(g["permcopy_in"])
    torch.manual_seed(21)
    np.testing.assert_array_equal(NS.sample_perm_copy(ei, 8, 3).numpy(), g["permcopy_out_t8_c3"])
    torch.manual_seed(22)
    np.testing.assert_array_equal(NS.sample_perm_copy(ei, 5, 2).numpy(), g["permcopy_out_t5_c2"])
    # padding branch: same short structured sample as the fixture -> same output
    short = T(g["globalpad_short"])
    seen = {}

    def fake(edge_index, num_nodes, num_neg_samples, method="sparse", generator=None):
        seen.update(ei=edge_index, n=num_nodes, m=num_neg_samples)
        return short
    monkeypatch.setattr(NS, "structured_negative_sampling", fake)
    torch.manual_seed(23)
    out = NS.global_neg_sample(torch.tensor([[0, 1, 2], [1, 2, 0]]), 10, 3, 2)
    np.testing.assert_array_equal(out.numpy(), g["globalpad_out"])
    NS.global_neg_sample(torch.tensor([[0, 1, 2], [1, 2, 0]]), 10, 2, 2)
    np.testing.assert_array_equal(seen["ei"].numpy(), g["globalcall_edge_index"])     # edges + self loops
    assert seen["n"] == int(g["globalcall_n"]) and seen["m"] == int(g["globalcall_m"])


def test_structured_negative_sampling_contract():
    rng = np.random.default_rng(1)
    n = 80
    lo, hi = np.triu_indices(n, 1)
    pick = rng.choice(lo.size, 900, replace=False)
    ei = torch.tensor(np.stack([np.r_[lo[pick], hi[pick]], np.r_[hi[pick], lo[pick]]]))
    torch.manual_seed(3)
    out = NS.global_neg_sample(ei, n, 900, 3)
    assert out.shape == (900, 3, 2) and out.dtype == torch.int64
    flat = out.reshape(-1, 2)
    keys = (flat[:, 0] * n + flat[:, 1]).tolist()
    assert (flat[:, 0] != flat[:, 1]).all()
    assert len(set(keys)) >= 0.97 * len(keys)        # repeats only through the reference's padding branch
    assert not set((ei[0] * n + ei[1]).tolist()).intersection(keys)
    # roughly uniform over the free cells: chi-square on row marginals
    counts = np.bincount(flat[:, 0].numpy(), minlength=n)
    assert counts.std() / counts.mean() < 0.35


def test_batch_permutation_bit_exact_vs_dataloader(golden):
    g = golden("g6_dataloader")
    for seed, n, B in [(123, 10, 4), (5, 1000, 64), (77, 65, 65), (8, 3, 10)]:
        torch.manual_seed(seed)
        batches = U.batch_permutation(n, B, True)
        after = torch.randint(0, 1 << 30, (4,))
        np.testing.assert_array_equal(torch.cat(batches).numpy(), g[f"s{seed}_n{n}_B{B}_perm"])
        assert [b.numel() for b in batches] == g[f"s{seed}_n{n}_B{B}_sizes"].tolist()
        np.testing.assert_array_equal(after.numpy(), g[f"s{seed}_n{n}_B{B}_after"])     # same RNG consumption
    torch.manual_seed(9)
    U.batch_permutation(10, 4, False)
    np.testing.assert_array_equal(torch.randint(0, 1 << 30, (4,)).numpy(), g["noshuffle_after"])


def test_get_pos_neg_edges_vs_reference(golden):
    g = golden("g7_pos_neg_edges")
    se = {s: {k: T(g[f"cit_in_{s}_{k}"]) for k in ("source_node", "target_node", "target_node_neg")}
          for s in ("train", "valid", "test")}
    for s in ("valid", "test"):
        pos, neg = U.get_pos_neg_edges(s, se)
        np.testing.assert_array_equal(pos.numpy(), g[f"cit_{s}_pos"])
        np.testing.assert_array_equal(neg.numpy(), g[f"cit_{s}_neg"])
    torch.manual_seed(71)
    pos, neg = U.get_pos_neg_edges("train", se, num_nodes=40, neg_sampler_name="local", num_neg=3)
    np.testing.assert_array_equal(pos.numpy(), g["cit_train_pos"])
    np.testing.assert_array_equal(neg.numpy(), g["cit_train_neg"])
    se2 = {"train": {"edge": torch.zeros(1, 2, dtype=torch.long)},
           "valid": {"edge": T(g["edge_in_valid_edge"]), "edge_neg": T(g["edge_in_valid_edge_neg"])}}
    pos, neg = U.get_pos_neg_edges("valid", se2)
    np.testing.assert_array_equal(neg.numpy(), g["edge_valid_neg"])


def test_factories_follow_reference_names(golden):
    g = golden("g2_predictors")
    for name, want in zip(g["pred_factory_keys"].tolist(), g["pred_factory_vals"].tolist()):
        got = P.create_predictor_layer(8, 2, 0.0, name)
        assert ("None" if got is None else type(got).__name__) == want, name
    for name, want in zip(g["enc_factory_keys"].tolist(), g["enc_factory_vals"].tolist()):
        if want in ("WSAGE", "Transformer"):
            with pytest.raises(NotImplementedError):
                P.create_gnn_layer(4, 8, 2, 0.0, name)
        else:
            assert type(P.create_gnn_layer(4, 8, 2, 0.0, name)).__name__ == want
    # state_dict keys = PyG's (weight exchange with a reference checkpoint)
    assert list(P.SAGE(4, 8, 8, 2, 0.0).state_dict()) == [
        "convs.0.lin_l.weight", "convs.0.lin_l.bias", "convs.0.lin_r.weight",
        "convs.1.lin_l.weight", "convs.1.lin_l.bias", "convs.1.lin_r.weight"]
    assert list(P.GCN(4, 8, 8, 1, 0.0).state_dict()) == ["convs.0.bias", "convs.0.lin.weight"]
    assert list(P.MLPPredictor(8, 8, 1, 2, 0.0).state_dict()) == ["lins.0.weight", "lins.0.bias", "lins.1.weight",
                                                                    "lins.1.bias"]
    # init laws: same draws as the oracle modules under the same seed
    torch.manual_seed(5)
    a = P.SAGE(6, 6, 6, 2, 0.0)
    a.reset_parameters()
    torch.manual_seed(5)
    b = O.GNNRef("SAGE", 6, 6, 6, 2, 0.0)
    b.reset_parameters()
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(v, w), k
    torch.manual_seed(6)
    a = P.GCN(6, 6, 6, 2, 0.0)
    a.reset_parameters()
    torch.manual_seed(6)
    b = O.GNNRef("GCN", 6, 6, 6, 2, 0.0)
    b.reset_parameters()
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(v, w), k


def test_create_input_layer_branches(tmp_path):
    d, e = P.create_input_layer(100, 7, 16, use_node_feats=False)
    assert d == 16 and e.weight.shape == (100, 16)
    d, e = P.create_input_layer(100, 7, 16, use_node_feats=True, train_node_emb=True)
    assert d == 23 and e.weight.shape == (100, 16)
    d, e = P.create_input_layer(100, 7, 16, use_node_feats=True, train_node_emb=False)
    assert d == 7 and e is None
    path = str(tmp_path / "emb.pt")
    torch.save(torch.randn(100, 5), path)
    d, e = P.create_input_layer(100, 7, 16, use_node_feats=True, train_node_emb=False, pretrain_emb=path)
    assert d == 12 and not e.weight.requires_grad
    d, e = P.create_input_layer(100, 7, 16, use_node_feats=False, pretrain_emb=path)
    assert d == 5


def test_logger_and_adjust_lr_vs_reference(golden):
    g = golden("g9_logger")
    res = g["results"]
    lg = P.Logger(3)
    for r in range(3):
        for e in range(5):
            lg.add_result(r, (float(res[r, e, 0]), float(res[r, e, 1])))
    calls = {"run1": dict(run=1), "run1_last": dict(run=1, last_best=True), "all": dict(), "all_last": dict(last_best=True)}
    for key, want in zip(g["keys"].tolist(), g["texts"].tolist()):
        buf = io.StringIO()
        lg.print_statistics(f=buf, **calls[key])
        assert buf.getvalue() == want, key
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=0.01)
    lrs = [P.adjust_lr(opt, r, 0.01) for r in (0.0, 0.25, 0.5, 0.99995, 1.0)]
    np.testing.assert_allclose(lrs, g["lrs"], rtol=1e-12)


def test_evaluator_matches_oracle_and_bruteforce():
    gen = torch.Generator().manual_seed(5)
    pos, neg = torch.randn(300, generator=gen), torch.randn(700, generator=gen)
    ev = U.Evaluator("ogbl-collab")
    res = U.evaluate_hits(ev, pos, neg, pos[:100], neg[:400])
    ref = O.evaluate_hits_ref(pos, neg, pos[:100], neg[:400])
    assert res == ref
    negm = torch.randn(300, 40, generator=gen)
    res = U.evaluate_mrr(U.Evaluator("ogbl-citation2"), pos, negm.reshape(-1), pos, negm.reshape(-1))
    ranks = 1 + (negm > pos[:, None]).sum(1)
    assert abs(res["MRR"][0] - float((1.0 / ranks.float()).mean())) < 1e-6
    assert U.Evaluator("ogbl-ddi").K == 20 and U.Evaluator("ogbl-collab").K == 50


def test_graph_construction_and_normalisation_vs_oracle():
    gen = torch.Generator().manual_seed(8)
    n = 60
    r, c = torch.randint(0, n, (400,), generator=gen), torch.randint(0, n, (400,), generator=gen)
    v = torch.rand(400, generator=gen) + 0.1
    g = P.Graph.from_coo(r, c, v, n, n)
    o = O.CSR.from_coo(r, c, v, n)
    assert torch.equal(g.rowptr, o.rowptr) and torch.equal(g.col.long(), o.col) and torch.equal(g.val, o.val)
    gt, ot = g.t(), o.t()
    assert torch.equal(gt.rowptr, ot.rowptr) and torch.equal(gt.col.long(), ot.col) and torch.equal(gt.val, ot.val)
    assert g.size(0) == n and g.nnz == 400
    np.testing.assert_allclose(g.sum(dim=1).numpy(), o.rowsum().numpy(), rtol=1e-6)
    gn, on = P.gcn_normalization(g), O.gcn_norm_csr(o)
    assert torch.equal(gn.rowptr, on.rowptr) and torch.equal(gn.col.long(), on.col)
    np.testing.assert_allclose(gn.val.numpy(), on.val.numpy(), rtol=1e-6)
    sym = P.Graph.from_coo(r, c, None, n, n).to_symmetric()
    d = sym.coo()
    dense = torch.zeros(n, n)
    dense[d[0], d[1]] = 1
    assert torch.equal(dense, dense.t()) and int(dense.sum()) == sym.nnz
    ei = torch.stack([r, c])
    adj_t = P.Graph.from_edge_index(ei, None, n)          # transposed adjacency: row = target
    rr, cc, _ = adj_t.coo()
    assert set(zip(rr.tolist(), cc.tolist())) == set(zip(c.tolist(), r.tolist()))


def test_loss_dispatch_names():
    from plnlp_amd import loss
    assert loss.BY_NAME["WeightedHingeAUC"][0] is loss.weighted_hinge_auc_loss and loss.BY_NAME["WeightedHingeAUC"][1]
    assert set(loss.BY_NAME) == {"CE", "InfoNCE", "LogRank", "HingeAUC", "AdaAUC", "WeightedAUC", "AdaHingeAUC",
                                 "WeightedHingeAUC"}
    # CE / InfoNCE run on stock ops (CPU fine) and match the reference fixtures
    pos, neg = torch.randn(6, 1), torch.randn(18, 1)
    np.testing.assert_allclose(loss.ce_loss(pos, neg).item(), O.LOSSES["ce"](pos, neg).item(), rtol=1e-6)
    np.testing.assert_allclose(loss.info_nce_loss(pos, neg, 3).item(), O.LOSSES["info_nce"](pos, neg, 3).item(), rtol=1e-6)


@pytest.mark.parametrize("year,valedges,coalesce", [(-1, True, False), (2010, True, False), (2012, False, False),
                                                    (-1, True, True), (2010, True, True)])
def test_driver_graph_prep_matches_oracle_restatement(year, valedges, coalesce):
    """train.py::prepare_graph (main.py:109-150) vs oracle.collab_graph_prep_ref on a toy collab split:
    the year filter, train+valid edges as encoder input with the reference's [valid, train] edge order
    against [train, valid] weight order, the degree-normalised pair weights, --use_coalesce."""
    import train as driver
    g = torch.Generator().manual_seed(31)
    n = 30
    tr = torch.randint(0, n, (80, 2), generator=g)
    tr[5] = tr[4]                                   # a repeated pair and a reversed pair: to_undirected sums them
    tr[7] = tr[6].flip(0)
    tr[9] = torch.tensor([3, 3])                    # a self loop
    va = torch.randint(0, n, (20, 2), generator=g)
    va[2] = tr[0]
    tw = torch.randint(1, 6, (80,), generator=g).float()
    vw = torch.randint(1, 6, (20,), generator=g).float()
    ty = torch.randint(2005, 2016, (80,), generator=g)

    class D:
        pass
    data = D()
    ei = torch.cat([tr.t(), tr.flip(1).t()], dim=1)
    data.adj_t = P.Graph.from_edge_index(ei, torch.cat([tw, tw]), n)
    data.edge_index = ei
    split = {"train": {"edge": tr.clone(), "weight": tw.clone(), "year": ty.clone()},
             "valid": {"edge": va.clone(), "weight": vw.clone()}}
    args = driver.argument(["--data_name=ogbl-collab", f"--year={year}", f"--use_valedges_as_input={valedges}",
                            f"--use_coalesce={coalesce}"])
    before = data.adj_t
    driver.prepare_graph(args, data, split, n)
    ref = O.collab_graph_prep_ref(tr, tw, ty, va, vw, n, year, valedges, coalesce)
    if ref["adj"] is None:
        assert data.adj_t is before
    else:
        r, c, v = data.adj_t.coo()
        dense = torch.zeros(n, n, dtype=torch.float64).index_put_((r, c), v.double(), accumulate=True)
        np.testing.assert_allclose(dense.numpy(), ref["adj"].numpy(), rtol=1e-6)
        assert set(zip(data.edge_index[0].tolist(), data.edge_index[1].tolist())) == ref["edge_index_keys"]
        assert data.edge_index.size(1) == len(ref["edge_index_keys"])           # coalesced: no duplicates
    np.testing.assert_array_equal(split["train"]["edge"].numpy(), ref["train_edge"].numpy())      # order too
    np.testing.assert_allclose(split["train"]["weight"].numpy(), ref["train_weight"].numpy(), rtol=2e-6)
    if year > 0:
        np.testing.assert_array_equal(split["train"]["year"].numpy(), ref["train_year"].numpy())


def test_hub_chunk_threshold_follows_the_source_size(monkeypatch):
    """ops.split_threshold: 128-edge chunks while the gathered matrix has up to 2^20 rows (cache-resident graphs),
    1024 beyond; PLNLP_SPLIT_THRESHOLD (ops.SPLIT_THRESHOLD) pins it"""
    from plnlp_amd import ops
    monkeypatch.setattr(ops, "SPLIT_THRESHOLD", 0)
    assert ops.split_threshold(4267) == 128 and ops.split_threshold(1 << 20) == 128
    assert ops.split_threshold((1 << 20) + 1) == 1024 and ops.split_threshold(50_000_000) == 1024
    monkeypatch.setattr(ops, "SPLIT_THRESHOLD", 256)
    assert ops.split_threshold(100) == 256 and ops.split_threshold(10 ** 8) == 256


def test_fused_adam_state_only_for_plain_adam():
    """optim.fused_adam_state: the state an update applied outside optimizer.step() needs -- created on first use,
    step reported as the count AFTER the pending update, refused when the group decays weights"""
    from plnlp_amd.optim import FusedAdam, fused_adam_state
    p, q = torch.nn.Parameter(torch.zeros(4, 3)), torch.nn.Parameter(torch.zeros(2))
    opt = FusedAdam([{"params": [p], "lr": 0.02}, {"params": [q], "weight_decay": 0.1}], lr=0.01)
    st = fused_adam_state(opt, p)
    assert st["step"] == 1 and st["lr"] == 0.02 and st["betas"] == (0.9, 0.999) and st["exp_avg"].shape == p.shape
    assert st["exp_avg"] is opt.state[p]["exp_avg"] and opt.state[p]["step"] == 0      # the caller advances it
    opt.state[p]["step"] = 7
    assert fused_adam_state(opt, p)["step"] == 8
    assert fused_adam_state(opt, q) is None
    assert fused_adam_state(opt, torch.nn.Parameter(torch.zeros(1))) is None           # not one of the optimiser's


def test_prepared_edges_must_match_the_step():
    """BaseModel.prepare_edges / train_step(prepared=): a handle prepared for another batch shape is refused, the
    right one gives the same loss as building the structures inside the step (CPU, oracle modules injected)"""
    n, h = 60, 8
    gen = torch.Generator().manual_seed(3)
    row, col = torch.randint(0, n, (300,), generator=gen), torch.randint(0, n, (300,), generator=gen)

    class D:
        pass
    data = D()
    data.adj_t = P.Graph.from_coo(row, col, None, n, n)

    def make():
        torch.manual_seed(5)
        enc, pred = O.GNNRef("SAGE", h, h, h, 1, 0.0), O.DotPredictorRef()
        m = P.BaseModel(lr=0.01, dropout=0.0, grad_clip_norm=1.0, gnn_num_layers=1, mlp_num_layers=2,
                        emb_hidden_channels=h, gnn_hidden_channels=h, mlp_hidden_channels=h, num_nodes=n,
                        num_node_feats=0, gnn_encoder_name="SAGE", predictor_name="DOT", loss_func="AUC",
                        optimizer_name="Adam", device="cpu", use_node_feats=False, train_node_emb=True,
                        modules=(_CsrEncoder(enc), pred, lambda p_, n_, k_, w_: O.LOSSES[O.select_loss("AUC", False)](p_, n_, k_, w_)))
        m.param_init()
        return m
    pos = torch.randint(0, n, (16, 2), generator=gen)
    neg = torch.randint(0, n, (16, 2, 2), generator=gen)
    a, b = make(), make()
    la = a.train_step(data, pos, neg, 2)
    lb = b.train_step(data, pos, neg, 2, prepared=b.prepare_edges(pos, neg))
    assert float(la) == float(lb)
    with pytest.raises(ValueError):
        b.train_step(data, pos, neg, 2, prepared=b.prepare_edges(pos[:8], neg[:8]))


class _CsrEncoder(torch.nn.Module):
    """oracle encoder behind the (x, adj_t) call BaseModel makes"""

    def __init__(self, ref):
        super().__init__()
        self.ref = ref

    def reset_parameters(self):
        self.ref.reset_parameters()

    def forward(self, x, adj):
        return self.ref(x, O.CSR(adj.rowptr, adj.col.to(torch.int64), None, adj.n_cols))


# ---------------------------------------------- the streamed DataLoader permutation (host entry points) ----
def test_host_randperm_entry_points_reproduce_torch_randperm_bit_for_bit():
    """plnlp_host_randperm_init / _advance (HOST functions of the C-ABI library: no GPU involved) == torch.randperm on
    a seeded CPU generator -- what DataLoader(range(n), B, shuffle=True) permutes with (model.py:147, fixture G6) --
    for several (seed, n), whole and in slices; after advance(.., to) the prefix [0, to) is already final."""
    from plnlp_amd import _lib as L
    lib = L.load()
    for seed, n, cuts in ((123, 10, [4, 7]), (0, 1, []), (5, 2, [1]), (2 ** 40 + 77, 100_003, [1, 4096, 65_536]),
                          (0x7FFF_FFFF_FFFF_FFFF, 300_000, [65_536, 131_072, 299_999])):
        want = torch.randperm(n, generator=torch.Generator().manual_seed(seed))
        perm = torch.empty(n, dtype=torch.int64)
        mt = torch.empty(625, dtype=torch.int32)
        L.check(lib.plnlp_host_randperm_init(seed, n, perm.data_ptr(), mt.data_ptr()), "init")
        assert torch.equal(perm, torch.arange(n))
        pos = 0
        for to in cuts + [n]:
            L.check(lib.plnlp_host_randperm_advance(n, perm.data_ptr(), mt.data_ptr(), pos, to), "advance")
            assert torch.equal(perm[:to], want[:to]), (seed, n, to)          # the prefix is final
            pos = to
        assert torch.equal(perm, want)
    # argument validation (nothing is touched)
    mt = torch.empty(625, dtype=torch.int32)
    assert lib.plnlp_host_randperm_init(1, -1, None, mt.data_ptr()) == -2
    assert lib.plnlp_host_randperm_init(1, 0xFFFFFFFF // 20, None, mt.data_ptr()) == -4
    assert lib.plnlp_host_randperm_advance(10, None, mt.data_ptr(), 5, 3) == -2
    assert lib.plnlp_host_randperm_advance(10, None, mt.data_ptr(), 0, 10) == -1


def test_host_randperm_matches_the_dataloader_fixture(golden):
    """through the reference's own fixture (G6: batches DataLoader(range(n), B, shuffle=True) produced under
    torch.manual_seed): the two default-generator draws of utils.batch_permutation, then the native shuffle"""
    from plnlp_amd import _lib as L
    lib = L.load()
    g6 = golden("g6_dataloader")
    for seed, n, B in [(123, 10, 4), (5, 1000, 64), (77, 65, 65), (8, 3, 10)]:
        torch.manual_seed(seed)
        torch.empty((), dtype=torch.int64).random_()
        s2 = int(torch.empty((), dtype=torch.int64).random_().item())
        perm = torch.empty(n, dtype=torch.int64)
        mt = torch.empty(625, dtype=torch.int32)
        L.check(lib.plnlp_host_randperm_init(s2, n, perm.data_ptr(), mt.data_ptr()), "init")
        L.check(lib.plnlp_host_randperm_advance(n, perm.data_ptr(), mt.data_ptr(), 0, n), "advance")
        np.testing.assert_array_equal(perm.numpy(), g6[f"s{seed}_n{n}_B{B}_perm"])
        after = torch.randint(0, 1 << 30, (4,))
        np.testing.assert_array_equal(after.numpy(), g6[f"s{seed}_n{n}_B{B}_after"])     # same RNG consumption


def test_source_ordered_split_tables_cover_every_long_row_entry_once():
    """graph.SourceOrderedSplit (explicit chunks of plnlp_row_split): every entry of a long row in exactly one
    chunk, no entry of a short row in any; a chunk stays inside one source range and under max_len; its workspace
    slot lies in its row's slot range, every slot is used once; processing order is range-major; slot order within a
    row is position order (the finalize pass adds the partial sums in slot order)"""
    import torch
    from plnlp_amd.graph import SourceOrderedSplit
    torch.manual_seed(0)
    n, thr, part, max_len = 300, 64, 32, 4
    rows, cols = [], []
    for r in range(n):
        d = 200 if r % 50 == 0 else (64 if r == 7 else 5)          # row 7 sits exactly AT the threshold: short
        c = torch.randperm(n)[:d].sort().values
        rows += [r] * d
        cols += c.tolist()
    rows = torch.tensor(rows)
    col = torch.tensor(cols, dtype=torch.int32)
    rowptr = torch.zeros(n + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0)
    sp = SourceOrderedSplit(rowptr, col, thr, part_rows=part, max_len=max_len)
    assert sp.n_long == 6 and sp.active
    cover = torch.zeros(col.numel(), dtype=torch.int32)
    first_of_slot = {}
    for c in range(sp.n_chunks):
        b, l, slot = int(sp.seg_beg[c]), int(sp.seg_len[c]), int(sp.seg_slot[c])
        assert 0 < l <= max_len
        cover[b:b + l] += 1
        ranges = col[b:b + l] // part
        assert bool((ranges == ranges[0]).all())
        row = int(torch.searchsorted(rowptr, torch.tensor(b), right=True)) - 1
        li = int((sp.long_rows == row).nonzero())
        assert int(sp.chunk_beg[li]) <= slot < int(sp.chunk_beg[li]) + int(sp.chunk_cnt[li])
        first_of_slot[slot] = b
    deg = rowptr[1:] - rowptr[:-1]
    for r in range(n):
        assert bool((cover[rowptr[r]:rowptr[r + 1]] == (1 if deg[r] > thr else 0)).all())
    assert sorted(first_of_slot) == list(range(sp.n_chunks))
    starts = [first_of_slot[s] for s in range(sp.n_chunks)]
    assert starts == sorted(starts)                                  # slot order == position order
    order = [int(col[int(b)] // part) for b in sp.seg_beg]
    assert order == sorted(order)                                    # range-major processing order
    # unsorted columns inside a row: still a cover (more, shorter chunks)
    perm_col = col.clone()
    seg = slice(int(rowptr[50]), int(rowptr[51]))
    perm_col[seg] = col[seg][torch.randperm(200)]
    sp2 = SourceOrderedSplit(rowptr, perm_col, thr, part_rows=part, max_len=max_len)
    cover = torch.zeros(col.numel(), dtype=torch.int32)
    for c in range(sp2.n_chunks):
        b, l = int(sp2.seg_beg[c]), int(sp2.seg_len[c])
        cover[b:b + l] += 1
    assert bool((cover[seg] == 1).all()) and sp2.n_chunks > sp.n_chunks
    # no long row at all
    sp3 = SourceOrderedSplit(rowptr, col, 1000)
    assert not sp3.active


def test_aggregation_form_helpers():
    """host-side form bookkeeping of the aggregation tuner (no GPU): the range rule of the source-ordered hub split,
    the form a mapped launch runs, and the words the bench line prints for a form"""
    from plnlp_amd import ops, _lib as L
    old = dict(ops.HUB_RANGES)
    try:
        ops.HUB_RANGES.update(part_rows=65536, max_len=256, max_ranges=8)
        assert ops.hub_ranges(235868) == (65536, 256)                  # collab: 4 ranges of the floor size
        assert ops.hub_ranges(12_500_000) == (1_562_500, 256)          # R-MAT x0.25: capped at 8 ranges
        assert ops.hub_ranges(10) == (65536, 256)
    finally:
        ops.HUB_RANGES.update(old)
    hub = L.AGG_HUB_XCD | ops.AGG_HUB_RANGES
    assert ops.mapped_form(hub) == 0 and ops.mapped_form(L.AGG_SLABS_128) == L.AGG_SLABS_128
    assert ops.mapped_form(hub | L.AGG_FUSED_PASSES) == L.AGG_FUSED_PASSES
    assert "source range" in ops.describe_form(hub) and "128-column slab" in ops.describe_form(L.AGG_SLABS_128)
    assert "position chunks" in ops.describe_form(0) and "pinned" in ops.describe_form(L.AGG_HUB_XCD)
    # the host-side bit never reaches the library's flag word
    assert ops.AGG_HUB_RANGES > 0xFFFF and (hub & 0xFFFF) == L.AGG_HUB_XCD


def test_switch_tables_match_the_code():
    """INTEGRATION.md section 5 is generated from plnlp_amd/switches.py, and that table is the code's: every PLNLP_* variable
    the host reads from the environment is listed, nothing listed is unread, every in-process switch named exists with the
    default stated, and no document or comment offers an environment variable that nothing reads (a dozen
    documented switches had silently stopped working)."""
    import re
    import importlib
    from plnlp_amd import switches
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    assert switches.BEGIN in doc and switches.END in doc
    section = doc[doc.index(switches.BEGIN):doc.index(switches.END) + len(switches.END)]
    assert section == switches.markdown(), "regenerate: python -m plnlp_amd.switches (INTEGRATION.md section 5)"
    # environment reads in the product and the bench
    read = set()
    files = [os.path.join(root, "plnlp_amd", f) for f in os.listdir(os.path.join(root, "plnlp_amd")) if f.endswith(".py")]
    files += [os.path.join(root, "bench.py"), os.path.join(root, "train.py")]
    for path in files:
        if path.endswith("switches.py") or path.endswith("build.py"):
            continue
        src = open(path).read()
        read |= set(re.findall(r"environ(?:\.get\(|\[)\s*[\"'](PLNLP_[A-Z0-9_]+)[\"']", src))
    listed = {row[0] for row in switches.ENV}
    assert read == listed, (sorted(read - listed), sorted(listed - read))
    # the in-process switches exist and hold the defaults the table states
    for attr, default, _ in switches.MODULE:
        for part, want in zip(attr.split(" / "), default.split(" / ")):
            mod_name, rest = part.split(".", 1) if not part.startswith("[") else (last_mod, last_name + part)
            m = re.match(r"([A-Z_]+)(?:\['([a-z_]+)'\])?$", rest)
            assert m, part
            last_mod, last_name = mod_name, m.group(1)
            obj = getattr(importlib.import_module("plnlp_amd." + mod_name), m.group(1))
            val = obj[m.group(2)] if m.group(2) else obj
            assert str(val) == want, (part, val, want)
    # nowhere else: a PLNLP_* name in the documents / host comments is an ABI constant of the header, a build macro, or listed
    header = set(re.findall(r"PLNLP_[A-Z0-9_]+", open(os.path.join(root, "include", "plnlp_hip.h")).read()))
    build_macros = {"PLNLP_GEMM_BK", "PLNLP_GEMM_X3"}
    for path in files + [os.path.join(root, "INTEGRATION.md"), os.path.join(root, "README.md")]:
        for name in set(re.findall(r"PLNLP_[A-Z0-9_]+", open(path).read())):
            name = name.rstrip("_")
            assert name in header or name in listed or name in build_macros or any(h.startswith(name) for h in header), \
                f"{os.path.basename(path)} names {name}: not an environment switch the code reads (plnlp_amd/switches.py)"


def test_pinned_aggregation_forms_are_looked_up_not_measured(tmp_path):
    """ops.AGG_FORMS: a (graph shape, width) found in the shipped table is not measured -- two boxes then run the same
    kernels and give the same bits; the key is the same for a graph and its transposed view"""
    import json
    import plnlp_amd as P
    from plnlp_amd import ops
    g = torch.Generator().manual_seed(1)
    row, col = torch.randint(0, 50, (400,), generator=g), torch.randint(0, 70, (400,), generator=g)
    graph = P.Graph.from_coo(row, col, None, 50, 70)
    key = ops.agg_form_key(graph, 256)
    assert key == "50:70:400:f256" and ops.agg_form_key(graph.t(), 256) == key
    path = tmp_path / "forms.json"
    path.write_text(json.dumps({"forms": {key: 16}}))
    old = dict(ops.AGG_FORMS)
    try:
        ops.AGG_FORMS.update(path=str(path), table=None, hits=0)
        assert ops.pinned_agg_form(graph, 256) == 16 and ops.pinned_agg_form(graph, 512) is None
        # the measurement entry point returns the pinned form without launching anything (CPU tensors would raise)
        x = torch.zeros(70, 256)
        assert ops._time_agg_forms(graph, x, torch.zeros(50, 256), "mean", False, None, None) == 16
        ops.AGG_FORMS.update(path="none", table=None)
        assert ops.pinned_agg_form(graph, 256) is None
    finally:
        ops.AGG_FORMS.clear()
        ops.AGG_FORMS.update(old)
    # the shipped table parses and every form in it is a combination of known flag bits
    shipped = json.load(open(os.path.join(os.path.dirname(ops.__file__), "agg_forms.json")))
    known = (P._lib.AGG_SLABS_128 | P._lib.AGG_SLABS_256 | P._lib.AGG_SLABS_XCD | P._lib.AGG_HUB_XCD | ops.AGG_HUB_RANGES)
    for k, v in shipped["forms"].items():
        assert re_key(k) and int(v) & ~known == 0, (k, v)


def re_key(k):
    import re
    return re.match(r"\d+:\d+:\d+:f\d+$", k) is not None


@pytest.mark.parametrize("name,ext", [("ogbl-collab", "pt"), ("ogbl-ddi", "npz"), ("ogbl-citation2", "pt")])
def test_data_path_reads_ogbs_raw_layout(tmp_path, name, ext):
    """train.py --data_path (main.py:74-95): a 200-node dataset written in OGB's on-disk layout (raw/edge.csv.gz,
    raw/edge_weight.csv.gz, raw/node-feat.csv.gz, split/<type>/{train,valid,test}.pt of numpy arrays) is read back without
    the ogb wheel into what main.py holds after line 95 -- inverse edges added for the undirected datasets with their
    per-edge attributes, adj_t = the transposed adjacency carrying edge_weight, edge_index rebuilt from it, the split
    dictionary as tensors -- and the driver falls back to the synthetic stand-in when the directory is absent."""
    import train as driver
    from plnlp_amd import ogb_raw
    gen = torch.Generator().manual_seed(7)
    n, e = 200, 900
    edge = torch.randint(0, n, (e, 2), generator=gen)
    edge = edge[edge[:, 0] != edge[:, 1]]
    e = edge.shape[0]
    weight = torch.randint(1, 6, (e,), generator=gen).float() if name == "ogbl-collab" else None
    year = torch.randint(2000, 2018, (e,), generator=gen) if name == "ogbl-collab" else None
    x = torch.randn(n, 16, generator=gen) if name == "ogbl-citation2" else None
    cut = [0, e * 8 // 10, e * 9 // 10, e]
    split = {}
    for i, s in enumerate(("train", "valid", "test")):
        part = edge[cut[i]:cut[i + 1]]
        if name == "ogbl-citation2":
            split[s] = {"source_node": part[:, 0].clone(), "target_node": part[:, 1].clone()}
            if s != "train":
                split[s]["target_node_neg"] = torch.randint(0, n, (part.shape[0], 10), generator=gen)
        else:
            split[s] = {"edge": part.clone()}
            if s != "train":
                split[s]["edge_neg"] = torch.randint(0, n, (50, 2), generator=gen)
            if weight is not None:
                split[s]["weight"] = weight[cut[i]:cut[i + 1]].clone()
                split[s]["year"] = year[cut[i]:cut[i + 1]].clone()
    root = str(tmp_path / "dataset")
    assert not ogb_raw.available(name, root)
    ogb_raw.write_link_dataset(name, root, edge, n, split, edge_weight=weight, edge_year=year, x=x, split_ext=ext)
    assert ogb_raw.available(name, root) and os.path.isdir(os.path.join(root, name.replace("-", "_"), "raw"))
    data, split_r, num_nodes = ogb_raw.read_link_dataset(name, root)
    assert num_nodes == n and data.num_nodes == n
    for s in split:
        assert set(split_r[s]) == set(split[s])
        for k in split[s]:
            assert torch.equal(split_r[s][k], split[s][k]), (s, k)
    # the graph: what T.ToSparseTensor() makes of the (inverse-augmented) edge list -- checked against dense matrices
    undirected = name != "ogbl-citation2"
    ei = torch.cat([edge.t(), edge.t().flip(0)], 1) if undirected else edge.t()
    w = None if weight is None else (torch.cat([weight, weight]) if undirected else weight)
    dense = torch.zeros(n, n, dtype=torch.float64)
    dense.index_put_((ei[1], ei[0]), torch.ones(ei.shape[1], dtype=torch.float64) if w is None else w.double(), accumulate=True)
    row, col, val = data.adj_t.coo()
    got = torch.zeros(n, n, dtype=torch.float64)
    got.index_put_((row, col), torch.ones(row.numel(), dtype=torch.float64) if val is None else val.double(), accumulate=True)
    assert torch.equal(got, dense) and row.numel() == ei.shape[1]          # duplicates kept, like SparseTensor
    assert (val is None) == (weight is None)
    assert torch.equal(data.edge_index, torch.stack([col, row]))            # main.py:82-83
    if x is not None:
        assert data.num_features == 16 and torch.allclose(data.x, x, rtol=1e-6) and data.x.dtype == torch.float32
    else:
        assert data.x is None
    assert hasattr(data, "edge_year") == (name == "ogbl-collab")
    # the driver: reads the directory when it is there, else the synthetic stand-in of the same shape
    args = driver.argument(["--data_name", name, "--data_path", root, "--data_scale", "0.002"])
    d2, s2, n2 = driver.load_dataset(args, "cpu")
    assert n2 == n and torch.equal(d2.adj_t.col, data.adj_t.col)
    args = driver.argument(["--data_name", name, "--data_path", str(tmp_path / "nowhere"), "--data_scale", "0.002"])
    d3, s3, n3 = driver.load_dataset(args, "cpu")
    assert n3 != n and "train" in s3
    # and main.py:109-150's preparation runs on what was read (collab: year filter + validation edges as input)
    if name == "ogbl-collab":
        args = driver.argument(["--data_name", name, "--year", "2008", "--use_valedges_as_input", "True"])
        driver.prepare_graph(args, data, split_r, n)
        assert int(split_r["train"]["year"].min()) >= 2008 or split_r["train"]["year"].numel() == 0
        assert split_r["train"]["edge"].shape[0] == split_r["train"]["weight"].shape[0]
    if name == "ogbl-citation2":
        args = driver.argument(["--data_name", name])
        driver.prepare_graph(args, data, split_r, n)
        r2, c2, _ = data.adj_t.coo()
        sym = torch.zeros(n, n, dtype=torch.bool)
        sym[r2, c2] = True
        assert torch.equal(sym, sym.t())


def test_emb_pad_granule_is_validated():
    """PLNLP_EMB_PAD: 0 used to divide by zero in every optimiser step, 6 built a 'padded' table whose rows are not
    16-byte aligned and handed it to the vector kernels"""
    from plnlp_amd import ops
    assert ops._emb_pad_floats("4") == 4 and ops._emb_pad_floats("16") == 16 and ops._emb_pad_floats("64") == 64
    for bad in ("0", "1", "2", "6", "-4", "18"):
        with pytest.raises(ValueError):
            ops._emb_pad_floats(bad)
    with pytest.raises(ValueError):
        ops._emb_pad_floats("four")


def test_kernel_form_switches_reject_unknown_values():
    """PLNLP_GEMM_BLOCK / PLNLP_EDGE_SEGMENT: a misspelt value is an error that names the choices, before the library is touched"""
    from plnlp_amd import ops
    for table, key, apply in ((ops.GEMM_BLOCK, "mode", ops._apply_gemm_block), (ops.EDGE_SEGMENT, "form", ops._apply_edge_segment)):
        old = table[key]
        try:
            table[key] = "fastest"
            with pytest.raises(ValueError, match="one of"):
                apply()
        finally:
            table[key] = old
