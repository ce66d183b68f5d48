"""GPU parity, fifth batch (round 5):

  * trained-regime parity THROUGH THE KERNELS THE BENCHMARK RUNS: the two recipes at their own widths (h = 256 / 512) on
    graphs big enough that every multi-step run goes through the stationary-weights split-bf16 GEMM, the F = 256 / 512
    aggregation forms with their fused hub-chunk pass, the touched-rows forward -- asserted with the library's launch
    counters (plnlp_launch_counts), not assumed;
  * a TEACHER-FORCED epoch of the ddi recipe: the HIP state is reset to the oracle's before every step, so each step is
    compared alone -- which separates "Adam's sign lottery on round-off-sized gradients" (the free-running epoch's few
    per-cent) from a difference in logic (none: every step agrees to 1e-5);
  * the public create_input_feat is the real matrix, zero-row launches are defined, the launch counters count.
Same rules as tests/test_hip_parity.py: through the C ABI, fp32 tolerance 1e-5 relative, integer outputs bit-exact."""
import os

import numpy as np
import pytest
import torch

import oracle as O
from gpu_util import close, dev, rand_csr, to_graph

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def P():
    import plnlp_amd
    from plnlp_amd import _lib
    _lib.load()                      # no library -> the GPU suite must fail, not skip
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return plnlp_amd


def _delta(P, before):
    after = P.ops.launch_counts()
    return {k: after[k] - before[k] for k in after}


def _stationary(d):
    """launches of the stationary-weights split-bf16 product: the 128-row kernel (gemm_x3s.hip) + the whole-block one (gemm_x3b.hip)"""
    return d["gemm_x3s"] + d["gemm_x3b"]


# ------------------------------------------------------------------------- launch counters ----
def test_launch_counters_name_the_kernel_family_that_ran(P):
    """plnlp_launch_counts: a 20 000-row split-bf16 product with K-contiguous A runs the stationary-weights kernel, 2 000
    rows the tile kernel, the f32 form the f32 tile kernel, a weight gradient cuts K and reduces the slices (the tile kernel below
    32 768 rows, the whole-block kernel from there on); a graph with
    a hub row runs the fused main + chunk pass, one without runs the plain one-wave-per-row launch.  And the host's
    question (plnlp_gemm_stationary_applies) has the launch's answer."""
    ops = P.ops
    old = ops.GEMM_MATH["mode"]
    try:
        w = torch.randn(256, 256, device="cuda")
        for rows, math, want in ((20000, "bf16x3", "gemm_x3s"), (2000, "bf16x3", "gemm_tile_x3"), (20000, "f32", "gemm_tile_f32")):
            ops.GEMM_MATH["mode"] = math
            a = torch.randn(rows, 256, device="cuda")
            c0 = ops.launch_counts()
            ops.gemm([(a, w)], False, True)
            d = _delta(P, c0)
            # (the 2 000-row product has few tiles: it also cuts K and reduces the slices)
            assert d[want] == 1 and sum(v for k, v in d.items() if k != "gemm_splitk_reduce") == 1, (rows, math, d)
        ops.GEMM_MATH["mode"] = "bf16x3"
        a = torch.randn(40000, 256, device="cuda")
        for rows, want in ((20000, "gemm_tile_x3"), (40000, "gemm_wgrad_wide")):
            c0 = ops.launch_counts()
            ops.gemm([(a[:rows], a[:rows])], True, False)    # [256, 256] = a^T a: split-K; from 32 768 rows on the whole-block kernel
            d = _delta(P, c0)
            assert d[want] == 1 and d["gemm_splitk_reduce"] == 1 and _stationary(d) == 0, (rows, d)
            assert d["gemm_tile_x3"] + d["gemm_wgrad_wide"] == 1, (rows, d)
        # an unaligned result (leading dimension 257) declines the stationary form -- and keeps its split-K
        out = torch.empty(20000, 257, device="cuda")[:, :256]
        c0 = ops.launch_counts()
        ops.gemm([(torch.randn(20000, 256, device="cuda"), w)], False, True, out=out)
        assert _delta(P, c0)["gemm_tile_x3"] == 1
    finally:
        ops.GEMM_MATH["mode"] = old
    hub = to_graph(P, rand_csr(3000, 20000, 5, weighted=False, hub=900))
    flat = to_graph(P, rand_csr(3000, 20000, 5, weighted=False))
    x = torch.randn(3000, 256, device="cuda")
    c0 = ops.launch_counts()
    ops.csr_aggregate(hub, x, "mean", False, tune=0)
    d = _delta(P, c0)
    assert d["agg_fused"] == 1 and d["agg_finalize"] == 1 and d["agg_vec"] == 0, d
    c0 = ops.launch_counts()
    ops.csr_aggregate(flat, x, "mean", False, tune=0)
    d = _delta(P, c0)
    assert d["agg_vec"] == 1 and d["agg_fused"] == 0 and d["agg_chunk"] == 0, d


def test_zero_row_launches_are_defined(P):
    """an empty result has no storage (its pointer is NULL): the entry points return success before they look at it.  A rank
    of a row-sharded step whose block no edge of the batch touches used to fail here (tests/test_hip_multirank.py)."""
    ops = P.ops
    g = to_graph(P, rand_csr(500, 3000, 3, weighted=False))
    x = torch.randn(500, 64, device="cuda")
    rows = torch.empty(0, dtype=torch.int32, device="cuda")
    out_map = torch.full((500,), -1, dtype=torch.int32, device="cuda")
    assert ops.csr_aggregate(g, x, "mean", False, row_index=rows, out_map=out_map).shape == (0, 64)
    assert ops.gemm([(torch.empty(0, 64, device="cuda"), torch.randn(32, 64, device="cuda"))], False, True).shape == (0, 32)
    e = torch.empty(0, dtype=torch.int64, device="cuda")
    assert ops.edge_dot_fwd(x, e, e).numel() == 0 and ops.edge_hadamard_fwd(x, e, e).shape == (0, 64)


# --------------------------------------------------- the 1-output head's backward in one pass ----
@pytest.mark.parametrize("rows,feat,p", [(5000, 512, 0.3), (777, 64, 0.0), (262144, 512, 0.3), (33, 200, 0.5)])
def test_fused_head_backward_is_the_four_separate_passes(P, rows, feat, p):
    """plnlp_mlp_head_backward_f32 (MLPPredictor's last linear behind relu + dropout, layer.py:82-86): dz, the head's weight
    gradient and the hidden layer's bias gradient are the BITS of the separate passes (outer product with the gate
    epilogue, plnlp_colsum_f32 with / without row weights); the head's bias gradient -- a sum of `rows` numbers in another
    association -- to round-off; the four against float64.  Then through MLPStackFn: same gradients with the fusion on
    and off (ddi's full scorer shape among the cases)."""
    ops = P.ops
    gen = torch.Generator().manual_seed(rows + feat)
    z = torch.randn(rows, feat, generator=gen)
    keep = (torch.rand(rows, feat, generator=gen) >= p).float() / (1.0 - p)
    a = (torch.relu(z) * keep).cuda()                     # what the forward stored: dropout(relu(z))
    g = torch.randn(rows, generator=gen).cuda()
    w = (torch.randn(1, feat, generator=gen) * 0.1).cuda()
    scale = 1.0 / (1.0 - p)
    dz, dw, dbp, db = ops.mlp_head_backward(a, g, w, scale)
    from plnlp_amd import _lib
    want_dz = ops.outer(g, w, epilogue=_lib.make_epilogue(gate=a, gate_scale=scale))
    assert torch.equal(dz, want_dz)
    assert torch.equal(dw, ops.colsum(a, row_weight=g).reshape(1, -1))
    assert torch.equal(dbp, ops.colsum(want_dz))
    a64, g64, w64 = a.double().cpu(), g.double().cpu(), w.double().cpu()
    dz64 = (g64[:, None] * w64) * (a64 > 0) * scale
    close(dz, dz64, rtol=2e-6)
    close(dw.reshape(-1), (g64[:, None] * a64).sum(0), rtol=1e-5)
    close(dbp, dz64.sum(0), rtol=1e-5, atol=1e-5 * float(dz64.abs().sum(0).max()))
    assert torch.equal(db, ops.colsum(g.reshape(-1, 1)))
    assert abs(float(db) - float(g64.sum())) <= 1e-5 * float(g64.abs().sum())
    # the stack: fusion on == fusion off
    if rows <= 5000:
        lin1 = torch.nn.Linear(feat, feat).cuda()
        x = torch.randn(rows, feat, generator=gen).cuda().requires_grad_(True)
        grads = {}
        for on in (True, False):
            ops.FUSE_HEAD_BACKWARD["enabled"] = on
            try:
                for t in (x, lin1.weight, lin1.bias, w):
                    t.grad = None
                wl = w.detach().clone().requires_grad_(True)
                bl = torch.zeros(1, device="cuda", requires_grad=True)
                ops.manual_seed(5)
                out = ops.MLPStackFn.apply(x, p, True, lin1.weight, lin1.bias, wl, bl)
                out.backward(g.reshape(-1, 1))
                grads[on] = [t.grad.clone() for t in (x, lin1.weight, lin1.bias, wl, bl)]
            finally:
                ops.FUSE_HEAD_BACKWARD["enabled"] = True
        for i_, (u, v) in enumerate(zip(grads[True], grads[False])):
            assert torch.equal(u, v), i_


@pytest.mark.parametrize("rows,feat,p", [(20000, 512, 0.3), (40000, 256, 0.0), (262144, 512, 0.3), (17000, 384, 0.2)])
def test_head_in_the_hidden_products_epilogue(P, rows, feat, p):
    """PLNLP_EPI_ROWDOT: MLPPredictor's 1-output head (layer.py:86) evaluated in the epilogue of the hidden layer's product on
    the stationary-weights kernel -- the hidden activation is the bits of the plain launch; the scores equal the separate
    pass (plnlp_matvec_f32 over the stored activation) to fp32 round-off of a `feat`-term dot product and float64 at 1e-5;
    the launch counters show no matvec-sized second product; through MLPStackFn the output and every gradient agree with
    the fusion off (the backward never sees the difference: it reads the stored activation)."""
    ops = P.ops
    gen = torch.Generator().manual_seed(rows + feat)
    x = torch.randn(rows, feat, generator=gen).cuda()
    w1 = (torch.randn(feat, feat, generator=gen) / feat ** 0.5).cuda()
    b1 = (torch.randn(feat, generator=gen) * 0.1).cuda()
    w2 = (torch.randn(1, feat, generator=gen) / feat ** 0.5).cuda()
    b2 = torch.tensor([0.25]).cuda()
    from plnlp_amd import _lib
    epi = lambda: _lib.make_epilogue(bias=b1, relu=True, dropout_p=p, dropout_seed=77)
    c0 = ops.launch_counts()
    hid, score = ops.gemm([(x, w1)], False, True, epilogue=epi(), rowdot=(w2, b2))
    d = _delta(P, c0)
    assert score is not None and _stationary(d) == 1 and sum(d.values()) == 1, d
    plain = ops.gemm([(x, w1)], False, True, epilogue=epi())
    assert torch.equal(hid, plain)
    want = ops.matvec(plain, w2, b2)
    ref64 = plain.double().cpu() @ w2.double().cpu().reshape(-1) + 0.25
    mag = (plain.double().cpu().abs() @ w2.double().cpu().abs().reshape(-1)) + 0.25
    assert float(((score.reshape(-1).double().cpu() - ref64).abs() / mag).max()) <= 2e-6
    assert float(((score.reshape(-1) - want).abs().double().cpu() / mag).max()) <= 2e-6
    if rows <= 40000:
        g = torch.randn(rows, 1, generator=gen).cuda()
        outs = {}
        for on in (True, False):
            ops.FUSE_HEAD_FORWARD["enabled"] = on
            try:
                leaves = [t.detach().clone().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
                ops.manual_seed(5)
                c0 = ops.launch_counts()
                out = ops.MLPStackFn.apply(leaves[0], p, True, *leaves[1:])
                out.backward(g)
                outs[on] = (out.detach(), [t.grad.clone() for t in leaves])
            finally:
                ops.FUSE_HEAD_FORWARD["enabled"] = True
        close(outs[True][0], outs[False][0], rtol=1e-5, atol=2e-6 * float(mag.max()))
        for u, v in zip(outs[True][1], outs[False][1]):
            assert torch.equal(u, v)          # the backward reads g and the stored activation: identical either way
    # a shape the form does not cover (too few rows for the stationary kernel): (C, None), the caller makes its own pass
    small, none = ops.gemm([(x[:2000], w1)], False, True, epilogue=epi(), rowdot=(w2, b2))
    assert none is None
    close(small, plain[:2000], rtol=1e-5)         # (the tile kernel cuts K here: another association of the same sums)


# ------------------------------------------------------------------ create_input_feat ----
def test_public_create_input_feat_is_the_real_matrix_with_its_gradient(P):
    """BaseModel.create_input_feat mirrors model.py:98-105: a caller gets torch.cat([emb.weight, data.x], -1) -- current
    values, an autograd edge to emb.weight -- whatever the trainer's own passes defer internally (a first GCNConv takes
    the parts and the per-step copy is skipped: that form stays private, BaseModel._input_feat)."""
    n, e, f, h = 900, 40, 18, 64
    m = P.BaseModel(lr=0.01, dropout=0.0, grad_clip_norm=1.0, gnn_num_layers=2, mlp_num_layers=2, emb_hidden_channels=e,
                    gnn_hidden_channels=h, mlp_hidden_channels=h, num_nodes=n, num_node_feats=f, gnn_encoder_name="GCN",
                    predictor_name="MLP", loss_func="AUC", optimizer_name="Adam", device="cuda", use_node_feats=True,
                    train_node_emb=True)
    m.param_init()

    class D:
        pass
    data = D()
    data.x = torch.randn(n, f, device="cuda")
    data.adj_t = P.gcn_normalization(to_graph(P, rand_csr(n, 6000, 9, weighted=False)))
    x1 = m.create_input_feat(data)
    assert not getattr(x1, "_plnlp_stale", False)
    assert torch.equal(x1.detach(), torch.cat([m.emb.weight.detach(), data.x], -1))
    with torch.no_grad():
        m.emb.weight.add_(0.5)                                   # the table moves on (an optimiser step) ...
    x2 = m.create_input_feat(data)                               # ... and the next call sees it
    assert torch.equal(x2.detach(), torch.cat([m.emb.weight.detach(), data.x], -1))
    (x2[:, :e].float() * 2.0).sum().backward()                   # sliced, cast, consumed by foreign ops: the gradient arrives
    assert torch.equal(m.emb.weight.grad, torch.full_like(m.emb.weight, 2.0))
    # the trainer's own pass may defer; a foreign consumer of THAT tensor is not part of the contract, the encoder is
    xi = m._input_feat(data)
    assert getattr(xi, "_plnlp_stale", False)
    m.encoder.eval()
    with torch.no_grad():
        assert torch.equal(m.encoder(xi, data.adj_t), m.encoder(m.create_input_feat(data), data.adj_t))


def test_unaligned_embedding_table_is_kept_padded_and_trains_to_the_same_bits(P):
    """citation2's recipe (README.md:40) trains a 50-wide table next to 128 features under a GCN: BaseModel keeps such a
    table padded (ops.EMB_PAD: emb.weight = the [:, :50] view of a zero-padded [N, 64] buffer), the 64-wide
    aggregation gathers from it directly, the gradient arrives and Adam steps in that layout -- and nothing changes:
    five steps end on the bits of the model whose table is a plain contiguous [N, 50] tensor (which pays the padded copy
    and the strided -> contiguous gradient copy every step).  state_dict holds the [N, 50] parameter either way."""
    from plnlp_amd import ops
    n, e, f, h, B, k = 3000, 50, 16, 64, 512, 3
    g = to_graph(P, O.gcn_norm_csr(rand_csr(n, 20000, 21, weighted=False)))

    class D:
        pass
    data = D()
    data.adj_t = g
    data.x = torch.randn(n, f, generator=torch.Generator().manual_seed(2)).cuda()
    gen = torch.Generator().manual_seed(3)
    pos = torch.randint(0, n, (5 * B, 2), generator=gen).cuda()
    neg = torch.randint(0, n, (5 * B, k, 2), generator=gen).cuda()
    out = {}
    for padded in (True, False):
        torch.manual_seed(9)
        m = P.BaseModel(lr=0.01, dropout=0.0, grad_clip_norm=1.0, gnn_num_layers=2, mlp_num_layers=2, emb_hidden_channels=e,
                        gnn_hidden_channels=h, mlp_hidden_channels=h, num_nodes=n, num_node_feats=f, gnn_encoder_name="GCN",
                        predictor_name="MLP", loss_func="AUC", optimizer_name="Adam", device="cuda", use_node_feats=True,
                        train_node_emb=True)
        assert ops.padded_base(m.emb.weight.detach()) is not None and m.emb.weight.shape == (n, e)
        m.param_init()
        if padded:
            init = {"emb": m.emb.weight.detach().clone(), "enc": {k_: v.clone() for k_, v in m.encoder.state_dict().items()},
                    "pred": {k_: v.clone() for k_, v in m.predictor.state_dict().items()}}
        else:
            m.emb.weight.data = init["emb"].clone()                     # a plain contiguous table: the old path
            m.encoder.load_state_dict(init["enc"])
            m.predictor.load_state_dict(init["pred"])
            assert ops.padded_base(m.emb.weight.detach()) is None
        m.encoder.train()
        m.predictor.train()
        losses = [float(m.train_step(data, pos[i * B:(i + 1) * B], neg[i * B:(i + 1) * B], k)) for i in range(5)]
        torch.cuda.synchronize()
        if padded:
            base = ops.padded_base(m.emb.weight.detach())
            assert torch.equal(base[:, e:], torch.zeros_like(base[:, e:]))        # the pad columns never moved
            assert m.emb.state_dict()["weight"].shape == (n, e)
        out[padded] = (losses, m.emb.weight.detach().clone().contiguous(),
                       torch.cat([p.detach().reshape(-1) for p in list(m.encoder.parameters()) + list(m.predictor.parameters())]))
    assert out[True][0] == out[False][0]
    assert torch.equal(out[True][1], out[False][1]) and torch.equal(out[True][2], out[False][2])


# --------------------------------------------------------------- teacher-forced ddi epoch ----
def _copy_state(model, ref):
    """the oracle trainer's parameters and Adam state into the HIP model (same parameter order: encoder, predictor, emb)"""
    from plnlp_amd.optim import fused_adam_state
    with torch.no_grad():
        for p, q in zip(model.para_list, ref.params):
            p.copy_(q.detach().to(p.device))
            fused_adam_state(model.optimizer, p)                       # creates the state entry if this is the first step
            st, sq = model.optimizer.state[p], ref.optimizer.state.get(q, {})
            if sq:
                st["exp_avg"].copy_(sq["exp_avg"].to(p.device))
                st["exp_avg_sq"].copy_(sq["exp_avg_sq"].to(p.device))
                st["step"] = int(sq["step"])
            else:
                st["exp_avg"].zero_()
                st["exp_avg_sq"].zero_()
                st["step"] = 0


@pytest.mark.parametrize("recipe,math,max_steps", [("ddi", "bf16x3", 100), ("ddi", "f32", 100), ("ddi_wide", "bf16x3", 6),
                                                   ("collab_wide", "bf16x3", 6)])
def test_teacher_forced_epoch_agrees_step_by_step(P, recipe, math, max_steps):
    """The ddi recipe's free-running epoch-1 loss sits 1e-3 (median) to 2e-2 (worst seed) from the float32 oracle's -- and
    the oracle's own float32 and float64 runs sit exactly as far apart (6e-4 / 1.9e-2 over the same 48 seeds, fixture
    g11; at h = 512 the two oracles part by 4.5e-2 in the first epoch, fixture g12).  Lottery or logic?  Here the HIP model
    is RESET to the oracle's state (parameters, Adam moments, step count) before each step of the epoch and takes the
    same batch: every step's loss then agrees to 1e-5 and every weight matrix after the step to 2e-5 in its 99 % quantile --
    the steps are the reference's steps; what remains after a step is the share of elements whose gradient is round-off
    (the scorer's output bias first of all: the pairwise loss is invariant under a shift of all scores, so its true
    gradient is zero), which Adam moves by up to lr in a direction round-off decides.  That is what compounds over a
    free-running epoch.  The WIDE cases take their steps through the kernels the benchmark runs (asserted by the launch
    counters): the sharp per-step statement for the recipes at h = 256 / 512."""
    import trained_parity as T
    seed = 0
    r = T.RECIPES[recipe]
    g = T.problem(recipe)
    n, h = g["num_nodes"], T.width(recipe)
    old = P.ops.GEMM_MATH["mode"]
    P.ops.GEMM_MATH["mode"] = math
    try:
        enc, pred, emb = T.initial_modules(recipe, seed)
        adj = g["adj_t"]
        csr = O.CSR(adj.rowptr, adj.col.to(torch.int64), None, n)
        ref = O.TrainerRef(enc, pred, emb, csr, loss_name=r["loss"], lr=r["lr"], clip_norm=r["clip"])
        m = P.BaseModel(lr=r["lr"], dropout=0.0, grad_clip_norm=r["clip"], gnn_num_layers=r["layers"], mlp_num_layers=2,
                        emb_hidden_channels=h, gnn_hidden_channels=h, mlp_hidden_channels=h, num_nodes=n,
                        num_node_feats=0, gnn_encoder_name="SAGE", predictor_name=r["predictor"], loss_func=r["loss"],
                        optimizer_name="Adam", device="cuda", use_node_feats=False, train_node_emb=True)
        m.encoder.train()
        m.predictor.train()

        class D:
            pass
        data = D()
        data.adj_t = adj.to("cuda")
        torch.manual_seed(T.epoch_seed(0, seed))
        pos, w = g["train"], None
        if r["walk_length"]:
            walk = O.random_walk_ref(csr, pos.reshape(-1), r["walk_length"], T.walk_seed(0, seed))
            pos, w = O.random_walk_pairs_ref(walk, r["walk_length"])
        _, neg = O.pos_neg_edges_ref("train", {"train": {"edge": pos}}, num_nodes=n, neg_sampler_name="local", num_neg=r["k"])
        batches = O.batch_permutation(pos.size(0), r["batch"], True)[:max_steps]
        assert len(batches) >= min(6, max_steps)
        c0 = P.ops.launch_counts()
        worst_loss, worst_bulk, lottery, bias_lottery = 0.0, 0.0, [], []
        for perm in batches:
            _copy_state(m, ref)
            wb = None if w is None else w[perm]
            loss_hip = float(m.train_step(data, pos[perm].cuda(), neg[perm].cuda(), r["k"], None if wb is None else wb.cuda()))
            loss_ref = float(ref.step(pos[perm], neg[perm], r["k"], wb)[0])
            worst_loss = max(worst_loss, abs(loss_hip - loss_ref) / abs(loss_ref))
            assert abs(loss_hip - loss_ref) <= 1e-5 * abs(loss_ref), (loss_hip, loss_ref)
            # after the step, from identical state: the bulk of every tensor agrees tightly; the stragglers moved at most lr
            torch.cuda.synchronize()
            for p, q in zip(m.para_list, ref.params):
                d = (p.detach().cpu().double() - q.detach().double()).abs().flatten()
                if d.numel() >= 4096:          # the weight matrices and the table (bias vectors: the lottery's home, below)
                    worst_bulk = max(worst_bulk, float(d.kthvalue(int(0.99 * d.numel()))[0]))
                else:
                    bias_lottery.append(float((d > 1e-4).double().mean()))
                lottery.append(float((d > 1e-4).double().mean()))
                assert float(d.max()) <= 2.0 * r["lr"] + 1e-7
        dlt = _delta(P, c0)
        print(f"teacher-forced {recipe} epoch, HIP {math}: {len(batches)} steps, worst per-step loss deviation {worst_loss:.2e}, worst "
              f"99 % quantile of |weight matrix / table - oracle| after a step {worst_bulk:.2e}, share of elements moved > 1e-4 apart "
              f"in one step {np.mean(lottery):.2e} (mean over tensors and steps; bias vectors alone {np.mean(bias_lottery):.2e}); "
              f"launches {({k: v for k, v in dlt.items() if v})}")
        assert worst_bulk <= 2e-5
        if recipe in T.WIDE:
            assert _stationary(dlt) >= 2 * len(batches), dlt
            assert dlt["agg_fused"] + dlt["agg_fused_hub_xcd"] + dlt["agg_vec_slabs"] + dlt["agg_chunk"] + dlt["agg_dense"] >= len(batches), dlt
    finally:
        P.ops.GEMM_MATH["mode"] = old


# ----------------------------------- trained regime at the recipes' widths, through the default kernels ----
_wide = {}


def _wide_fixture(golden, T, recipe):
    g12 = golden("g12_trained_curves_wide")
    r = T.RECIPES[recipe]
    np.testing.assert_allclose(g12[f"{recipe}_problem"], [float(v) for v in T.PROBLEMS[r["problem"]].values()])
    np.testing.assert_allclose(g12[f"{recipe}_hyper"], [r["lr"], r["clip"], r["epochs"], r["batch"], r["walk_length"], r["k"],
                                                        float(r["decay"])])
    return (g12[f"{recipe}_f32"].astype(np.float64), g12[f"{recipe}_f64"].astype(np.float64), g12[f"{recipe}_f32_loss"],
            g12[f"{recipe}_f64_loss"])


@pytest.mark.parametrize("recipe", ["collab_wide", "ddi_wide"])
def test_trained_regime_parity_through_the_benchmarks_kernels(P, golden, recipe):
    """the trained-regime harness of round 4 runs at h = 64 on 2-3 000 nodes -- every GEMM below the 16 384
    rows from which the stationary-weights kernel (the default of all three workloads) is used, every aggregation below
    the F = 256 forms.  These legs train the recipes AT THEIR WIDTHS (README.md:35 h = 256, README.md:24 h = 512), same
    initial weights / walks / negatives / permutations as the oracle's float32 and float64 runs (fixture g12, 16 / 8 seeds):
      collab_wide  40 000-node soft geometric graph with 24 hub nodes (~1 500 neighbours: rows beyond the long-row
                   threshold), SAGE x1 + DOT, WeightedHingeAUC on one-hop walk pairs, batches of 32 768 (+ as many negatives):
                   ~96 % of the nodes touched, inside the row-sparse window, so the last conv runs at ~38 000 touched rows;
      ddi_wide     ddi's own size and density (4 267 nodes, eight communities of 500: ~450 neighbours each), SAGE x2 + MLP at
                   h = 512, 8 192 x (1 + 3) = 32 768 scorer rows per step, 30 epochs of 12 steps: long enough for every
                   converged run to sit on the 90 % Hits@20 plateau the 10 % unrankable positives leave (README.md:8's metric).
    Asserted: (a) by the launch counters, that the runs went through gemm_x3s and the fused / slab aggregation forms (ddi_wide since
    round 6: the dense-graph aggregation on the matrix cores, csrc/aggregate_dense.hip);
    (b) epoch-1 loss, paired per seed: mean deviation from the float32 oracle no larger than the oracle's own float64 run's
    (+ 1e-4); every later epoch's mean deviation within twice that gap (+ 0.2 %) -- the lottery the teacher-forced tests pin down;
    (c) collab: the final Hits@50, mean over 16 seeds, within 0.3 points + 2 s.e. (0.2) of the float32 oracle's on valid and test,
    epochs-to-level distributed like the oracle's (Mann-Whitney, two-sided, p > 0.05 / 4), reached by the same share of seeds;
    ddi: seed by seed -- see the comments at the assertions (round 6: the +-6.5-point band is gone)."""
    import trained_parity as T
    ref32, ref64, loss32, loss64 = _wide_fixture(golden, T, recipe)
    n = ref32.shape[0]
    c0 = P.ops.launch_counts()
    runs = [T.run_hip(P, recipe, s, P.ops.GEMM_MATH["mode"]) for s in range(n)]
    d = _delta(P, c0)
    hip, losses = np.stack([h for h, _ in runs]), np.stack([l for _, l in runs])
    # (a) the forms
    steps = sum(1 for _ in range(n)) * T.RECIPES[recipe]["epochs"]
    assert _stationary(d) >= 2 * steps, d                              # at least forward + data-gradient per step
    assert d["agg_fused"] + d["agg_fused_hub_xcd"] + d["agg_vec_slabs"] + d["agg_chunk"] + d["agg_dense"] > steps, d
    if recipe == "collab_wide":
        assert d["agg_fused"] + d["agg_fused_hub_xcd"] > steps, d     # hub rows: the chunk pass inside the main launch
    # (b) losses
    rel1 = np.abs(losses[:, 0] - loss32[:, 0]) / loss32[:, 0]
    gap = np.abs(loss32 - loss64) / loss64
    rel = np.abs(losses - loss32) / loss32
    text = (f"{recipe}: {n} seeds x {losses.shape[1]} epochs; epoch-1 loss vs oracle f32: median {np.median(rel1):.2e} max {rel1.max():.2e}; "
            f"all epochs: HIP vs f32 median {np.median(rel):.2e} max {rel.max():.2e}; oracle f32 vs f64 median {np.median(gap):.2e} "
            f"max {gap.max():.2e}; launches {({k: v for k, v in d.items() if v})}")
    # epoch 1, paired per seed: no further from the float32 oracle than its own float64 run is, on average (the mutation leg of
    # tests/test_hip_round6.py holds single-term bf16 products against the same check: red)
    ok1, t1 = T.first_epoch_check(losses[:, 0], loss32[:, 0], loss64[:, 0])
    text += "\n    " + t1
    assert ok1, text
    # every later epoch: the mean deviation over the seeds within twice the oracles' own mean gap at that epoch (+ 0.2 %) -- ddi at
    # h = 512 crosses its steep phase at a different epoch per seed and arithmetic (mean gap up to 0.28 around epoch 11), which
    # is the lottery the teacher-forced tests pin down step by step
    assert (rel.mean(0) <= 2.0 * gap.mean(0) + 2e-3).all(), text + f"\n    per epoch HIP {np.round(rel.mean(0), 4)} oracle gap {np.round(gap.mean(0), 4)}"
    # (c) level
    c = T.compare(hip, ref32, ref64, recipe)
    text += "\n" + T.describe(f"{recipe}, HIP {P.ops.GEMM_MATH['mode']}", c)
    ki = T.metrics_of(recipe).index("Hits@20")
    ka = T.metrics_of(recipe).index("AUC")
    text += (f"\n    last epoch, mean over seeds: Hits@20 valid HIP {hip[:, -1, ki, 0].mean():.2f}  oracle f32 {ref32[:, -1, ki, 0].mean():.2f};"
             f"  AUC valid HIP {hip[:, -1, ka, 0].mean():.2f}  oracle f32 {ref32[:, -1, ka, 0].mean():.2f}")
    print(text)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "trained_parity_wide_r05.txt"), "a") as f:
            f.write(text + "\n")
        np.savez_compressed(os.path.join(out_dir, f"trained_curves_{recipe}.npz"), hits=hip.astype(np.float32), losses=losses)
    if recipe == "collab_wide":
        assert (np.abs(c["diff_f32"]) <= 0.3 + 2.0 * c["diff_f32_se"]).all(), text       # (s.e. 0.2: a band of +-0.7)
        assert c["mw_p"] > 0.05 / 4, text
        assert abs(c["reached_hip"] - c["reached_f32"]) <= 0.26, text                      # (16 seeds: steps of 0.06)
        assert 70.0 < c["final_f32"].min() and c["final_f32"].max() < 99.0, text          # trained, not saturated
        return
    # ddi_wide.  The centre of 8 seeds carries a standard error of 3 points here (one float32-oracle seed is still short of the
    # plateau after 30 epochs), which made "0.3 + 2 s.e." a band of +-6.5.  Held instead, seed by seed:
    #   * every seed on which both the HIP run and the float32 oracle sit on the plateau: the two final levels within 0.3, outright;
    #   * HIP reaches the plateau on at least as many seeds as the float32 oracle, less one;
    #   * BEFORE the steep phase (epochs 1-6, AUC 55 -> 79 %), paired per seed: the mean |AUC - oracle f32| over the seeds within
    #     twice the oracle's own float64 gap at that epoch (+ 0.05 points) -- a product with a bias in its gradients leaves
    #     this band in the first epochs, long before the plateau hides it (scripts/calibrate_wide_parity.py has the table);
    #   * epochs to 88 % distributed like the float32 oracle's (Mann-Whitney).
    fh, f32 = T.final_level(hip, recipe), T.final_level(ref32, recipe)
    on32, onh = (f32 >= 88.0).all(1), (fh >= 88.0).all(1)
    text += f"\n    per seed final level (valid / test): HIP {np.round(fh, 2).tolist()}  oracle f32 {np.round(f32, 2).tolist()}"
    assert on32.sum() >= 6, text
    # (WHICH seed is the one-in-eight straggler differs between arithmetics -- the float32 and float64 oracles disagree on it too)
    both = on32 & onh
    assert both.sum() >= 5 and (np.abs(fh[both] - f32[both]) <= 0.3).all(), text
    assert onh.sum() >= on32.sum() - 1, text
    pre = slice(0, 6)
    d_hip = np.abs(hip[:, pre, ka, 0] - ref32[:, pre, ka, 0]).mean(0)
    d_orc = np.abs(ref64[:, pre, ka, 0] - ref32[:, pre, ka, 0]).mean(0)
    text += f"\n    AUC valid, epochs 1-6, mean |x - oracle f32| over seeds: HIP {np.round(d_hip, 3).tolist()}  oracle f64 {np.round(d_orc, 3).tolist()}"
    print(text.splitlines()[-1])
    assert (d_hip <= 2.0 * d_orc + 0.05).all(), text
    assert c["mw_p"] > 0.05 / 4, text
    assert c["final_f32"].min() > 85.0, text                                               # the plateau was reached


# ------------------------------------------------ the wide weight gradient (csrc/gemm_wgw.hip) ----
@pytest.mark.parametrize("m,n,k,pad", [(200, 200, 300_001, 0), (200, 180, 65_536, 0), (224, 224, 40_000, 0), (132, 132, 50_007, 0),
                                       (200, 200, 32_768, 24), (196, 160, 33_333, 8), (204, 192, 100_000, 0),
                                       (200, 200, 32_790, 0)])        # 2 050 K-steps in 256 slices of 9: the last 28 slices are empty
def test_wide_weight_gradient_is_fp32_grade_and_deterministic(P, m, n, k, pad):
    """C[m, n] = A^T B over k rows with the whole result held by one workgroup per K slice (layers 129 .. 224 wide, citation2's
    h = 200): asserted by the launch counter to be the kernel that ran; against float64 inside the per-product bound of every
    split-bf16 launch (2^-22 of sum |a||b|) and the f32 summation bound; the same bits twice; and the 128 x 128 kernels'
    result (the form switched off: another summation order) within f32 round-off of it.  k odd / not a multiple of 16 (the
    partial last K-step), widths that leave 1 .. 3 column blocks partly empty, operands that are column slices of wider
    buffers (pad > 0: leading dimension > width)."""
    ops = P.ops
    old = ops.GEMM_MATH["mode"]
    ops.GEMM_MATH["mode"] = "bf16x3"
    try:
        gen = torch.Generator(device="cuda").manual_seed(m * 7 + n * 3 + k)
        a_buf = torch.randn(k, m + pad, device="cuda", generator=gen) * 0.05
        b_buf = torch.randn(k, n + pad, device="cuda", generator=gen)
        a, b = a_buf[:, :m], b_buf[:, pad:pad + n] if pad % 4 == 0 else b_buf[:, :n]
        c0 = ops.launch_counts()
        got = ops.gemm([(a, b)], True, False)
        d = _delta(P, c0)
        assert d["gemm_wgrad_wide"] == 1 and d["gemm_splitk_reduce"] == 1 and d["gemm_tile_x3"] == 0, d
        assert torch.equal(got, ops.gemm([(a, b)], True, False))
        a64, b64 = a.double(), b.double()
        want = a64.t() @ b64
        mag = a64.abs().t() @ b64.abs()
        err = float(((got.double() - want).abs() / mag).max())
        ops.GEMM_WIDE_WGRAD["enabled"] = False
        c0 = ops.launch_counts()
        tile = ops.gemm([(a, b)], True, False)
        d = _delta(P, c0)
        assert d["gemm_wgrad_wide"] == 0 and d["gemm_tile_x3"] == 1, d
        err_tile = float(((tile.double() - want).abs() / mag).max())
        print(f"{m}x{n} over {k}: wide {err:.2e}  tile {err_tile:.2e} of sum |a||b|")
        assert err <= 2.0 ** -22 and err <= 1.5 * err_tile + 2e-8
    finally:
        ops.GEMM_MATH["mode"] = old
        ops.GEMM_WIDE_WGRAD["enabled"] = True


def test_wide_weight_gradient_rule(P):
    """where the form applies (plnlp_gemm_wide_wgrad_slices) and that everything else keeps the 128 x 128 kernels: a gathered
    operand, a width <= 128 or > 224 or not a multiple of 4, a short reduction, the f32 math, an operand off 16-byte alignment"""
    ops = P.ops
    old = ops.GEMM_MATH["mode"]
    ops.GEMM_MATH["mode"] = "bf16x3"
    try:
        def ran(a, b, **kw):
            c0 = ops.launch_counts()
            ops.gemm([(a, b)], True, False, **kw)
            return _delta(P, c0)["gemm_wgrad_wide"]
        k = 40_000
        x = torch.randn(k, 232, device="cuda")
        assert ran(x[:, :200], x[:, :200]) == 1
        assert ran(x[:, :128], x[:, :200]) == 0 and ran(x[:, :200], x[:, :128]) == 0
        assert ran(x[:, :228], x[:, :200]) == 0 and ran(x[:, :202], x[:, :200]) == 0
        assert ran(x[:, 2:202], x[:, :200]) == 0                      # 8 bytes off
        assert ran(x[:30_000, :200], x[:30_000, :200]) == 0
        rows = torch.arange(k, device="cuda", dtype=torch.int32)
        assert ran(x[:, :200], x[:, :200], b_index=rows) == 0                  # (gathered rows: the 256-block geometry only)
        ops.GEMM_MATH["mode"] = "f32"
        assert ran(x[:, :200], x[:, :200]) == 0
    finally:
        ops.GEMM_MATH["mode"] = old


@pytest.mark.parametrize("m,n,k,form", [(256, 256, 40_000, "plain"), (256, 512, 132_224, "plain"), (512, 512, 65_536, "plain"),
                                        (512, 512, 32_790, "plain"),   # 2 050 K-steps in 64 slices of 33: the last slice is partial, one is empty
                                        (256, 512, 32_769, "pair_compact"),
                                        (256, 512, 50_007, "gathered"), (256, 512, 132_224, "pair"),
                                        (256, 512, 40_001, "pair_gathered"), (256, 512, 132_224, "pair_compact")])
def test_wide_weight_gradient_in_256_blocks(P, m, n, k, form):
    """the same kernel with 8 waves and 256 x 256 blocks of the result (collab's 256 x 512, ddi's 512 x 512): B in one buffer or
    as the pair [x1 | x2] of two (ops.wgrad_pair), its rows as stored, gathered through `rows`, or gathered in x2 only
    (x1 compact -- the collab step's launch).  Asserted by the launch counter; against float64 at the split-bf16 bound and
    against the 128 x 128 kernels' own error; the same bits twice."""
    ops = P.ops
    old = ops.GEMM_MATH["mode"]
    ops.GEMM_MATH["mode"] = "bf16x3"
    try:
        gen = torch.Generator(device="cuda").manual_seed(m + n + k)
        dz = torch.randn(k, m, device="cuda", generator=gen) * 0.05
        n_src = k if form in ("plain", "pair") else 3 * k // 2
        rows = None
        if form not in ("plain", "pair"):
            rows = torch.randperm(n_src, device="cuda", generator=gen)[:k].sort().values.to(torch.int32)
        if form.startswith("pair"):
            n1 = n // 2
            x1 = torch.randn(k if form == "pair_compact" else n_src, n1, device="cuda", generator=gen)
            x2 = torch.randn(n_src, n - n1, device="cuda", generator=gen)

            def run():
                return torch.cat(ops.wgrad_pair(dz, x1, x2, rows=rows, x1_compact=form == "pair_compact"), dim=1)
            b1 = x1 if (rows is None or form == "pair_compact") else x1[rows.long()]
            b_eff = torch.cat([b1, x2 if rows is None else x2[rows.long()]], dim=1)
        else:
            x = torch.randn(n_src, n, device="cuda", generator=gen)

            def run():
                return ops.gemm([(dz, x)], True, False, b_index=rows)
            b_eff = x if rows is None else x[rows.long()]
        c0 = ops.launch_counts()
        got = run()
        d = _delta(P, c0)
        assert d["gemm_wgrad_wide"] == 1 and d["gemm_splitk_reduce"] == 1 and d["gemm_tile_x3"] == 0, d
        assert torch.equal(got, run())
        want = dz.double().t() @ b_eff.double()
        mag = dz.double().abs().t() @ b_eff.double().abs()
        err = float(((got.double() - want).abs() / mag).max())
        ops.GEMM_WIDE_WGRAD["enabled"] = False
        c0 = ops.launch_counts()
        tile = run()
        d = _delta(P, c0)
        assert d["gemm_wgrad_wide"] == 0 and d["gemm_tile_x3"] >= 1, d
        err_tile = float(((tile.double() - want).abs() / mag).max())
        print(f"{form} {m}x{n} over {k}: wide {err:.2e}  tile {err_tile:.2e} of sum |a||b|")
        assert err <= 2.0 ** -22 and err <= 1.5 * err_tile + 2e-8
    finally:
        ops.GEMM_MATH["mode"] = old
        ops.GEMM_WIDE_WGRAD["enabled"] = True
