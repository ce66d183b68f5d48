"""GPU parity, sixth batch (round 6):

  * the stationary-weights split-bf16 product with a whole 256-row block of the result per workgroup (csrc/gemm_x3b.hip): the
    SAME BITS as the 128-row kernel (csrc/gemm_x3s.hip) whatever the walk hands out -- full blocks, leading / trailing half
    blocks, a partial last block, an odd number of K-steps (the register sets trade places between blocks), ragged K, two
    K-segments with gathered rows, the pair form's two results, every epilogue, the row-dot head -- which kernel ran asserted
    by the launch counters; and float64;
  * parity in the configuration bench.py TIMES: teacher-forced steps with the recipes' dropout ON, the oracle
    handed the same counter masks; REVERSE teacher-forced steps at trained states (the HIP model free-runs, the float32 oracle
    takes over its state after epochs 1 / 5 / 15 / last and steps beside it); a mutation at width that must go red.
Same rules as tests/test_hip_parity.py: through the C ABI, fp32 tolerance 1e-5 relative, integer outputs bit-exact."""
import os

import pytest
import torch

from gpu_util import close

pytestmark = pytest.mark.gpu


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


def _arms(P, fn, modes=("off", "all", "all-nolead")):
    """fn() under each setting of ops.GEMM_BLOCK (split-bf16 products): [(mode, result, launch-count delta)]"""
    old_math, old_blk = P.ops.GEMM_MATH["mode"], P.ops.GEMM_BLOCK["mode"]
    P.ops.GEMM_MATH["mode"] = "bf16x3"
    out = []
    try:
        for mode in modes:
            P.ops.GEMM_BLOCK["mode"] = mode
            c0 = P.ops.launch_counts()
            res = fn()
            torch.cuda.synchronize()
            out.append((mode, res, _delta(P, c0)))
    finally:
        P.ops.GEMM_MATH["mode"], P.ops.GEMM_BLOCK["mode"] = old_math, old_blk
        P.ops._apply_gemm_block()
    return out


def _same_bits(arms, launches=1):
    ref = arms[0][1]
    assert arms[0][2]["gemm_x3b"] == 0, arms[0][2]
    for mode, res, d in arms[1:]:
        assert d["gemm_x3b"] == launches and d["gemm_x3s"] == 0, (mode, d)
        if isinstance(ref, (tuple, list)):
            for u, v in zip(res, ref):
                assert torch.equal(u, v), mode
        else:
            assert torch.equal(res, ref), mode


# rows: 517 blocks (the collab step: 2 full + a half per workgroup), exactly one half block per CU, a partial last block whose
# last wave is empty, an uneven share (some workgroups one half block more), two blocks per CU
@pytest.mark.parametrize("m,n,k", [(132_224, 256, 256), (32_768, 256, 48), (70_001, 512, 112), (33_000, 200, 192),
                                   (65_536, 224, 200), (98_405, 200, 180), (131_072, 512, 512), (40_000, 256, 520)])
def test_block_kernel_has_the_bits_of_the_panel_kernel(P, m, n, k):
    """whole 256-row blocks per workgroup vs 128-row panels: the same six bf16 products per K-step in the same order, K walked in
    the same order, the same write-back -- bit for bit, with and without the leading half blocks, for weights stored [N, K]
    and [K, N]; f32-grade distance from float64 on sampled rows; launch-to-launch determinism."""
    gen = torch.Generator(device="cuda").manual_seed(m + n + k)
    a = torch.randn(m, k, device="cuda", generator=gen)
    w_nk = torch.randn(n, k, device="cuda", generator=gen) * 0.1
    w_kn = w_nk.t().contiguous()
    tile_baseline = 192 < n <= 224 and k % 16 == 0      # (off: the 128 x 128 tile kernel takes these -- another order of the same sums)
    for b, bt in ((w_nk, True), (w_kn, False)):
        arms = _arms(P, lambda: P.ops.gemm([(a, b)], False, bt))
        if tile_baseline:
            assert arms[0][2]["gemm_tile_x3"] == 1 and arms[1][2]["gemm_x3b"] == 1 and arms[2][2]["gemm_x3b"] == 1
            assert torch.equal(arms[1][1], arms[2][1])
            bound = a.abs().double().sum(1, keepdim=True) * float(w_nk.abs().max())
            assert float(((arms[1][1].double() - arms[0][1].double()).abs() / bound.clamp_min(1e-30)).max()) <= 2e-7
        else:
            _same_bits(arms)
    again = _arms(P, lambda: P.ops.gemm([(a, w_nk)], False, True), modes=("off", "all"))
    assert torch.equal(again[1][1], arms[1][1])
    rows = torch.randint(0, m, (256,), device="cuda", generator=gen)
    rows[:4] = torch.tensor([0, m - 1, m - 2, min(m - 1, 255)], device="cuda")
    want = a[rows].double() @ w_nk.double().t()
    bound = a[rows].double().abs() @ w_nk.double().abs().t()
    assert float(((arms[1][1][rows].double() - want).abs() / bound).max()) <= 5e-7


def test_block_kernel_default_rule(P):
    """by default the whole-block kernel takes the 224-column tiles (a layer 193 .. 224 wide: citation2's h = 200) from 32 768
    rows on -- K a multiple of 16 (which used to run the 128 x 128 tile kernel) and ragged K alike --, the 256-column tiles stay
    on the panel kernel, fewer rows too"""
    old = P.ops.GEMM_MATH["mode"]
    P.ops.GEMM_MATH["mode"] = "bf16x3"
    try:
        P.ops._apply_gemm_block()
        for m, n, k, want in ((40_000, 200, 192, "gemm_x3b"), (40_000, 200, 200, "gemm_x3b"), (40_000, 224, 64, "gemm_x3b"),
                              (40_000, 256, 256, "gemm_x3s"), (20_000, 200, 200, "gemm_x3s"), (20_000, 200, 192, "gemm_tile_x3")):
            a = torch.randn(m, k, device="cuda")
            w = torch.randn(n, k, device="cuda")
            c0 = P.ops.launch_counts()
            got = P.ops.gemm([(a, w)], False, True)
            d = _delta(P, c0)
            assert d[want] == 1 and d["gemm_x3b"] + d["gemm_x3s"] + d["gemm_tile_x3"] == 1, (m, n, k, d)
            close(got[:512], (a[:512].double() @ w.double().t()).float(), rtol=1e-5, atol=1e-4)
    finally:
        P.ops.GEMM_MATH["mode"] = old


def test_block_kernel_step_forms(P):
    """the launches of a training step on the whole-block kernel, bits of the panel kernel: the conv at the touched rows (two
    K-segments, the root operand's rows GATHERED, bias + relu + dropout drawn at the ORIGINAL rows), the pair of data gradients
    (B from two [K, N] buffers, the result into two tensors), the 1-output head in the epilogue (PLNLP_EPI_ROWDOT); accumulate and
    gate epilogues (row-dependent operands) stay on the panel kernel under every setting (x3b::takes)"""
    from plnlp_amd import _lib as L
    gen = torch.Generator(device="cuda").manual_seed(6)
    t_rows, n_src, h = 66_061, 90_000, 256
    agg = torch.randn(t_rows, h, device="cuda", generator=gen)
    x = torch.randn(n_src, h, device="cuda", generator=gen)
    rows = torch.randperm(n_src, device="cuda", generator=gen)[:t_rows].sort().values.to(torch.int32)
    w_l = torch.randn(h, h, device="cuda", generator=gen) * 0.05
    w_r = torch.randn(h, h, device="cuda", generator=gen) * 0.05
    bias = torch.randn(h, device="cuda", generator=gen)
    epi = L.Epilogue()
    epi.flags = L.EPI_BIAS | L.EPI_RELU | L.EPI_DROPOUT
    epi.bias = bias.data_ptr()
    epi.dropout_p, epi.dropout_seed = 0.3, 77
    epi.dropout_row_index = rows.data_ptr()
    arms = _arms(P, lambda: P.ops.gemm([(agg, w_l), (x, w_r)], False, True, epilogue=epi, a_index=[None, rows]))
    _same_bits(arms)
    want = torch.relu(agg.double() @ w_l.double().t() + x[rows.long()].double() @ w_r.double().t() + bias.double())
    got = arms[1][1]
    kept = (got != 0) & (want > 1e-4)
    assert 0.5 < float(kept.float().mean() / (want > 1e-4).float().mean()) < 0.9           # p = 0.3 dropped
    close(got[kept], (want / 0.7).float()[kept], rtol=1e-5, atol=1e-4)
    dz = torch.randn(t_rows, h, device="cuda", generator=gen)
    arms = _arms(P, lambda: P.ops.dgrad_pair(dz, w_r, w_l))
    _same_bits(arms)
    close(arms[1][1][0], (dz.double() @ w_r.double()).float(), rtol=1e-5, atol=1e-4)
    close(arms[1][1][1], (dz.double() @ w_l.double()).float(), rtol=1e-5, atol=1e-4)
    base = torch.randn(t_rows, h, device="cuda", generator=gen)
    gate = torch.randn(t_rows, h, device="cuda", generator=gen)
    for flags in (L.EPI_ACCUM, L.EPI_GATE, L.EPI_ACCUM | L.EPI_GATE):
        e2 = L.Epilogue()
        e2.flags = flags
        if flags & L.EPI_GATE:
            e2.gate, e2.ld_gate, e2.gate_scale = gate.data_ptr(), h, 1.25
        arms = _arms(P, lambda: P.ops.gemm([(dz, w_l)], False, False, out=base.clone(), epilogue=e2))
        assert all(d["gemm_x3b"] == 0 and d["gemm_x3s"] == 1 for _, _, d in arms), arms
        assert all(torch.equal(res, arms[0][1]) for _, res, _ in arms)
        want = dz.double() @ w_l.double()
        if flags & L.EPI_ACCUM:
            want = want + base.double()
        if flags & L.EPI_GATE:
            want = torch.where(gate > 0, want * 1.25, torch.zeros_like(want))
        close(arms[1][1], want.float(), rtol=1e-5, atol=1e-4)
    # the head in the epilogue: 512-wide hidden layer = two column tiles, each leaves its partial dot product (ddi's scorer: at
    # 262 144 rows the panel kernel picks 256-column tiles too -- with 128-column tiles the partials associate differently)
    feat = 512
    xs = torch.randn(262_144, feat, device="cuda", generator=gen)
    w1 = torch.randn(feat, feat, device="cuda", generator=gen) / feat ** 0.5
    b1 = torch.randn(feat, device="cuda", generator=gen) * 0.1
    w2 = torch.randn(1, feat, device="cuda", generator=gen) / feat ** 0.5
    b2 = torch.tensor([0.25], device="cuda")
    mk = lambda: L.make_epilogue(bias=b1, relu=True, dropout_p=0.3, dropout_seed=9)
    arms = _arms(P, lambda: P.ops.gemm([(xs, w1)], False, True, epilogue=mk(), rowdot=(w2, b2)))
    assert arms[0][1][1] is not None
    _same_bits(arms)


def test_block_kernel_gcn_shapes(P):
    """citation2's products at h = 200 (one 224-column tile, NB = 7: the default of this kernel): first layer K = 192 with bias +
    relu + dropout, second layer K = 200 (ragged, 13 K-steps: odd) -- bits of the panel kernel, float64 on sampled rows; the data
    gradient with a gate is the panel kernel's under every setting"""
    from plnlp_amd import _lib as L
    gen = torch.Generator(device="cuda").manual_seed(16)
    m, h = 150_013, 200
    for k in (192, 200):
        a = torch.randn(m, k, device="cuda", generator=gen)
        w = torch.randn(h, k, device="cuda", generator=gen) * 0.1
        bias = torch.randn(h, device="cuda", generator=gen)
        mk = lambda: L.make_epilogue(bias=bias, relu=True, dropout_p=0.5, dropout_seed=k)
        for e in (None, mk):
            arms = _arms(P, lambda: P.ops.gemm([(a, w)], False, True, epilogue=e() if e else None), modes=("off", "auto", "nolead"))
            if k % 16 == 0:              # (off: the 128 x 128 tile kernel takes whole K-steps at this width -- another order of the same sums)
                assert arms[0][2]["gemm_tile_x3"] == 1 and arms[1][2]["gemm_x3b"] == 1 and arms[2][2]["gemm_x3b"] == 1
                assert torch.equal(arms[1][1], arms[2][1])
                if e is None:
                    close(arms[1][1], arms[0][1], rtol=1e-5, atol=1e-5)
            else:
                _same_bits(arms)
        rows = torch.randint(0, m, (128,), device="cuda", generator=gen)
        plain = _arms(P, lambda: P.ops.gemm([(a, w)], False, True), modes=("auto",))[0][1]
        want = a[rows].double() @ w.double().t()
        bound = a[rows].double().abs() @ w.double().abs().t()
        assert float(((plain[rows].double() - want).abs() / bound).max()) <= 5e-7
    dz = torch.randn(m, h, device="cuda", generator=gen)
    w2 = torch.randn(h, h, device="cuda", generator=gen) * 0.1
    gate = torch.randn(m, h, device="cuda", generator=gen)
    e2 = L.Epilogue()
    e2.flags = L.EPI_GATE
    e2.gate, e2.ld_gate, e2.gate_scale = gate.data_ptr(), h, 2.0
    arms = _arms(P, lambda: P.ops.gemm([(dz, w2)], False, False, epilogue=e2), modes=("off", "auto", "nolead"))
    assert all(d["gemm_x3b"] == 0 and d["gemm_x3s"] == 1 for _, _, d in arms), arms
    assert all(torch.equal(res, arms[0][1]) for _, res, _ in arms)
    close(arms[1][1], torch.where(gate > 0, (dz.double() @ w2.double()) * 2.0, torch.zeros(m, h, device="cuda", dtype=torch.float64)).float(),
          rtol=1e-5, atol=1e-4)


# ------------------------------------ parity in the configuration that is TIMED: the recipes' dropout ON ----
def _step_seeds(P, n):
    """the next n seeds of the product's dropout stream, without consuming them (ops.next_seed: one per dropout call)"""
    base, ctr = P.ops.seed_state()
    seeds = [P.ops.next_seed() for _ in range(n)]
    P.ops.set_seed_state(base, ctr)
    return seeds


def _hand_masks_to_the_oracle(O, ref, seeds, p, layers, mlp):
    """the oracle applies the masks the HIP step is about to draw (oracle/reference_path.py: GNNRef.dropout_fn, MLPPredictorRef.
    dropout_fn; the masks themselves are bit-identical -- tests/test_hip_parity.py).  Seeds are consumed in call order: the
    encoder's layers that carry a dropout (layer.py:21-26), then the scorer's hidden layer -- ONE draw over the [pos | neg]
    rows the product scores in one pass, which the reference's two predictor calls (model.py:155-156) see as rows 0 .. B-1
    and B .. B + kB - 1 of the same mask."""
    n_enc = layers - 1 if layers > 1 else 1
    assert len(seeds) == n_enc + (1 if mlp else 0)
    ref.encoder.dropout_fn = lambda x, i: O.counter_dropout(x, p, seeds[i])
    if mlp:
        state = {"row0": 0}

        def fn(x, i):
            keep = torch.from_numpy(O.dropout_keep_mask(seeds[n_enc + i], x.shape[0], x.shape[1], p, row0=state["row0"]))
            state["row0"] += x.shape[0]
            return x * keep.to(x.dtype) * torch.tensor(1.0 / (1.0 - p), dtype=torch.float32).to(x.dtype)
        ref.predictor.dropout_fn = fn


def _oracle_to_hip(model, ref):
    """the oracle trainer's parameters, Adam state and learning rate into the HIP model (same parameter order)"""
    from plnlp_amd.optim import fused_adam_state
    with torch.no_grad():
        for p, q in zip(model.para_list, ref.params):
            p.copy_(q.detach().to(p.device))
            fused_adam_state(model.optimizer, p)
            st, sq = model.optimizer.state[p], ref.optimizer.state.get(q, {})
            if sq:
                st["exp_avg"].copy_(sq["exp_avg"].to(p.device))
                st["exp_avg_sq"].copy_(sq["exp_avg_sq"].to(p.device))
                st["step"] = int(sq["step"])
            else:
                st["exp_avg"].zero_()
                st["exp_avg_sq"].zero_()
                st["step"] = 0


def _hip_to_oracle(ref, model):
    """... and the other way round: the HIP model's TRAINED state into the float32 oracle"""
    from plnlp_amd.optim import fused_adam_state
    with torch.no_grad():
        for p, q in zip(model.para_list, ref.params):
            q.copy_(p.detach().cpu())
            fused_adam_state(model.optimizer, p)
            st = model.optimizer.state[p]
            ref.optimizer.state[q] = {"step": torch.tensor(float(int(st["step"]))),
                                      "exp_avg": st["exp_avg"].detach().cpu().reshape(q.shape).clone(),
                                      "exp_avg_sq": st["exp_avg_sq"].detach().cpu().reshape(q.shape).clone()}
    lrs = {float(g["lr"]) for g in model.optimizer.param_groups}
    assert len(lrs) == 1
    for g in ref.optimizer.param_groups:
        g["lr"] = lrs.pop()


def _paired_step(P, O, m, ref, data, r, pos, neg, w, p_drop):
    """one step of both trainers from the state they share; -> (relative loss deviation, worst 99 % quantile of |weight matrix / table
    - oracle| after the step, largest single deviation)"""
    n_seeds = ((r["layers"] - 1 if r["layers"] > 1 else 1) + (1 if r["predictor"] == "MLP" else 0)) if p_drop > 0 else 0
    if p_drop > 0:
        seeds = _step_seeds(P, n_seeds)
        _hand_masks_to_the_oracle(O, ref, seeds, p_drop, r["layers"], r["predictor"] == "MLP")
    ctr0 = P.ops.seed_state()[1]
    loss_hip = float(m.train_step(data, pos.cuda(), neg.cuda(), r["k"], None if w is None else w.cuda()))
    assert P.ops.seed_state()[1] == ctr0 + n_seeds          # the step drew exactly the masks the oracle was handed
    loss_ref = float(ref.step(pos, neg, r["k"], w)[0])
    torch.cuda.synchronize()
    bulk, worst = 0.0, 0.0
    for p, q in zip(m.para_list, ref.params):
        d = (p.detach().cpu().double() - q.detach().double()).abs().flatten()
        if d.numel() >= 4096:
            bulk = max(bulk, float(d.kthvalue(int(0.99 * d.numel()))[0]))
        worst = max(worst, float(d.max()))
    return abs(loss_hip - loss_ref) / abs(loss_ref), bulk, worst


def _epoch_batches(O, T, g, csr, r, recipe, seed, epoch, steps):
    """positives (walk pairs where the recipe has them), negatives and the first `steps` batches of an epoch, as the harness draws them"""
    n = g["num_nodes"]
    torch.manual_seed(T.epoch_seed(epoch, seed))
    pos, w = g["train"], None
    if r["walk_length"]:
        walk = O.random_walk_ref(csr, pos.reshape(-1), r["walk_length"], T.walk_seed(epoch, seed))
        pos, w = O.random_walk_pairs_ref(walk, r["walk_length"])
    _, neg = O.pos_neg_edges_ref("train", {"train": {"edge": pos}}, num_nodes=n, neg_sampler_name="local", num_neg=r["k"])
    return pos, neg, w, O.batch_permutation(pos.size(0), r["batch"], True)[:steps]


def _teacher_forced_with_dropout(P, recipe, p_drop, steps):
    import oracle as O
    import trained_parity as T
    r = T.RECIPES[recipe]
    g = T.problem(recipe)
    n = g["num_nodes"]
    enc, pred, emb = T.initial_modules(recipe, 0)
    adj = g["adj_t"]
    csr = O.CSR(adj.rowptr, adj.col.to(torch.int64), None, n)
    ref = O.TrainerRef(enc, pred, emb, csr, loss_name=r["loss"], lr=r["lr"], clip_norm=r["clip"])
    m, data, _ = T.hip_model(P, recipe, 0, dropout=p_drop)
    m.encoder.train()
    m.predictor.train()
    P.manual_seed(2024)
    pos, neg, w, batches = _epoch_batches(O, T, g, csr, r, recipe, 0, 0, steps)
    c0 = P.ops.launch_counts()
    out = []
    for perm in batches:
        _oracle_to_hip(m, ref)
        out.append(_paired_step(P, O, m, ref, data, r, pos[perm], neg[perm], None if w is None else w[perm], p_drop))
    return out, _delta(P, c0), len(batches)


@pytest.mark.parametrize("recipe", ["collab_wide", "ddi_wide"])
def test_teacher_forced_steps_with_the_recipes_dropout(P, recipe):
    """VERDICT r5 #2a.  bench.py times the recipes with their dropout of 0.3 (README.md:24,35; layer.py:21-26,84-85); every
    trajectory / teacher-forced test so far ran dropout = 0.  Here the whole step -- the forward mask in the conv's / the hidden
    layer's epilogue, its gate in the backward, the clips, Adam -- is compared with the float32 oracle handed the SAME counter masks:
    the HIP model is reset to the oracle's state before each of 6 steps at h = 256 / 512 and takes the same batch.  Per-step loss
    within 1e-5, the 99 % quantile of |weight - oracle| after the step within 2e-5, through the kernels the benchmark runs
    (launch counters)."""
    out, d, steps = _teacher_forced_with_dropout(P, recipe, 0.3, 6)
    worst_loss, worst_bulk = max(o[0] for o in out), max(o[1] for o in out)
    print(f"teacher-forced {recipe}, dropout 0.3: {steps} steps, worst per-step loss deviation {worst_loss:.2e}, worst 99 % quantile of "
          f"|weight - oracle| after a step {worst_bulk:.2e}; launches {({k: v for k, v in d.items() if v})}")
    assert steps == 6 and worst_loss <= 1e-5 and worst_bulk <= 2e-5, out
    assert d["gemm_x3s"] + d["gemm_x3b"] >= 2 * steps, d
    assert d["agg_fused"] + d["agg_fused_hub_xcd"] + d["agg_vec_slabs"] + d["agg_chunk"] + d["agg_dense"] >= steps, d


@pytest.mark.parametrize("recipe", ["collab_wide", "ddi_wide"])
def test_a_mutation_at_width_goes_red(P, recipe):
    """the power of the two statements above and below: with every dense operand rounded to ONE bf16 term (tests/trained_parity.py::
    Mutation) the teacher-forced steps leave the 1e-5 band, and the first-epoch check of the trained-regime legs
    (trained_parity.first_epoch_check, asserted for the clean product by tests/test_hip_round5.py) fails"""
    import numpy as np
    import trained_parity as T
    with T.Mutation(P, "bf16_operands"):
        out, _, _ = _teacher_forced_with_dropout(P, recipe, 0.3, 3)
    assert max(o[0] for o in out) > 1e-5 or max(o[1] for o in out) > 2e-5, out
    g12 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g12_trained_curves_wide.npz"))
    l32, l64 = g12[f"{recipe}_f32_loss"][:, 0], g12[f"{recipe}_f64_loss"][:, 0]
    losses = np.array([T.run_hip(P, recipe, s, "bf16x3", mutation="bf16_operands", epochs=1)[1][0] for s in range(l32.shape[0])])
    ok, text = T.first_epoch_check(losses, l32, l64)
    print(f"{recipe}, single-term bf16 products: " + text)
    assert not ok, text


@pytest.mark.parametrize("recipe", ["collab_wide", "ddi_wide"])
def test_reverse_teacher_forced_steps_at_trained_states(P, recipe):
    """VERDICT r5 #2b.  Every per-step statement so far starts from the INITIAL state, where every relu / hinge gate sits near its
    symmetric point.  Here the HIP model free-runs the recipe (dropout 0.3 on), and after epochs 1, 5, 15 and the last one ITS state
    -- parameters, Adam moments, step count, learning rate -- is copied into the float32 oracle, which then takes the next 3 steps
    beside it (state re-copied before each, the oracle handed the same masks): per-step loss within 1e-5, 99 % quantile of
    |weight - oracle| after the step within 2e-5 -- the h = 256 / 512 steps of a TRAINED model are the reference's steps."""
    import oracle as O
    import trained_parity as T
    r = T.RECIPES[recipe]
    g = T.problem(recipe)
    n = g["num_nodes"]
    adj = g["adj_t"]
    csr = O.CSR(adj.rowptr, adj.col.to(torch.int64), None, n)
    marks = sorted({e for e in (1, 5, 15, r["epochs"]) if e <= r["epochs"]})
    report = []
    for seed in (0, 1):
        enc, pred, emb = T.initial_modules(recipe, seed)
        ref = O.TrainerRef(enc, pred, emb, csr, loss_name=r["loss"], lr=r["lr"], clip_norm=r["clip"])
        m, data, split = T.hip_model(P, recipe, seed, dropout=0.3)
        P.manual_seed(77 + seed)
        first_loss = None
        for epoch in range(r["epochs"]):
            loss = T.hip_epoch(P, m, data, split, recipe, seed, epoch)
            first_loss = loss if first_loss is None else first_loss
            if epoch + 1 not in marks:
                continue
            m.encoder.train()
            m.predictor.train()
            pos, neg, w, batches = _epoch_batches(O, T, g, csr, r, recipe, seed, 500 + epoch, 3)
            for perm in batches:
                _hip_to_oracle(ref, m)
                dl, bulk, worst = _paired_step(P, O, m, ref, data, r, pos[perm], neg[perm], None if w is None else w[perm], 0.3)
                report.append((seed, epoch + 1, loss, dl, bulk, worst))
                assert dl <= 1e-5 and bulk <= 2e-5, report[-1]
                assert worst <= 2.0 * max(float(gp["lr"]) for gp in m.optimizer.param_groups) + 1e-7, report[-1]
        assert loss < 0.8 * first_loss, (first_loss, loss)             # it did train
    print(f"reverse teacher-forced {recipe}: (seed, after epoch, epoch loss, per-step loss deviation, 99 % weight quantile, max) "
          + "; ".join(f"({s}, {e}, {l:.3f}, {dl:.1e}, {b:.1e}, {w:.1e})" for s, e, l, dl, b, w in report))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "reverse_teacher_forced_r06.txt"), "a") as f:
            for row in report:
                f.write(recipe + " " + " ".join(f"{v:.4g}" for v in row) + "\n")


def test_wide_weight_gradient_is_asked_for_not_implied(P):
    """the whole-block weight-gradient kernel used to be taken whenever a launch's split_k EQUALLED the slice count the
    form wants -- a caller that cut K that way by coincidence got it even with GEMM_WIDE_WGRAD off.  It is a flag of the launch now
    (PLNLP_GEMM_FLAG_WIDE_WGRAD): same slice count without the flag runs the 128 x 128 kernels."""
    ops = P.ops
    old_math, old_wide = ops.GEMM_MATH["mode"], ops.GEMM_WIDE_WGRAD["enabled"]
    ops.GEMM_MATH["mode"] = "bf16x3"
    try:
        a = torch.randn(40_000, 256, device="cuda")
        ops.GEMM_WIDE_WGRAD["enabled"] = True
        c0 = ops.launch_counts()
        wide = ops.gemm([(a, a)], True, False)
        d = _delta(P, c0)
        assert d["gemm_wgrad_wide"] == 1 and d["gemm_tile_x3"] == 0, d
        ops.GEMM_WIDE_WGRAD["enabled"] = False
        c0 = ops.launch_counts()
        tile = ops.gemm([(a, a)], True, False, split_k=256)         # the wide form's own slice count for this shape
        d = _delta(P, c0)
        assert d["gemm_wgrad_wide"] == 0 and d["gemm_tile_x3"] == 1 and d["gemm_splitk_reduce"] == 1, d
        close(tile, wide, rtol=1e-5, atol=1e-3)
        close(wide, (a.double().t() @ a.double()).float(), rtol=1e-5, atol=2e-2)
    finally:
        ops.GEMM_MATH["mode"], ops.GEMM_WIDE_WGRAD["enabled"] = old_math, old_wide


def test_padded_table_gradient_reaches_autograd_outside_the_trainers_step(P):
    """the first GCN layer over [padded table | features] used to set emb.weight.grad itself and return None for that
    input whenever the table was kept padded -- torch.autograd.grad got nothing, hooks never fired.  The direct (no-copy) path
    is now the trainer's own opt-in (ops.direct_table_grad, BaseModel._train_step_core); everywhere else the gradient is returned."""
    from gpu_util import rand_csr, to_graph
    import oracle as O
    n, e, f, h = 3000, 50, 16, 64
    torch.manual_seed(3)
    m = P.BaseModel(lr=0.01, dropout=0.0, grad_clip_norm=1.0, gnn_num_layers=2, mlp_num_layers=2, emb_hidden_channels=e,
                    gnn_hidden_channels=h, mlp_hidden_channels=h, num_nodes=n, num_node_feats=f, gnn_encoder_name="GCN",
                    predictor_name="MLP", loss_func="AUC", optimizer_name="Adam", device="cuda", use_node_feats=True,
                    train_node_emb=True)
    m.param_init()
    assert P.ops.padded_base(m.emb.weight.detach()) is not None

    class D:
        pass
    data = D()
    data.adj_t = to_graph(P, O.gcn_norm_csr(rand_csr(n, 20000, 21, weighted=False)))
    data.x = torch.randn(n, f, device="cuda")
    m.encoder.train()
    fired = []
    hook = m.emb.weight.register_hook(lambda g: fired.append(g.shape))
    out = m.encoder(m._input_feat(data), data.adj_t)
    (g,) = torch.autograd.grad(out.square().sum(), m.emb.weight)
    assert g.shape == (n, e) and m.emb.weight.grad is None and fired == [torch.Size([n, e])]
    hook.remove()
    with P.ops.direct_table_grad():                  # the trainer's path: the same gradient, as a view of the padded buffer
        out = m.encoder(m._input_feat(data), data.adj_t)
        out.square().sum().backward()
    assert torch.equal(m.emb.weight.grad, g) and P.ops.padded_base(m.emb.weight.grad) is not None


@pytest.mark.parametrize("m,n,k", [(200, 200, 300_001), (200, 180, 65_536), (224, 224, 40_000), (256, 256, 40_007), (512, 512, 70_000),
                                   (200, 200, 32_790)])
def test_bias_gradient_out_of_the_wide_weight_gradient_kernel(P, m, n, k):
    """the whole-block weight-gradient kernel (csrc/gemm_wgw.hip) stages every row of dz anyway -- the column sums of
    dz, the layer's bias gradient (the sums autograd forms for F.linear's bias, plnlp/layer.py:83, and PyG's conv bias), now
    come out of the same launch (plnlp_gemm_operand.a_colsum) instead of a second pass over dz.  Against float64 at 1e-5 of
    sum |dz|, the same bits twice, the weight gradient's bits untouched, the separate colsum kernel not launched."""
    ops = P.ops
    old = ops.GEMM_MATH["mode"]
    ops.GEMM_MATH["mode"] = "bf16x3"
    try:
        gen = torch.Generator(device="cuda").manual_seed(m + n + k)
        dz = torch.randn(k, m, device="cuda", generator=gen) * (1.0 + torch.arange(m, device="cuda") % 7)
        x = torch.randn(k, n, device="cuda", generator=gen)
        plain = ops.gemm([(dz, x)], True, False)
        outs = []
        for _ in range(2):
            cs = []
            c0 = ops.launch_counts()
            gw = ops.gemm([(dz, x)], True, False, a_colsum=cs)
            d = _delta(P, c0)
            assert d["gemm_wgrad_wide"] == 1 and len(cs) == 1 and cs[0].shape == (m,), d
            assert torch.equal(gw, plain)
            outs.append(cs[0].clone())
        assert torch.equal(outs[0], outs[1])
        want = dz.double().sum(0)
        bound = dz.double().abs().sum(0)
        assert float(((outs[0].double() - want).abs() / bound).max()) <= 1e-5
        close(outs[0], ops.colsum(dz), rtol=1e-5, atol=1e-5 * float(bound.max()))
        ops.COLSUM_IN_WGRAD["enabled"] = False
        cs = []
        assert torch.equal(ops.gemm([(dz, x)], True, False, a_colsum=cs), plain) and cs == []
    finally:
        ops.GEMM_MATH["mode"] = old
        ops.COLSUM_IN_WGRAD["enabled"] = True


def test_bias_gradient_rides_with_the_gathered_pair(P):
    """the collab step's weight gradients [dWl | dWr] = dz^T [agg | x[rows]] (two B buffers, gathered rows, 256 x 512) with the
    bias gradient of lin_l out of the same launch; and through SAGEConvFn: every gradient equals the run with the separate pass"""
    ops = P.ops
    old = ops.GEMM_MATH["mode"]
    ops.GEMM_MATH["mode"] = "bf16x3"
    try:
        gen = torch.Generator(device="cuda").manual_seed(8)
        t_rows, n_src, h = 40_032, 70_000, 256
        dz = torch.randn(t_rows, h, device="cuda", generator=gen)
        agg = torch.randn(t_rows, h, device="cuda", generator=gen)
        x = torch.randn(n_src, h, device="cuda", generator=gen)
        rows = torch.randperm(n_src, device="cuda", generator=gen)[:t_rows].sort().values.to(torch.int32)
        cs = []
        c0 = ops.launch_counts()
        gwl, gwr = ops.wgrad_pair(dz, agg, x, rows=rows, x1_compact=True, a_colsum=cs)
        d = _delta(P, c0)
        assert d["gemm_wgrad_wide"] == 1 and len(cs) == 1, d
        ref_l, ref_r = ops.wgrad_pair(dz, agg, x, rows=rows, x1_compact=True)
        assert torch.equal(gwl, ref_l) and torch.equal(gwr, ref_r)
        want = dz.double().sum(0)
        assert float(((cs[0].double() - want).abs() / dz.double().abs().sum(0)).max()) <= 1e-5
    finally:
        ops.GEMM_MATH["mode"] = old


# ------------------------------------------ a dense graph's aggregation on the matrix cores ----
def _dense_graph(P, n=2600, density=0.08, seed=4, dup=True):
    gen = torch.Generator().manual_seed(seed)
    e = int(density * n * n / 2)
    a, b = torch.randint(0, n, (e,), generator=gen), torch.randint(0, n, (e,), generator=gen)
    if dup:                                     # parallel edges: counted twice, like the CSR sum
        a, b = torch.cat([a, a[:5000]]), torch.cat([b, b[:5000]])
    return P.Graph.from_coo(torch.cat([a, b]), torch.cat([b, a]), None, n, n).to("cuda")


def _both_agg_forms(P, fn):
    old = P.ops.DENSE_AGG["enabled"]
    try:
        P.ops.DENSE_AGG["enabled"] = True
        c0 = P.ops.launch_counts()
        dense = fn()
        d = _delta(P, c0)
        P.ops.DENSE_AGG["enabled"] = False
        c0 = P.ops.launch_counts()
        csr = fn()
        d2 = _delta(P, c0)
    finally:
        P.ops.DENSE_AGG["enabled"] = old
    assert d["agg_dense"] >= 1 and d2["agg_dense"] == 0, (d, d2)
    return dense, csr


@pytest.mark.parametrize("feat", [512, 200, 64])
def test_dense_graph_aggregation_on_the_matrix_cores(P, feat):
    """VERDICT r5 #4.  A graph like ogbl-ddi (8 % of all node pairs are edges, every row beyond the CSR kernels' long-row threshold):
    the mean / sum aggregation as counts (bf16, exact) x features (three bf16 terms, two-level f32 accumulation) on the MFMA (csrc/aggregate_dense.hip) against the
    CSR kernels and float64 -- forward mean, plain sum, the mean's backward operator A^T D^-1 (a valued graph whose values depend on
    the column only) with the indexed-addend + gate epilogue, and with the table's Adam step in the epilogue; parallel edges
    count twice; same bits twice; which kernel ran by the launch counters."""
    from plnlp_amd import _lib as L
    g = _dense_graph(P)
    n = g.n_rows
    gen = torch.Generator(device="cuda").manual_seed(feat)
    x = torch.randn(n, feat, device="cuda", generator=gen)
    rr, cc, _ = g.coo()
    a64 = torch.zeros(n, n, dtype=torch.float64, device="cuda")
    a64.view(-1).index_add_(0, rr.long() * n + cc.long(), torch.ones(rr.numel(), dtype=torch.float64, device="cuda"))
    assert float(a64.max()) >= 2.0
    deg = a64.sum(1).clamp_min(1.0)
    for reduce in ("mean", "sum"):
        dense, csr = _both_agg_forms(P, lambda: P.ops.csr_aggregate(g, x, reduce, False))
        want = a64 @ x.double()
        bound = a64 @ x.double().abs()
        if reduce == "mean":
            want, bound = want / deg[:, None], bound / deg[:, None]
        assert float(((dense.double() - want).abs() / bound.clamp_min(1e-30)).max()) <= 1e-6       # f32-grade: 2^-24 per term + the f32 sums
        close(dense, csr, rtol=1e-5, atol=1e-5 * float(want.abs().max()))
        assert torch.equal(dense, P.ops.csr_aggregate(g, x, reduce, False))
    # the mean's backward: gx = A^T D^-1 gagg + addend (the root path), gated by the layer input's relu / dropout
    gagg = torch.randn(n, feat, device="cuda", generator=gen)
    addend = torch.randn(n, feat, device="cuda", generator=gen)
    gate = torch.randn(n, feat, device="cuda", generator=gen)
    mk = lambda: L.make_epilogue(addend=addend, gate=gate, gate_scale=1.25)
    dense, csr = _both_agg_forms(P, lambda: P.ops.csr_aggregate(g.t_mean(), gagg, "sum", True, epilogue=mk()))
    want = torch.where(gate > 0, (a64.t() @ (gagg.double() / deg[:, None]) + addend.double()) * 1.25, torch.zeros(n, feat, dtype=torch.float64, device="cuda"))
    close(dense, want.float(), rtol=1e-5, atol=1e-5 * float(want.abs().max()))
    close(dense, csr, rtol=1e-5, atol=1e-5 * float(want.abs().max()))
    # ... and with the table's Adam step in the epilogue: parameter and moments as the CSR kernel leaves them (to rounding of the gradient)
    outs = []
    for on in (True, False):
        P.ops.DENSE_AGG["enabled"] = on
        table, m, v = x.clone(), torch.zeros_like(x), torch.zeros_like(x)
        e = L.make_epilogue(addend=addend, adam=(m, v, 1, 1e-2, 0.9, 0.999, 1e-8))
        P.ops.csr_aggregate(g.t_mean(), gagg, "sum", True, out=table, epilogue=e)
        outs.append((table, m, v))
    P.ops.DENSE_AGG["enabled"] = True
    grad = (a64.t() @ (gagg.double() / deg[:, None]) + addend.double())
    close(outs[0][1], (0.1 * grad).float(), rtol=1e-5, atol=1e-6 * float(grad.abs().max()))          # m after one step = (1 - beta1) g
    close(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-6 * float(grad.abs().max()))
    live = grad.abs() > 1e-3 * grad.abs().max()           # (Adam's first step is lr * sign(g) up to eps: compare where g is not round-off)
    assert float((outs[0][0] - outs[1][0]).abs()[live].max()) <= 1e-5


def test_dense_form_rule_and_the_sparse_graphs(P):
    """the dense form is taken by dense, mid-sized, unvalued graphs only: a sparse graph, a small one, a valued one (GCN's normalised
    adjacency), a row-restricted or source-mapped launch all stay on the CSR kernels"""
    from gpu_util import rand_csr, to_graph
    import oracle as O
    x = torch.randn(3000, 64, device="cuda")
    sparse = to_graph(P, rand_csr(3000, 20000, 5, weighted=False))
    c0 = P.ops.launch_counts()
    P.ops.csr_aggregate(sparse, x, "mean", False)
    assert _delta(P, c0)["agg_dense"] == 0
    small = _dense_graph(P, n=600, density=0.2, dup=False)
    c0 = P.ops.launch_counts()
    P.ops.csr_aggregate(small, x[:600], "mean", False)
    assert _delta(P, c0)["agg_dense"] == 0
    dense = _dense_graph(P, n=2600, density=0.08, dup=False)
    valued = P.gcn_normalization(dense)
    c0 = P.ops.launch_counts()
    P.ops.csr_aggregate(valued, x[:2600], "sum", True)
    assert _delta(P, c0)["agg_dense"] == 0
    c0 = P.ops.launch_counts()
    P.ops.csr_aggregate(valued, x[:2600], "mean", False)            # (values ignored: the pattern's counts)
    assert _delta(P, c0)["agg_dense"] == 1


def test_one_column_colsum_keeps_its_association(P):
    """plnlp_colsum_f32 at feat = 1 (the 1-output head's bias gradient, 98 us of the ddi step's main stream until its loads were
    pipelined eight deep): partial[b] = x[b] + x[b + blocks] + ... in row order, lane l of the final wave adds partial[l], partial[l
    + 64], ... and the 64 lanes fold in the xor tree 32, 16, .. 1 -- restated here in numpy float32: the kernel's bits."""
    import numpy as np
    from plnlp_amd import _lib
    lib = _lib.load()
    for rows in (262_144, 100_003, 1_500, 7):
        x = torch.randn(rows, generator=torch.Generator().manual_seed(rows)) * 3.0
        got = float(P.ops.colsum(x.cuda().reshape(-1, 1)))
        blocks = int(lib.plnlp_colsum_workspace_floats(rows, 1))
        xs = x.numpy().astype(np.float32)
        partial = np.zeros(blocks, dtype=np.float32)
        for r0 in range(0, rows, blocks):                      # row order inside every partial sum
            seg = xs[r0:r0 + blocks]
            partial[:seg.size] = (partial[:seg.size] + seg).astype(np.float32)
        lanes = np.zeros(64, dtype=np.float32)
        for b0 in range(0, blocks, 64):
            seg = partial[b0:b0 + 64]
            lanes[:seg.size] = (lanes[:seg.size] + seg).astype(np.float32)
        for o in (32, 16, 8, 4, 2, 1):
            lanes = (lanes + lanes[np.arange(64) ^ o]).astype(np.float32)
        assert got == float(lanes[0]), (rows, got, float(lanes[0]))


# ------------------------------------ the table's Adam step in the aggregation that finishes its gradient: [table | features] into a GCN ----
def test_table_adam_in_the_gcn_input_aggregation_gives_the_same_bits(P):
    """citation2's shape of model -- a padded 50-wide table next to constant features into a first GCNConv (ops.GCNInputConvFn): with
    model.FUSE_EMBEDDING_ADAM the transposed 64-wide aggregation that finishes the table's gradient applies the table's Adam step in
    its epilogue (PLNLP_EPI_ADAM; hub rows through the chunk + finalize passes) instead of writing the gradient for the optimiser
    kernel -- the same arithmetic on the same values: losses, every parameter, both moments (padded layout, pad columns zero) and
    the step count after 3 epochs are bit-identical, and emb.weight.grad is never materialised."""
    from plnlp_amd import model as M, synthetic
    n, B, k, e, f, h = 4000, 2048, 3, 50, 16, 200
    g = synthetic.make_graph("collab", seed=19, device="cpu", num_nodes=n, num_edges=30000)
    data = g["data"]
    data.adj_t = P.gcn_normalization(g["adj_t"].to("cuda"))
    assert int((data.adj_t.rowptr[1:] - data.adj_t.rowptr[:-1]).max()) > 256       # a hub row: chunk + finalize passes
    data.x = torch.randn(n, f, generator=torch.Generator().manual_seed(5)).cuda()
    split = {"train": {"edge": g["edges"]}}
    res = {}
    for fused in (True, False):
        M.FUSE_EMBEDDING_ADAM["enabled"] = fused
        try:
            m = P.BaseModel(lr=0.01, dropout=0.3, grad_clip_norm=1.0, gnn_num_layers=2, mlp_num_layers=2, emb_hidden_channels=e,
                            gnn_hidden_channels=h, mlp_hidden_channels=h, num_nodes=n, num_node_feats=f, gnn_encoder_name="GCN",
                            predictor_name="MLP", loss_func="AUC", optimizer_name="Adam", device="cuda", use_node_feats=True,
                            train_node_emb=True)
            torch.manual_seed(21)
            P.manual_seed(21)
            m.param_init()
            assert P.ops.padded_base(m.emb.weight.detach()) is not None
            losses = []
            c0 = P.ops.launch_counts()
            for ep in range(3):
                torch.manual_seed(70 + ep)
                losses.append(m.train(data, split, B, "global", k))
            torch.cuda.synchronize()
            st = m.optimizer.state[m.emb.weight]
            assert st["exp_avg"].shape == (n, 64) and not bool(st["exp_avg"][:, e:].any()) and not bool(st["exp_avg_sq"][:, e:].any())
            assert not bool(P.ops.padded_base(m.emb.weight.detach())[:, e:].any())
            if fused:
                assert m.emb.weight.grad is None
            res[fused] = (losses, [p.detach().clone() for p in m.para_list], st["exp_avg"].clone(), st["exp_avg_sq"].clone(),
                          st["step"])
        finally:
            M.FUSE_EMBEDDING_ADAM["enabled"] = True
    assert res[True][0] == res[False][0]
    assert res[True][4] == res[False][4] and res[True][4] > 0
    for a, b in zip(res[True][1], res[False][1]):
        assert torch.equal(a, b)
    assert torch.equal(res[True][2], res[False][2]) and torch.equal(res[True][3], res[False][3])


# ------------------------------------ the fused scorer's deterministic backward: work split by the number of segments ----
@pytest.mark.parametrize("n,e,feat", [(20_000, 30_000, 200), (20_000, 30_000, 512), (9_000, 200_000, 64), (3_000, 60_000, 512),
                                      (4_267, 100_000, 200), (50, 9_000, 256), (5_000, 80_000, 256), (1_000, 40_000, 512)])
def test_segment_backward_forms(P, n, e, feat):
    """csrc/edge_ops.hip::edge_segment_bwd_group_kernel: one wave per segment makes the launch as long as its longest segment (the hub's
    items are one serial chain).  From 8 192 segments on a workgroup of eight waves takes eight consecutive segments: a wave sums
    its own if it has at most 64 items (item order: the round-5 bits), the longer ones are shared by all eight waves (contiguous
    eighths in order, partial sums added in wave order: another association, a function of the segment's length only); below 8 192
    segments (ddi: 4 267 nodes, hundreds of items at the hubs) every segment is shared by the four waves of its own workgroup -- and
    when the table is beyond one XCD's L2 (4 MB) at 256 / 512 columns, by eight workgroups that take one XCD-pinned column slab each.
    Segments without items (nodes the batch does not touch) come out as zeros; the producing layer's gate in the epilogue; launch
    to launch the same bits; float64-close relative to sum |terms|."""
    from plnlp_amd import _lib as L
    gen = torch.Generator(device="cuda").manual_seed(n + e + feat)
    h = torch.randn(n, feat, device="cuda", generator=gen)
    src = torch.randint(0, n, (e,), device="cuda", generator=gen)
    dst = torch.randint(0, n, (e,), device="cuda", generator=gen)
    src[: e // 20] = 7                                   # a hub
    dst[e // 20: e // 16] = 7
    lonely = torch.arange(11, n, 97, device="cuda")    # nodes without items
    src[torch.isin(src, lonely)] = 3
    dst[torch.isin(dst, lonely)] = 5
    go = torch.randn(e, feat, device="cuda", generator=gen)
    inc = P.ops.Incidence(src, dst, n)
    old = P.ops.EDGE_SEGMENT["form"]
    res = {}
    try:
        for form in ("wave", "auto"):
            P.ops.EDGE_SEGMENT["form"] = form
            plain = P.ops.edge_segment_bwd(h, inc, go)
            gated = P.ops.edge_segment_bwd(h, inc, go, epilogue=L.make_epilogue(gate=h, gate_scale=1.5))
            assert torch.equal(plain, P.ops.edge_segment_bwd(h, inc, go))
            res[form] = (plain, gated)
    finally:
        P.ops.EDGE_SEGMENT["form"] = old
        P.ops._apply_edge_segment()
    want = torch.zeros(n, feat, dtype=torch.float64, device="cuda")
    want.index_add_(0, src, go.double() * h[dst].double())
    want.index_add_(0, dst, go.double() * h[src].double())
    mag = torch.zeros(n, feat, dtype=torch.float64, device="cuda")
    mag.index_add_(0, src, (go.double() * h[dst].double()).abs())
    mag.index_add_(0, dst, (go.double() * h[src].double()).abs())
    for form in ("wave", "auto"):
        plain, gated = res[form]
        assert float(((plain.double() - want).abs() / (mag + 1e-30)).max()) <= 2e-6, form
        assert torch.equal(gated, torch.where(h > 0, plain * 1.5, torch.zeros_like(plain))), form
        assert not bool(plain[lonely].any())
    items = torch.bincount(src, minlength=n) + torch.bincount(dst, minlength=n)
    assert int(items.max()) > 64
    if n >= 8192:                  # a wave per short segment, item order kept: the round-5 bits; the long ones by the workgroup
        short = items <= 64
        assert torch.equal(res["auto"][0][short], res["wave"][0][short])
        assert not torch.equal(res["auto"][0][~short], res["wave"][0][~short])
    else:                          # every segment by the four waves of a workgroup: another association
        assert not torch.equal(res["auto"][0], res["wave"][0])


def test_segment_backward_scalar_rows_keep_the_round5_kernel(P):
    """rows that are not whole 16-byte groups (feat % 4 != 0) stay on the scalar form under every setting"""
    gen = torch.Generator(device="cuda").manual_seed(2)
    n, e, feat = 9000, 20_000, 30
    h = torch.randn(n, feat, device="cuda", generator=gen)
    src = torch.randint(0, n, (e,), device="cuda", generator=gen)
    dst = torch.randint(0, n, (e,), device="cuda", generator=gen)
    go = torch.randn(e, feat, device="cuda", generator=gen)
    inc = P.ops.Incidence(src, dst, n)
    got = P.ops.edge_segment_bwd(h, inc, go)
    want = torch.zeros(n, feat, dtype=torch.float64, device="cuda")
    want.index_add_(0, src, go.double() * h[dst].double())
    want.index_add_(0, dst, go.double() * h[src].double())
    close(got, want.float(), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("n,feat", [(4_267, 512), (6_000, 256), (3_000, 256)])
def test_hadamard_forward_in_xcd_pinned_column_slabs(P, n, feat):
    """a table beyond one XCD's L2 (4 MB) whose rows every edge gathers (ddi: 4 267 x 2 KB): the columns in eight slabs, workgroup b
    on slab b % 8 -- the same products, bit for bit, -1 endpoints (the appended mean row of test()) included; the small table keeps
    the plain kernel"""
    gen = torch.Generator(device="cuda").manual_seed(n)
    h = torch.randn(n, feat, device="cuda", generator=gen)
    src = torch.randint(0, n, (70_001,), device="cuda", generator=gen)
    dst = torch.randint(0, n, (70_001,), device="cuda", generator=gen)
    src[::11] = -1
    old = P.ops.EDGE_SEGMENT["form"]
    try:
        P.ops.EDGE_SEGMENT["form"] = "noslab"
        plain = P.ops.edge_hadamard_fwd(h, src, dst)
        P.ops.EDGE_SEGMENT["form"] = "auto"
        slabs = P.ops.edge_hadamard_fwd(h, src, dst)
    finally:
        P.ops.EDGE_SEGMENT["form"] = old
        P.ops._apply_edge_segment()
    assert torch.equal(plain, slabs)
    assert torch.equal(plain, h[src] * h[dst])
