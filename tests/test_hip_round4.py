"""GPU parity, fourth batch (round 4): the driver's run loop against the oracle's restatement of main.py:228-305
(SURVEY 8 row f1), the exact assembly of the random-walk training pairs, Hits@K in a trained regime that can fail
(tests/trained_parity.py), whole training steps at BASELINE.json's sizes.  Same rules as tests/test_hip_parity.py:
through the C ABI, fp32 tolerance 1e-5 relative, integer outputs bit-exact."""
import contextlib
import io
import os
import re

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


@pytest.fixture(params=["f32", "bf16x3"])
def math(request, P):
    old = P.ops.GEMM_MATH["mode"]
    P.ops.GEMM_MATH["mode"] = request.param
    yield request.param
    P.ops.GEMM_MATH["mode"] = old


# ------------------------------------------------------ random-walk training pairs, exactly ----
def test_random_walk_pairs_are_the_reference_assembly(P):
    """main.py:243-253 fixes the ORDER of the epoch's training pairs (hop-major concatenation, then the self-pair
    mask) -- the order the batch permutation indexes -- and their weights 1 / hop.  ops.random_walk_pairs ==
    oracle.random_walk_pairs_ref on the oracle's own walks: every pair, every weight, in order."""
    csr = rand_csr(700, 2600, 23, weighted=False, hub=300)        # a hub row, an isolated row (walkers stay put there)
    g = to_graph(P, csr)
    gen = torch.Generator().manual_seed(4)
    start = torch.cat([torch.randint(0, 700, (5000,), generator=gen), torch.tensor([5, 5, 3, 3])])
    for seed, L in ((11, 1), (0x1234_5678_9ABC, 4), (7, 10)):
        walk = O.random_walk_ref(csr, start, L, seed)
        want_p, want_w = O.random_walk_pairs_ref(walk, L)
        got_p, got_w = P.ops.random_walk_pairs(g, dev(start), L, seed)
        assert got_p.dtype == torch.int64 and got_w.dtype == torch.float32
        assert torch.equal(got_p.cpu(), want_p), (seed, L)
        assert torch.equal(got_w.cpu(), want_w), (seed, L)
        assert want_p.size(0) < L * start.numel()                  # the mask removed something (isolated starts)


# ---------------------------------------------------------------- f1: the driver's run loop ----
_LINE = re.compile(r"Run: (\d+), Epoch: (\d+), Loss: ([-\d.]+), Learning Rate: ([\d.]+), Valid: ([\d.]+)%, Test: ([\d.]+)%")


def _oracle_dataset(args, split_edge, num_nodes):
    """main.py:112-150 on the oracle side: the year filter, validation edges as input, degree-normalised weights"""
    tr = split_edge["train"]
    prep = O.collab_graph_prep_ref(tr["edge"], tr["weight"], tr["year"], split_edge["valid"]["edge"],
                                   split_edge["valid"]["weight"], num_nodes, year=args.year,
                                   use_valedges_as_input=args.use_valedges_as_input, use_coalesce=args.use_coalesce)
    a = prep["adj"]
    r, c = a.nonzero(as_tuple=True)                                # (row, col) order: what SparseTensor(row, col) holds
    csr = O.CSR.from_coo(r, c, a[r, c].float(), num_nodes, num_nodes)
    split = {"train": {"edge": prep["train_edge"], "weight": prep["train_weight"]},
             "valid": dict(split_edge["valid"]), "test": dict(split_edge["test"])}
    return csr, split


@pytest.mark.parametrize("recipe", ["collab_rw", "ddi_plain"])
def test_driver_run_loop_matches_the_oracle_run_loop(P, recipe, math):
    """train.main([...]) -- the reference's command line on the MI355X path -- against oracle.run_loop_ref, the
    restatement of main.py:235-305, on a small synthetic OGB-shaped set with dropout 0 and the same seeds: per run
    param_init, per epoch a fresh set of random-walk pairs from the SAME walks (collab recipe), model.train, model.test
    every eval_steps, adjust_lr AFTER the epoch, Logger.  Held: the number and order of printed lines; run / epoch /
    learning-rate fields identical; epoch-1 loss at 2e-5 (f32-MFMA products; 1.5e-4 split-bf16: the trajectory floor of
    tests/test_hip_parity.py, a whole epoch of Adam steps lies inside the number), later epochs drift-bounded; Hits within
    a few flipped positives; Logger text identical wherever the results are."""
    import train
    common = ["--epochs=4", "--runs=2", "--eval_steps=2", "--log_steps=1", "--dropout=0", "--lr=0.01", "--seed=5",
              "--emb_hidden_channels=64", "--gnn_hidden_channels=64", "--mlp_hidden_channels=64", "--neg_sampler=local",
              "--data_scale=0.01", "--res_dir=" + os.path.join(ROOT, "gpurun_out", "driver_parity")]
    if recipe == "collab_rw":          # README.md:35
        argv = ["--data_name=ogbl-collab", "--predictor=DOT", "--use_valedges_as_input=True", "--year=2010",
                "--eval_last_best=True", "--gnn_num_layers=1", "--grad_clip_norm=1", "--use_lr_decay=True",
                "--random_walk_augment=True", "--walk_length=3", "--loss_func=WeightedHingeAUC", "--batch_size=8192"]
    else:                              # README.md:24 (lr decay off, MLP scorer, three negatives)
        argv = ["--data_name=ogbl-ddi", "--num_neg=3", "--batch_size=2048", "--data_scale=0.05", "--eval_steps=1",
                "--lr=0.001"]
    argv = common + argv
    args = train.argument(argv)
    device = torch.device("cuda")
    # the same host-side dataset for both sides (load_dataset is seeded; main() would build the identical one)
    data, split_edge, num_nodes = train.load_dataset(args, device)
    data_o, split_o, _ = train.load_dataset(args, device)
    init = []                           # per run: the weights after param_init and the CPU generator state

    def on_run_start(run, model):
        init.append(({k: v.detach().cpu().clone() for k, v in model.encoder.state_dict().items()},
                     {k: v.detach().cpu().clone() for k, v in model.predictor.state_dict().items()},
                     model.emb.weight.detach().cpu().clone(), torch.get_rng_state()))
    hip_losses = []
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        loggers = train.main(argv, dataset=(data, split_edge, num_nodes),
                             hooks={"on_run_start": on_run_start,
                                    "on_epoch": lambda run, epoch, model, loss: hip_losses.append(float(loss))})
    text = buf.getvalue().splitlines()
    hip_lines = [l for l in text if _LINE.match(l)]
    # ---- the oracle's run loop
    h = args.gnn_hidden_channels
    if recipe == "collab_rw":
        csr, split = _oracle_dataset(args, split_o, num_nodes)
        pred = O.DotPredictorRef()
        rw_start = split["train"]["edge"].reshape(-1)             # main.py:229-231: after the graph prep, ONCE (before the loop)
    else:
        adj = data_o.adj_t
        csr = O.CSR(adj.rowptr, adj.col.to(torch.int64), None, num_nodes)
        split = {k: dict(v) for k, v in split_o.items()}
        pred = O.MLPPredictorRef(h, h, 1, args.mlp_num_layers, 0.0)
        rw_start = None
    enc = O.GNNRef("SAGE", h, h, h, args.gnn_num_layers, 0.0)
    emb = torch.nn.Embedding(num_nodes, h)
    trainer = O.TrainerRef(enc, pred, emb, csr, loss_name=args.loss_func, lr=args.lr, clip_norm=args.grad_clip_norm)

    def install(run, tr):
        e, p, w, rng = init[run]
        tr.encoder.load_state_dict(e)
        tr.predictor.load_state_dict(p)
        with torch.no_grad():
            tr.emb.weight.copy_(w)
        torch.set_rng_state(rng)
    ref = O.run_loop_ref(trainer, split, num_nodes=num_nodes, runs=args.runs, epochs=args.epochs,
                         batch_size=args.batch_size, neg_sampler=args.neg_sampler, num_neg=args.num_neg, lr=args.lr,
                         eval_steps=args.eval_steps, log_steps=args.log_steps, use_lr_decay=args.use_lr_decay,
                         eval_metric=args.eval_metric, eval_last_best=args.eval_last_best,
                         random_walk_augment=args.random_walk_augment, walk_length=args.walk_length, rw_adj=csr,
                         rw_start=rw_start, rw_seed=args.seed, on_run_start=install)
    ref_lines = [l for l in ref["lines"] if _LINE.match(l)]
    assert len(hip_losses) == len(ref["losses"]) == args.runs * args.epochs
    assert len(hip_lines) == len(ref_lines) == 3 * args.runs * (args.epochs // args.eval_steps)
    n_valid = split["valid"]["edge"].size(0)
    for e, (a, b) in enumerate(zip(hip_losses, ref["losses"])):
        first = e % args.epochs == 0 and e < args.epochs          # epoch 1 of run 1: no drift yet
        floor = 2e-5 if math == "f32" else 1.5e-4
        assert abs(a - b) <= (floor if first else 2e-3) * abs(b), (recipe, math, e, a, b)
    same_results = True
    for a, b in zip(hip_lines, ref_lines):
        ma, mb = _LINE.match(a).groups(), _LINE.match(b).groups()
        assert ma[0] == mb[0] and ma[1] == mb[1], (a, b)          # run, epoch
        assert ma[3] == mb[3], (a, b)                             # the printed learning rate (main.py:237,288-291)
        assert abs(float(ma[2]) - float(mb[2])) <= 2e-3 * abs(float(mb[2])) + 1e-4, (a, b)
        for i in (4, 5):                                          # Hits in percent: a handful of flipped positives
            assert abs(float(ma[i]) - float(mb[i])) <= 100.0 * 12 / n_valid + 0.011, (a, b)
        same_results &= ma[4:] == mb[4:]
    # Logger text (plnlp/logger.py:16-50): the same lines in the same order; identical text where the results are
    shape = lambda ls: [re.sub(r"[-\d.]+", "#", l) for l in ls]
    stats_hip = [l for l in text if l.startswith(("Run 0", "Highest", "   Final", "All runs"))]
    stats_ref = [l for l in ref["logger_text"] if l.startswith(("Run 0", "Highest", "   Final", "All runs"))]
    assert shape(stats_hip) == shape(stats_ref)
    for a, b in zip(stats_hip, stats_ref):
        na, nb = [float(x) for x in re.findall(r"[-\d.]+", a)], [float(x) for x in re.findall(r"[-\d.]+", b)]
        assert len(na) == len(nb)
        for x, y in zip(na, nb):
            assert abs(x - y) <= 100.0 * 12 / n_valid + 0.011 or a.startswith("Highest Eval Point"), (a, b)
    if same_results:
        assert stats_hip == stats_ref
    assert isinstance(loggers, dict) and set(loggers) == {"Hits@20", "Hits@50", "Hits@100"}


# ------------------------------------------------ Hits@K parity in a TRAINED regime ----
def _oracle_curves(golden, T, recipe, n):
    g11 = golden("g11_trained_curves")
    r = T.RECIPES[recipe]
    np.testing.assert_allclose(g11[f"{recipe}_problem"], [float(v) for v in T.PROBLEMS[r["problem"]].values()])
    np.testing.assert_allclose(g11[f"{recipe}_hyper"], [r["lr"], r["clip"], r["epochs"], r["batch"], r["walk_length"],
                                                        r["k"], float(r["decay"])])
    assert g11[f"{recipe}_f32"].shape[0] >= n
    return (g11[f"{recipe}_f32"][:n].astype(np.float64), g11[f"{recipe}_f64"][:n].astype(np.float64),
            g11[f"{recipe}_f32_loss"][:n])


_trained = {}


def _seeds_of(T, recipe):
    """every seed the oracle fixture holds (64 / 48) per recipe and GEMM form: 0.3 s (collab) / 1.5 s (ddi) per run on an
    MI355X, ~3.5 minutes for the five trained-regime tests; PLNLP_PARITY_SEEDS=32 halves that (the assertions then carry
    the allowance their docstring states)"""
    want = os.environ.get("PLNLP_PARITY_SEEDS", "full")
    return T.RECIPES[recipe]["seeds"] if want == "full" else min(int(want), T.RECIPES[recipe]["seeds"])


def _hip_curves(P, T, recipe, math, n, mutation="none"):
    key = (recipe, math, n, mutation)
    if key not in _trained:
        runs = [T.run_hip(P, recipe, s, math, mutation) for s in range(n)]
        _trained[key] = (np.stack([h for h, _ in runs]), np.stack([l for _, l in runs]))
    return _trained[key]


def _record(text, **arrays):
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "trained_parity_r04.txt"), "a") as f:
            f.write(text + "\n")
        for name, arr in arrays.items():
            np.savez_compressed(os.path.join(out_dir, name + ".npz"), **arr)


@pytest.mark.parametrize("recipe", ["collab", "ddi"])
def test_trained_regime_hits_parity(P, golden, recipe, math):
    """BASELINE.json: "Hits@K within +-0.3 of reference", where it can fail (tests/trained_parity.py).  Each recipe with
    its OWN loss and schedule, 64 / 48 seeds per GEMM form, trained on the HIP path from the initial weights / walks /
    negatives / permutations of the oracle's float32 and float64 runs (fixture g11: their per-seed, per-epoch curves):
      collab (WeightedHingeAUC on random-walk pairs, lr decay) on the problem that does not saturate -- Hits@50 ~ 80 %;
      ddi (SAGE x2 + MLP, AUC) on the block model whose converged runs sit on a 90 % plateau that one seed in ten has
      not reached at Hits@20 after 60 epochs -- the path round 3 could not clear of "slower time-to-plateau".
    Asserted:
      * |final level (HIP) - final level (oracle float32)| <= 0.3 points at the recipe's own K (mean over seeds for
        collab, median for ddi) -- on valid AND on test, outright, with every seed of the fixture trained (the default;
        with PLNLP_PARITY_SEEDS < 48: on the mean of the two splits outright and on each split with 1.5 standard errors
        of the paired difference as allowance);
      * the epochs each seed needs to reach the level are distributed like the float32 oracle's: Mann-Whitney U,
        two-sided, p > 0.05 / 4 (four such comparisons in this suite: family-wise 5 %), and the same share of seeds
        gets there at all;
      * the epoch-1 loss of every seed agrees with the float32 oracle (same walks, negatives, batches): collab at 1e-4;
        ddi -- 24 Adam steps on an MLP scorer inside the number, Adam's sign lottery on its zero-gradient biases
        (profiles/r03_trajectory_drift.txt) -- within 4 x (median) / 1.6 x (worst seed) of the gap between the oracle's
        OWN float32 and float64 runs (round 5; was a flat 5e-3 / 5e-2)."""
    import trained_parity as T
    n = _seeds_of(T, recipe)
    assert n >= 32
    ref32, ref64, loss32 = _oracle_curves(golden, T, recipe, n)
    hip, losses = _hip_curves(P, T, recipe, math, n)
    c = T.compare(hip, ref32, ref64, recipe)
    text = T.describe(f"{recipe} recipe, HIP {math}", c)
    print(text)
    _record(text, **{f"trained_curves_{recipe}_{math}": dict(hits=hip.astype(np.float32), losses=losses)})
    rel = np.abs(losses[:, 0] - loss32[:, 0]) / loss32[:, 0]
    if recipe == "collab":
        assert rel.max() <= 1e-4, rel.max()
    else:
        # round 5: the bound is what the lottery needs, measured -- the ORACLE's own float32 and float64 runs of these 48
        # seeds part by median 6.3e-4 / max 1.86e-2 in this number (fixture g11), the HIP runs by 1.1e-3 / 2.0e-2 from the
        # float32 oracle (profiles/r04_trained_curves_ddi_*.npz); and with the state reset to the oracle's before every step
        # each of the epoch's steps agrees to 2e-7 (tests/test_hip_round5.py, teacher-forced).  Bound: 4 x the oracle's own
        # median, 1.6 x its own worst seed.
        loss64 = golden("g11_trained_curves")[f"{recipe}_f64_loss"][:n]
        own = np.abs(loss32[:, 0] - loss64[:, 0]) / loss64[:, 0]
        assert np.median(rel) <= 4.0 * np.median(own) and rel.max() <= 1.6 * own.max(), (np.median(rel), rel.max(),
                                                                                         np.median(own), own.max())
    if n >= 48:
        assert np.abs(c["diff_f32"]).max() <= 0.3, text
    else:
        assert abs(c["diff_f32"].mean()) <= 0.3, text
        assert (np.abs(c["diff_f32"]) <= 0.3 + 1.5 * c["diff_f32_se"]).all(), text
    assert c["mw_p"] > 0.05 / 4, text
    assert abs(c["reached_hip"] - c["reached_f32"]) <= 0.15, text
    if recipe == "collab":       # nothing saturated, nothing untrained: the level sits where ranking quality decides it
        assert 60.0 < c["final_f64"].min() and c["final_f64"].max() < 90.0, text


def test_trained_regime_harness_rejects_a_degraded_product(P, golden):
    """the assertion above has teeth: the same harness, the same seeds, on a product whose neighbour aggregation
    carries 10 % relative noise (tests/trained_parity.py::Mutation) lands more than 0.3 points from the oracle -- it
    FAILS the check the clean product passes on those seeds.  (What the harness cannot see is recorded too: single-term
    bf16 products and 1e-3 aggregation noise move Hits@K by less than the seed-to-seed spread --
    profiles/r04_trained_parity.md.)"""
    import trained_parity as T
    n = 16
    ref32, ref64, _ = _oracle_curves(golden, T, "collab", n)
    math = P.ops.GEMM_MATH["mode"]
    clean, _ = _hip_curves(P, T, "collab", math, _seeds_of(T, "collab"))
    bad, _ = _hip_curves(P, T, "collab", math, n, mutation="agg_noise:0.1")
    c_bad = T.compare(bad, ref32, ref64, "collab")
    c_clean = T.compare(clean[:n], ref32, ref64, "collab")
    text = T.describe("collab recipe, MUTATED (aggregation noise 0.1)", c_bad) + "\n" + \
        T.describe(f"collab recipe, clean, the same {n} seeds", c_clean)
    print(text)
    _record(text)
    assert np.abs(c_bad["diff_f32"]).max() > 0.3, text
    # and by a margin that is not noise: several standard errors of the paired difference
    assert np.abs(c_bad["diff_f32"] / c_bad["diff_f32_se"]).max() > 3.0, text
    assert np.abs(c_clean["diff_f32"]).max() <= 0.3 + 2 * c_clean["diff_f32_se"].max(), text


# ------------------------------------------------------- whole steps at BASELINE size ----
_c2 = {}


def _c2_problem():
    """BASELINE config 2 at full size: ddi-shaped graph (N = 4 267, nnz = 2 135 822), SAGE x2 h = 512, MLP scorer,
    B = 65 536, k = 3; one step's loss and gradients from the float64 and float32 oracles (cached: ~1 CPU-minute)"""
    if _c2:
        return _c2
    from plnlp_amd import synthetic
    g = synthetic.make_graph("ddi", seed=2, device="cpu")
    n, h, B, k = g["num_nodes"], 512, 65536, 3
    torch.manual_seed(31)
    enc = O.GNNRef("SAGE", h, h, h, 2, 0.0)
    pred = O.MLPPredictorRef(h, h, 1, 2, 0.0)
    emb = torch.nn.Embedding(n, h)
    enc.reset_parameters()
    pred.reset_parameters()
    torch.nn.init.xavier_uniform_(emb.weight)
    state = (enc.state_dict(), pred.state_dict(), emb.weight.detach().clone())
    gen = torch.Generator().manual_seed(8)
    pos = g["edges"][torch.randperm(g["edges"].size(0), generator=gen)[:B]]
    torch.manual_seed(9)
    neg = O.local_neg_sample_ref(pos, n, k)
    adj = g["adj_t"]
    ne = neg.reshape(-1, 2)
    grads = {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        e, p, w = O.GNNRef("SAGE", h, h, h, 2, 0.0), O.MLPPredictorRef(h, h, 1, 2, 0.0), torch.nn.Embedding(n, h)
        e.load_state_dict(state[0])
        p.load_state_dict(state[1])
        with torch.no_grad():
            w.weight.copy_(state[2])
        e, p, w = e.to(dt), p.to(dt), w.to(dt)
        csr = O.CSR(adj.rowptr, adj.col.to(torch.int64), None, n)
        hh = e(w.weight, csr)
        po, no = p(hh[pos[:, 0]], hh[pos[:, 1]]), p(hh[ne[:, 0]], hh[ne[:, 1]])
        loss = O.LOSSES["auc"](po, no, k, None)
        loss.backward()
        grads[name] = dict(loss=loss.detach().double(), pos=po.detach().double().reshape(-1),
                           emb=w.weight.grad.double(),
                           params=[q.grad.double() for q in list(e.parameters()) + list(p.parameters())])
    _c2.update(g=g, n=n, h=h, B=B, k=k, state=state, pos=pos, neg=neg, grads=grads)
    return _c2


def test_full_size_ddi_step_matches_the_oracle(P, math):
    """config C2 at BASELINE size -- the GEMM shapes (M = 262 144 x 512 x 512 scorer, M = 4 267 concat-K encoder,
    split-K weight gradients over 262 144 rows), the 2.1 M-entry aggregation and the B = 65 536 edge kernels the
    benchmark times: ONE full-size step's loss, scores and EVERY gradient against the float64 oracle, the float32
    oracle's own distance to it as the yardstick."""
    c = _c2_problem()
    n, h, k = c["n"], c["h"], c["k"]
    m = P.BaseModel(lr=1e-3, dropout=0.0, grad_clip_norm=2.0, gnn_num_layers=2, mlp_num_layers=2,
                    emb_hidden_channels=h, gnn_hidden_channels=h, mlp_hidden_channels=h, num_nodes=n, num_node_feats=0,
                    gnn_encoder_name="SAGE", predictor_name="MLP", loss_func="AUC", optimizer_name="Adam", device="cuda",
                    use_node_feats=False, train_node_emb=True)
    m.encoder.load_state_dict(c["state"][0])
    m.predictor.load_state_dict(c["state"][1])
    with torch.no_grad():
        m.emb.weight.copy_(c["state"][2])

    class Data:
        pass
    data = Data()
    data.adj_t = c["g"]["adj_t"].to("cuda")
    m.encoder.train()
    m.predictor.train()
    pos, ne = c["pos"], c["neg"].reshape(-1, 2)
    hh = m.encoder(m.create_input_feat(data), data.adj_t)
    src = torch.cat([pos[:, 0], ne[:, 0]]).cuda()
    dst = torch.cat([pos[:, 1], ne[:, 1]]).cuda()
    out = m._score(hh, src, dst)
    out.retain_grad()
    B = pos.size(0)
    loss = m.calculate_loss(out[:B], out[B:], k)
    loss.backward()
    g64, g32 = c["grads"]["f64"], c["grads"]["f32"]
    close(loss, g64["loss"], rtol=1e-5)
    # relu's kink.  Of the scorer's 134 M hidden pre-activations z = <x, W1 row> a handful cancel to within float32 round-off of
    # their own terms (|z| <= 2^-23 sum_k |x_k W_k|; measured on this problem: 3 elements, one of them at 7e-11 of that sum): which
    # side of zero such an element lands on is decided by the order of the additions -- any float32 evaluation, the reference's
    # included, may go either way -- and the side decides whether the element's whole contribution g_r w2_c enters the hidden
    # layer's bias gradient (a positive pair at initialisation: 6 x |w2_c| ~ 0.25, against a bound of 0.03 on a sum of scale 122).
    # Those elements are named here, in float64 from the model's own tensors, and their contributions are added to the bound of
    # the column they sit in; every other column keeps the plain bound.  (profiles/r06_ddi_dense_agg.txt, part 4: how this was found.)
    with torch.no_grad():
        l0, l1 = m.predictor.lins[0], m.predictor.lins[1]
        x64 = (hh[src] * hh[dst]).double()
        z64 = x64 @ l0.weight.double().t() + l0.bias.double()
        on_kink = z64.abs() <= 2.0 ** -23 * (x64.abs() @ l0.weight.double().abs().t())
        kink = (on_kink * (out.grad.double().abs().reshape(-1, 1) * l1.weight.double().abs().reshape(1, -1))).sum(0).cpu()
        assert int(on_kink.sum()) <= 16, int(on_kink.sum())          # (an exception, not a loophole)
        del x64, z64, on_kink
    yard = float((g32["pos"] - g64["pos"]).abs().max())
    assert float((out[:B].reshape(-1).cpu().double() - g64["pos"]).abs().max()) <= 4 * yard + 1e-5
    named = [("emb", m.emb.weight.grad)] + [(key, q.grad) for key, q in list(m.encoder.named_parameters()) +
                                            list(m.predictor.named_parameters())]
    refs64 = [g64["emb"]] + g64["params"]
    refs32 = [g32["emb"]] + g32["params"]
    assert len(named) == len(refs64)
    for (key, got), r64, r32 in zip(named, refs64, refs32):
        err = float((got.cpu().double() - r64).abs().max())
        yard = float((r32 - r64).abs().max())
        scale = float(r64.abs().max())
        # f32-MFMA products: 1e-5 of the gradient's scale, or a few times what the reference's own float32 arithmetic
        # leaves.  Split-bf16 products: 1e-4 -- the dropped 2^-24 residual of an operand element is the SAME in every
        # product it enters, so over reductions of 10^3 .. 10^5 terms its error adds up coherently where independent
        # roundings average out (measured here: 9e-5 of the scale on the first layer's bias gradient).
        # + the cancellation floor: the last bias's gradient is EXACTLY 0 (sum over pairs of +g and -g); in any fp32
        # arithmetic what is left is rounding noise of a sum of 786 432 O(1) terms
        rel = 1e-5 if math == "f32" else 1e-4
        floor = 1e-8 * float(2 * c["B"] * (1 + c["k"]))
        if key == "lins.0.bias":            # per column, with the kink elements' contributions (above)
            errs = (got.cpu().double() - r64).abs()
            over = errs > max(rel * scale, 4 * yard) + floor + kink
            assert not bool(over.any()), (key, over.nonzero().reshape(-1).tolist(), float(errs.max()), yard, scale)
            continue
        assert err <= max(rel * scale, 4 * yard) + floor, (key, err, yard, scale)


def _workload_model(P, name, seed=1234):
    import bench
    from plnlp_amd import synthetic
    cfg = bench.WORKLOADS[name]
    device = torch.device("cuda")
    torch.manual_seed(seed)
    P.manual_seed(seed)
    g = synthetic.make_graph(cfg["shape"], seed=2, device=device, weighted=cfg["weighted"])
    n = g["num_nodes"]
    data = g["data"]
    if cfg["encoder"] == "GCN":
        g["adj_t"] = data.adj_t = P.gcn_normalization(g["adj_t"])
    feats = cfg.get("feats", 0)
    if feats:
        data.x = torch.randn(n, feats, device=device, generator=torch.Generator(device=device).manual_seed(5))

    def make():
        torch.manual_seed(seed)
        P.manual_seed(seed)
        m = P.BaseModel(lr=1e-3, dropout=cfg["dropout"], grad_clip_norm=cfg["clip"], gnn_num_layers=cfg["gnn_layers"],
                        mlp_num_layers=cfg["mlp_layers"], emb_hidden_channels=cfg.get("emb", cfg["hidden"]),
                        gnn_hidden_channels=cfg["hidden"], mlp_hidden_channels=cfg["hidden"], num_nodes=n,
                        num_node_feats=feats, gnn_encoder_name=cfg["encoder"], predictor_name=cfg["predictor"],
                        loss_func=cfg["loss"], optimizer_name="Adam", device=device, use_node_feats=feats > 0,
                        train_node_emb=True)
        torch.manual_seed(seed + 1)
        m.param_init()
        m.encoder.train()
        m.predictor.train()
        return m
    B, k = cfg["batch"], cfg["num_neg"]
    gen = torch.Generator(device=device).manual_seed(777)
    steps = 3
    if cfg["shape"] == "collab":
        pairs, weights = P.ops.random_walk_pairs(g["adj_t"], g["edges"].reshape(-1), 2, 777)
        sel = torch.randperm(pairs.size(0), generator=gen, device=device)[: steps * B]
        pos, w = pairs[sel], weights[sel]
    else:
        sel = torch.randint(0, g["edges"].size(0), (steps * B,), generator=gen, device=device)
        pos, w = g["edges"][sel], None
    torch.manual_seed(99)
    neg = P.negative_sample.local_neg_sample(pos.cpu(), n, k).to(device)
    return cfg, data, make, pos, neg, w, steps, B, k


@pytest.mark.parametrize("name", ["collab", "citation2"])
def test_full_size_steps_are_deterministic_and_forms_agree(P, name):
    """configs C3 / C4 at BASELINE size (N = 235 868 h = 256 / N = 2 927 963 h = 200, B = 65 536): three full training
    steps -- the recipe's dropout on -- (a) twice from the same state: losses and EVERY parameter bit-identical (no
    atomics anywhere, fixed split-K order, counter-based dropout); (b) with the last layer evaluated at the touched rows
    only vs on the full matrix: the same loss bits at every step; (c) the epoch's running loss accumulates what the
    steps returned."""
    cfg, data, make, pos, neg, w, steps, B, k = _workload_model(P, name)

    def run(sparse_forward):
        old = P.ops.SPARSE_FORWARD["enabled"]
        P.ops.SPARSE_FORWARD["enabled"] = sparse_forward
        try:
            m = make()
            losses = []
            for i in range(steps):
                sl = slice(i * B, (i + 1) * B)
                losses.append(m.train_step(data, pos[sl], neg[sl], k, None if w is None else w[sl], edges_ready=True))
            torch.cuda.synchronize()
            return [float(l) for l in losses], [p.detach().clone() for p in m.para_list]
        finally:
            P.ops.SPARSE_FORWARD["enabled"] = old
    l1, p1 = run(True)
    l2, p2 = run(True)
    assert l1 == l2, (l1, l2)
    assert all(np.isfinite(l1)) and l1[0] > 0
    for a, b in zip(p1, p2):
        assert torch.equal(a, b)
    del p2
    l3, p3 = run(False)
    assert l1 == l3, (l1, l3)                  # row-sparse last layer == full matrix, bit for bit, at full size
    for a, b in zip(p1, p3):
        assert torch.equal(a, b)


def test_full_size_split_k_weight_gradient_against_fp64(P, math):
    """the weight gradient of the collab step at BASELINE size: [dWl | dWr] = dz^T [agg | x] over T = 132 K gathered rows
    (deterministic split-K + fixed-order reduce).  Sampled tiles of the result against float64: every sampled element
    within the fp32 summation bound of its own column of products."""
    gen = torch.Generator(device="cuda").manual_seed(12)
    T_rows, n_src, h = 132_224, 235_868, 256
    dz = torch.randn(T_rows, h, device="cuda", generator=gen) * 0.05
    agg = torch.randn(T_rows, h, device="cuda", generator=gen)
    x = torch.randn(n_src, h, device="cuda", generator=gen)
    rows = torch.randperm(n_src, device="cuda", generator=gen)[:T_rows].sort().values.to(torch.int32)
    d_l, d_r = P.ops.wgrad_pair(dz, agg, x, rows=rows, x1_compact=True)
    d_l2, d_r2 = P.ops.wgrad_pair(dz, agg, x, rows=rows, x1_compact=True)
    assert torch.equal(d_l, d_l2) and torch.equal(d_r, d_r2)          # fixed reduction order
    xs = x[rows.long()]
    for got, b in ((d_l, agg), (d_r, xs)):
        for (i0, j0) in ((0, 0), (96, 160), (200, 64), (128, 224)):
            a64 = dz[:, i0:i0 + 32].double()
            b64 = b[:, j0:j0 + 32].double()
            want = a64.t() @ b64
            bound = a64.abs().t() @ b64.abs()
            err = (got[i0:i0 + 32, j0:j0 + 32].double() - want).abs()
            # fp32 accumulation over 132 K terms in split-K slices: a few 1e-7 of sum |a||b| (the error test of
            # tests/test_hip_round2.py holds the per-product bound; this is the full-size sum)
            assert float((err / bound).max()) <= 1e-6, (i0, j0, float((err / bound).max()))


# ------------------------------------------- the stationary-weights GEMM (csrc/gemm_x3s.hip) ----
def _both_forms(P, fn):
    """fn() with the stationary-weights kernel and with the 128 x 128 kernels (split-bf16 products either way)"""
    old_math, old_st = P.ops.GEMM_MATH["mode"], P.ops.GEMM_STATIONARY_B["enabled"]
    P.ops.GEMM_MATH["mode"] = "bf16x3"
    try:
        P.ops.GEMM_STATIONARY_B["enabled"] = True
        new = fn()
        P.ops.GEMM_STATIONARY_B["enabled"] = False
        ref = fn()
    finally:
        P.ops.GEMM_MATH["mode"], P.ops.GEMM_STATIONARY_B["enabled"] = old_math, old_st
    return new, ref


def _same_to_rounding(new, ref, a, b_nk):
    """two summation orders of the same six products per K-step: equal to a few f32 roundings of sum |a||b|"""
    bound = a.abs().double().sum(1, keepdim=True) * float(b_nk.abs().max())
    assert float(((new.double() - ref.double()).abs() / bound.clamp_min(1e-30)).max()) <= 2e-7


@pytest.mark.parametrize("m,n,k", [(132_224, 256, 256), (132_224, 512, 256), (70_001, 200, 180), (20_000, 64, 100),
                                   (17_000, 128, 36), (262_144, 512, 512), (40_000, 224, 200), (16_984, 16, 16),
                                   (66_000, 256, 64)])
def test_stationary_weights_gemm_against_fp64_and_the_tile_kernel(P, m, n, k):
    """the stationary-weights kernel forms the SAME six bf16 products per K-step of 16 as the 128 x 128 split-bf16 kernel
    (same split, small terms first) -- the tile kernel walks the second 32-row half of its wave tile in another product
    order, so the two agree to rounding, not bit for bit.  Held here, for weights stored [N, K] and [K, N], ragged K
    (k % 16 != 0), ragged N, every column-tile width, a last round split off as a narrow-tile tail launch (132 224 rows =
    1 033 panels on 512 slots): f32-grade distance from float64 (the 2^-22 bound of tests/test_hip_round2.py), agreement
    with the tile kernel to rounding, launch-to-launch bit-determinism, and the identical dropout mask."""
    from plnlp_amd import _lib as L
    gen = torch.Generator(device="cuda").manual_seed(m + n + k)
    a = torch.randn(m, k, device="cuda", generator=gen)
    w_nk = torch.randn(n, k, device="cuda", generator=gen) * 0.1
    w_kn = w_nk.t().contiguous()
    bias = torch.randn(n, device="cuda", generator=gen)
    outs = []
    for b, bt in ((w_nk, True), (w_kn, False)):
        new, ref = _both_forms(P, lambda: P.ops.gemm([(a, b)], False, bt))
        _same_to_rounding(new, ref, a, w_nk)
        outs.append(new)
    assert torch.equal(outs[0], outs[1])              # the weight's storage layout does not change a bit: one image
    old = P.ops.GEMM_MATH["mode"]
    P.ops.GEMM_MATH["mode"] = "bf16x3"
    try:
        again = P.ops.gemm([(a, w_nk)], False, True)
        assert torch.equal(again, outs[0])
        # a row's bits do not depend on which launch (main / narrow-tile tail) or which panel computed it
        lo = max(0, m - 16_384 - 640)
        sub = P.ops.gemm([(a[lo:], w_nk)], False, True)
        assert torch.equal(sub, outs[0][lo:])
        epi = L.Epilogue()
        epi.flags = L.EPI_BIAS | L.EPI_RELU | L.EPI_DROPOUT
        epi.bias = bias.data_ptr()
        epi.dropout_p, epi.dropout_seed = 0.3, 0x1234_5678_9ABC
        new, ref = _both_forms(P, lambda: P.ops.gemm([(a, w_nk)], False, True, epilogue=epi))
        P.ops.GEMM_MATH["mode"] = "bf16x3"
        live = torch.relu(outs[0] + bias) > 1e-4      # away from the relu kink the two forms keep / drop the same elements
        assert torch.equal((new != 0)[live], (ref != 0)[live])
        assert 0.2 < float((new == 0).float().mean()) < 0.9
        _same_to_rounding(new, ref, a, w_nk)
    finally:
        P.ops.GEMM_MATH["mode"] = old
    rows = torch.randint(0, m, (256,), device="cuda", generator=gen)
    want = a[rows].double() @ w_nk.double().t()
    bound = a[rows].double().abs() @ w_nk.double().abs().t()
    # per product 2^-22 (tests/test_hip_round2.py); the f32 accumulation over K adds its own roundings on top
    assert float(((outs[0][rows].double() - want).abs() / bound).max()) <= 5e-7


def test_stationary_weights_gemm_step_forms(P):
    """the launches of a training step that take this kernel, each against the tile kernel (to rounding) and float64: the
    conv at the touched rows (two K-segments, the root operand's rows GATHERED, bias + relu + dropout drawn at the
    original rows), the pair of data gradients ([gx | gagg] = dz [Wr | Wl]: B from two [K, N] buffers, the result into
    two tensors), an accumulate epilogue and a gate epilogue"""
    from plnlp_amd import _lib as L
    gen = torch.Generator(device="cuda").manual_seed(5)
    t_rows, n_src, h = 33_333, 60_000, 256
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
    new, ref = _both_forms(P, lambda: P.ops.gemm([(agg, w_l), (x, w_r)], False, True, epilogue=epi, a_index=[None, rows]))
    want = torch.relu(agg.double() @ w_l.double().t() + x[rows.long()].double() @ w_r.double().t() + bias.double())
    live = want > 1e-4
    assert torch.equal((new != 0)[live], (ref != 0)[live])          # the same mask, drawn at the ORIGINAL row positions
    kept = (new != 0) & live
    close(new[kept], (want / 0.7).float()[kept], rtol=1e-5, atol=1e-4)
    close(new, ref, rtol=1e-5, atol=1e-5)
    dz = torch.randn(t_rows, h, device="cuda", generator=gen)
    (n1, n2), (r1, r2) = _both_forms(P, lambda: P.ops.dgrad_pair(dz, w_r, w_l))
    close(n1, r1, rtol=1e-5, atol=1e-5)
    close(n2, r2, rtol=1e-5, atol=1e-5)
    close(n1, (dz.double() @ w_r.double()).float(), rtol=1e-5, atol=1e-4)
    close(n2, (dz.double() @ w_l.double()).float(), rtol=1e-5, atol=1e-4)
    # accumulate into an existing result; gate by another matrix
    base = torch.randn(t_rows, h, device="cuda", generator=gen)
    gate = torch.randn(t_rows, h, device="cuda", generator=gen)
    for flags in (L.EPI_ACCUM, L.EPI_GATE):
        e2 = L.Epilogue()
        e2.flags = flags
        if flags == L.EPI_GATE:
            e2.gate, e2.ld_gate, e2.gate_scale = gate.data_ptr(), h, 1.25
        new, ref = _both_forms(P, lambda: P.ops.gemm([(dz, w_l)], False, False, out=base.clone(), epilogue=e2))
        close(new, ref, rtol=1e-5, atol=1e-5)
        want = dz.double() @ w_l.double()
        want = want + base.double() if flags == L.EPI_ACCUM else torch.where(gate > 0, want * 1.25, torch.zeros_like(want))
        close(new, want.float(), rtol=1e-5, atol=1e-4)


# ------------------------------------------------ config 5 at ONE RANK'S TRUE SHARE ----
def test_config5_one_ranks_true_share_of_the_full_problem(P):
    """BASELINE config 5 as an 8-GPU node meets it: rank 3 of 8 of the R-MAT 50 M-node / 1 B-edge graph -- its 6.25 M
    destination rows gathering from the FULL replicated source X [50 M, 512] (102 GB, generated in place; every
    --scale run of rounds 2-3 stopped at a 25 GB source).  Size-independent properties of the aggregation at that
    geometry (offsets beyond 2^32 bytes, 18x the source behind the default `roofline`):
      * the row block replays the stream: windows of plnlp_rmat_edges' stream, restated by oracle.rmat_edges_ref, land in
        the block's CSR rows exactly where the oracle puts them;
      * a constant source is a fixed point of the mean (empty rows give 0), bit for bit;
      * checksum of checksums: sum_r deg_r agg[r, :] == sum_j count_j x[j, :] in float64 (count = column multiplicity);
      * sampled rows (the longest one included) against a float64 gather-mean at 1e-5; launch-to-launch determinism."""
    from plnlp_amd import shard, synthetic
    n, nnz, F, world, rank = 50_000_000, 1_000_000_000, 512, 8, 3
    part = shard.RowPartition(n, world, rank)
    S, npad = part.rows, part.padded
    blk = synthetic.rmat_row_block(26, nnz, n, part.lo, S, npad, "cuda", seed=11)
    rowptr, col = blk.rowptr, blk.col
    assert rowptr.numel() == S + 1 and int(rowptr[0]) == 0 and int(rowptr[-1]) == col.numel() == blk.nnz
    deg = rowptr[1:] - rowptr[:-1]
    assert int(deg.min()) >= 0 and int(col.min()) >= 0 and int(col.max()) < n
    assert 0.05 * nnz < blk.nnz < 0.20 * nnz                      # an eighth of the edges, give or take the skew
    # windows of the stream against the oracle
    for edge_lo in (0, 123_456_789, 999_800_000):
        rr, cc = O.rmat_edges_ref(26, n, edge_lo, 200_000, 11)
        mine = (rr >= part.lo) & (rr < part.lo + S)
        r_l, c_l = torch.from_numpy(rr[mine] - part.lo).cuda(), torch.from_numpy(cc[mine]).cuda().to(torch.int32)
        assert r_l.numel() > 10_000
        # each such edge sits in its CSR row (rows are sorted by column: binary search inside the row's segment)
        lo, hi = rowptr[r_l], rowptr[r_l + 1]
        pos = lo.clone()
        step = int(deg.max())
        span = 1 << (max(step, 1) - 1).bit_length()
        while span >= 1:                                           # branch-free lower_bound over [lo, hi)
            probe = pos + span - 1
            ok = (probe < hi) & (col[probe.clamp(max=col.numel() - 1)] < c_l)
            pos = torch.where(ok, probe + 1, pos)
            span >>= 1
        assert bool(((pos < hi) & (col[pos.clamp(max=col.numel() - 1)] == c_l)).all()), edge_lo
    x = torch.empty(npad, F, device="cuda")
    agg = torch.empty(S, F, device="cuda")
    # constant fixed point
    x.fill_(1.25)
    P.ops.csr_aggregate(blk, x, "mean", False, out=agg)
    want = torch.where(deg > 0, 1.25, 0.0).to(torch.float32)
    assert torch.equal(agg, want[:, None].expand(S, F))
    # random source, generated in place
    gen = torch.Generator(device="cuda").manual_seed(12)
    for lo_ in range(0, npad, 1 << 20):
        x[lo_:lo_ + (1 << 20)].normal_(generator=gen)
    P.ops.csr_aggregate(blk, x, "mean", False, out=agg)
    again = torch.empty_like(agg)
    P.ops.csr_aggregate(blk, x, "mean", False, out=again)
    assert torch.equal(agg, again)
    del again
    count = torch.bincount(col.long(), minlength=npad).double()
    lhs = torch.zeros(F, dtype=torch.float64, device="cuda")
    rhs = torch.zeros(F, dtype=torch.float64, device="cuda")
    mag = torch.zeros(F, dtype=torch.float64, device="cuda")
    for lo_ in range(0, S, 1 << 20):
        lhs += (agg[lo_:lo_ + (1 << 20)].double() * deg[lo_:lo_ + (1 << 20), None].double()).sum(0)
    for lo_ in range(0, npad, 1 << 20):
        xs = x[lo_:lo_ + (1 << 20)].double()
        cs = count[lo_:lo_ + (1 << 20), None]
        rhs += (xs * cs).sum(0)
        mag += (xs.abs() * cs).sum(0)
    assert float(((lhs - rhs).abs() / mag).max()) <= 1e-7
    # sampled rows, the hub included
    pick = torch.cat([deg.argmax().reshape(1), torch.randint(0, S, (63,), device="cuda", generator=gen)])
    for r in pick.tolist():
        cols = col[int(rowptr[r]):int(rowptr[r + 1])].long()
        ref = x[cols].double().mean(0) if cols.numel() else torch.zeros(F, dtype=torch.float64, device="cuda")
        scale = float(x[cols].double().abs().mean()) if cols.numel() else 1.0
        assert float((agg[r].double() - ref).abs().max()) <= 1e-5 * max(scale, 1e-3), r


@contextlib.contextmanager
def _stale_memory_is_nan():
    """every float tensor the host side takes from torch.empty / torch.empty_like on the GPU arrives full of NaN (what a
    recycled allocator block may hold): a kernel that leaves part of its result unwritten, or reads a workspace it did
    not fill, then shows up as NaN in the parameters instead of passing on whatever the block happened to contain"""
    empty, empty_like = torch.empty, torch.empty_like

    def fill(t):
        if t.is_cuda and t.is_floating_point() and t.numel():
            t.fill_(float("nan"))
        return t
    torch.empty = lambda *a, **k: fill(empty(*a, **k))
    torch.empty_like = lambda *a, **k: fill(empty_like(*a, **k))
    try:
        yield
    finally:
        torch.empty, torch.empty_like = empty, empty_like


@pytest.mark.parametrize("name", ["collab", "ddi", "citation2"])
def test_full_size_steps_do_not_depend_on_stale_memory(P, name):
    """three BASELINE-size training steps with every freshly allocated float buffer pre-filled with NaN: the losses and
    EVERY parameter are the bits of the plain run.  (Round 4 found one dependence this way: the padding rows of the
    row-restricted aggregate -- CSR row 0 named again -- were left unwritten when node 0 is a long row, and the
    weight-gradient GEMM reduces over them against zero gradient rows: 0 x stale bits.)"""
    cfg, data, make, pos, neg, w, steps, B, k = _workload_model(P, name)

    def run():
        m = make()
        losses = []
        for i in range(steps):
            sl = slice(i * B, (i + 1) * B)
            losses.append(m.train_step(data, pos[sl], neg[sl], k, None if w is None else w[sl], edges_ready=True))
        torch.cuda.synchronize()
        return [float(l) for l in losses], [p.detach().clone() for p in m.para_list]
    l1, p1 = run()
    with _stale_memory_is_nan():
        l2, p2 = run()
    assert all(np.isfinite(l1)) and l1 == l2, (l1, l2)
    for a, b in zip(p1, p2):
        assert torch.isfinite(b).all()
        assert torch.equal(a, b)


@pytest.mark.parametrize("feat", [64, 256])
def test_row_indexed_aggregation_writes_every_result_row(P, feat):
    """plnlp_csr_aggregate_f32 with row_index naming a LONG row more than once (CompactIncidence pads its row list with row
    id 0): the split pass writes the one result row out_map names; the others are zero, never left as they were"""
    from plnlp_amd import ops
    n, hub_deg = 6000, 3000
    gen = torch.Generator().manual_seed(5)
    deg = torch.randint(1, 12, (n,), generator=gen)
    deg[0] = hub_deg                                                # node 0 is a long row
    deg[17] = hub_deg
    row = torch.repeat_interleave(torch.arange(n), deg)
    col = torch.randint(0, n, (row.numel(),), generator=gen)
    graph = to_graph(P, O.CSR.from_coo(row, col, None, n, n))
    x = dev(torch.randn(n, feat, generator=gen))
    full = ops.csr_aggregate(graph, x, "mean", use_values=False)
    assert graph.row_split(ops.split_threshold(n)).n_long >= 2
    for with_hub in (True, False):
        touched = torch.tensor(([0] if with_hub else []) + [3, 17, 40, 41, 999], dtype=torch.int32)
        t = touched.numel()
        rows = dev(torch.cat([touched, torch.zeros(32 - t, dtype=torch.int32)]))
        out_map = torch.full((n,), -1, dtype=torch.int32)
        out_map[touched.long()] = torch.arange(t, dtype=torch.int32)
        out = torch.full((32, feat), float("nan"), device="cuda")
        ops.csr_aggregate(graph, x, "mean", use_values=False, row_index=rows, out_map=dev(out_map), out=out)
        assert torch.equal(out[:t], full[dev(touched.long())])
        assert torch.equal(out[t:], torch.zeros_like(out[t:]))


def test_deferred_concat_is_written_only_when_it_is_read(P):
    """create_input_feat with a GCN encoder (ops.concat_features(defer=True)): a first GCNConv that takes the parts
    (GCNInputConvFn) never reads the concatenated [emb | x] matrix, so the per-step copy of the embedding block is skipped --
    the buffer's embedding columns stay STALE -- and the output, the embedding gradient and the weight gradients are the bits
    of the materialised path; a consumer that does read the matrix (input fusion off; the row-sharded block conv) gets it
    written first"""
    from plnlp_amd import ops
    torch.manual_seed(11)
    n, e, f, h = 900, 50, 128, 64
    g = to_graph(P, O.gcn_norm_csr(rand_csr(n, 9000, 17, weighted=False)))
    enc = P.GCN(e + f, h, h, 2, 0.0).cuda()
    feats = torch.randn(n, f).cuda()
    emb = torch.randn(n, e).cuda().requires_grad_(True)
    go = torch.randn(n, h).cuda()

    def run(defer, cache):
        for p_ in list(enc.parameters()) + [emb]:
            p_.grad = None
        x = ops.concat_features(emb, feats, cache, defer=defer)
        out = enc(x, g)
        out.backward(go)
        return x, out.detach().clone(), emb.grad.clone(), [p_.grad.clone() for p_ in enc.parameters()]

    c0, c1 = {}, {}
    x0, out0, ge0, gw0 = run(False, c0)
    with torch.no_grad():
        ops.concat_features(emb, feats, c1)                            # the buffer exists and holds the CURRENT embedding ...
        emb_before = emb.detach().clone()
        emb.add_(0.5)                                                    # ... which then moves on
    x0, out0, ge0, gw0 = run(False, c0)
    x1, out1, ge1, gw1 = run(True, c1)
    assert getattr(x1, "_plnlp_stale", False)
    assert torch.equal(out0, out1) and torch.equal(ge0, ge1)
    for a, b in zip(gw0, gw1):
        assert torch.equal(a, b)
    # the deferred buffer was NOT rewritten: its embedding block is still the old embedding
    assert torch.equal(c1["buf"][:, :e], emb_before) and not torch.equal(emb_before, emb.detach())
    assert torch.equal(c0["buf"][:, :e], emb.detach())
    # a reader of the matrix itself gets it materialised
    xm = ops.materialize_concat(x1)
    assert not getattr(xm, "_plnlp_stale", False) and torch.equal(xm, torch.cat([emb.detach(), feats], -1))
    old = ops.GCN_INPUT_FUSION["enabled"]
    try:
        ops.GCN_INPUT_FUSION["enabled"] = False
        with torch.no_grad():
            emb.add_(0.25)
        out_plain = enc(ops.concat_features(emb, feats, c0), g)
        out_lazy = enc(ops.concat_features(emb, feats, c1, defer=True), g)      # GCNConv.forward materialises it
    finally:
        ops.GCN_INPUT_FUSION["enabled"] = old
    assert torch.equal(out_plain, out_lazy)
