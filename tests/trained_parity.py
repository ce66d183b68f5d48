"""Hits@K parity in a TRAINED regime (BASELINE.json: "Hits@K within +-0.3 of reference"; VERDICT r2 #2c).

The problem is learnable by construction (plnlp_amd.synthetic.community_graph: a stochastic block model whose
held-out intra-community edges a trained encoder ranks far above random non-edges), so Hits@K lands where the
reference reports its own numbers (README.md:7-10: 70-91 %), not at the few percent of a random graph where the
K-th of 10 000 negatives decides everything.

Four arithmetics train the SAME problem from the SAME initial weights with the SAME negatives and batch permutations,
over several seeds (the reference reports mean +- std over 10 runs, main.py:43):
    HIP split-bf16 GEMMs (the product's default), HIP f32-MFMA GEMMs,
    the CPU oracle in float32 (the reference's arithmetic), the CPU oracle in float64 (the arbiter).
Test infrastructure: imports the oracle; used by tests/test_hip_round3.py and scripts/hits_parity_table.py only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RECIPES = {
    # README.md:35 (collab): SAGE x1 (+ relu) + DOT, one negative per positive.  Loss: the squared AUC loss, not the
    # recipe's hinge -- the hinge run never settles on this problem (it keeps oscillating around Hits@50 ~ 55-65 %:
    # over 10 seeds the ORACLE's own float32 and float64 means differ by 4 points, profiles/r03_trained_parity.md),
    # so no +-0.3 statement can be made there about any arithmetic; the AUC loss converges to the data's ceiling
    "collab": dict(layers=1, predictor="DOT", loss="AUC", k=1, metric="Hits@50", lr=0.02, clip=10.0, epochs=30),
    # README.md:24 (ddi): SAGE x2 + MLP predictor, AUC loss, three negatives per positive.  The MLP recipe needs a
    # gentler optimiser to CONVERGE on every seed (at lr 0.02 two or three of ten seeds are still climbing or have
    # diverged after 22 epochs -- in float32 and float64 alike, different seeds in each): lr 0.005, the reference's
    # default clip of 2, 60 epochs (at 40, one or two of ten seeds are still a few points short of the plateau at
    # Hits@20 / Hits@50 -- in every arithmetic, different seeds in each)
    "ddi": dict(layers=2, predictor="MLP", loss="AUC", k=3, metric="Hits@20", lr=0.005, clip=2.0, epochs=60),
}
# 10 % of the valid / test positives are random non-edges no model can rank: a CONVERGED model sits at the data's
# ceiling of 90 % (the reference's own numbers are such plateaus: 90.9 % on ddi), stable to a few hundredths of a point
PROBLEM = dict(num_nodes=2000, community=50, p_in=0.97, cross_per_node=1.0, seed=3, unlearnable=0.1)
H, B = 64, 2048


def problem():
    from plnlp_amd import synthetic
    return synthetic.community_graph(device="cpu", **PROBLEM)


def initial_modules(recipe: str, seed: int):
    """oracle modules with the run's initial weights (their state_dict keys are the product's)"""
    import oracle as O
    r = RECIPES[recipe]
    n = PROBLEM["num_nodes"]
    torch.manual_seed(21 + 7919 * seed)
    enc = O.GNNRef("SAGE", H, H, H, r["layers"], 0.0)
    pred = O.MLPPredictorRef(H, H, 1, 2, 0.0) if r["predictor"] == "MLP" else O.DotPredictorRef()
    emb = torch.nn.Embedding(n, H)
    enc.reset_parameters()
    if r["predictor"] == "MLP":
        pred.reset_parameters()
    torch.nn.init.xavier_uniform_(emb.weight)
    return enc, pred, emb


def epoch_seed(epoch: int, seed: int) -> int:
    return 1000 + epoch + 100003 * seed


def run_oracle(args):
    """(recipe, seed, 'f32' | 'f64', epochs) -> {K: (valid, test)} in percent.  Top-level so that a process pool
    can run the seeds side by side (each worker: 2 threads)."""
    recipe, seed, dtype, epochs = args
    import oracle as O
    torch.set_num_threads(2)
    r = RECIPES[recipe]
    g = problem()
    n = g["num_nodes"]
    enc, pred, emb = initial_modules(recipe, seed)
    adj = g["adj_t"]
    if dtype == "f64":
        enc, pred, emb = enc.double(), pred.double(), emb.double()
    csr = O.CSR(adj.rowptr, adj.col.to(torch.int64), None, n)
    tr = O.TrainerRef(enc, pred, emb, csr, loss_name=r["loss"], lr=r["lr"], clip_norm=r["clip"])
    train = g["train"]
    for epoch in range(epochs):
        torch.manual_seed(epoch_seed(epoch, seed))
        _, neg = O.pos_neg_edges_ref("train", {"train": {"edge": train}}, num_nodes=n, neg_sampler_name="local",
                                     num_neg=r["k"])
        tr.train_epoch(train, neg, B, r["k"], None)
    hh = tr.embed_for_eval()
    res = O.evaluate_hits_ref(tr.score(hh, g["valid"], B), tr.score(hh, g["valid_neg"], B),
                              tr.score(hh, g["test"], B), tr.score(hh, g["test_neg"], B))
    return {k: (100.0 * v[0], 100.0 * v[1]) for k, v in res.items()}


def run_hip(P, recipe: str, seed: int, math: str, epochs: int, g=None):
    """the same run on the HIP path (BaseModel.train / BaseModel.test) with the dense products formed as `math`"""
    from plnlp_amd.utils import Evaluator
    r = RECIPES[recipe]
    g = problem() if g is None else g
    n = g["num_nodes"]
    old = P.ops.GEMM_MATH["mode"]
    P.ops.GEMM_MATH["mode"] = math
    try:
        m = P.BaseModel(lr=r["lr"], dropout=0.0, grad_clip_norm=r["clip"], gnn_num_layers=r["layers"], mlp_num_layers=2,
                        emb_hidden_channels=H, gnn_hidden_channels=H, mlp_hidden_channels=H, num_nodes=n,
                        num_node_feats=0, gnn_encoder_name="SAGE", predictor_name=r["predictor"],
                        loss_func=r["loss"], optimizer_name="Adam", device="cuda", use_node_feats=False,
                        train_node_emb=True)
        enc, pred, emb = initial_modules(recipe, seed)
        m.encoder.load_state_dict(enc.state_dict())
        if r["predictor"] == "MLP":
            m.predictor.load_state_dict(pred.state_dict())
        with torch.no_grad():
            m.emb.weight.copy_(emb.weight)

        class D:
            pass
        data = D()
        data.adj_t = g["adj_t"].to("cuda")
        data.edge_index = g["data"].edge_index
        split = {"train": {"edge": g["train"]},
                 "valid": {"edge": g["valid"], "edge_neg": g["valid_neg"]},
                 "test": {"edge": g["test"], "edge_neg": g["test_neg"]}}
        for epoch in range(epochs):
            torch.manual_seed(epoch_seed(epoch, seed))
            m.train(data, split, B, "local", r["k"])
        res = m.test(data, split, B, Evaluator("ogbl-ddi"), "hits")
        return {k: (100.0 * v[0], 100.0 * v[1]) for k, v in res.items()}
    finally:
        P.ops.GEMM_MATH["mode"] = old


def table(P, recipe: str, seeds, epochs: int, workers: int = 8):
    """{arithmetic: {K: array [n_seeds, 2] (valid, test)}}; the oracle runs go through a process pool while this
    process drives the GPU"""
    import concurrent.futures as cf
    import multiprocessing as mp
    jobs = [(recipe, s, dt, epochs) for dt in ("f64", "f32") for s in seeds]
    out = {}
    g = problem()
    with cf.ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn")) as pool:
        futs = [pool.submit(run_oracle, j) for j in jobs]
        for math in ("bf16x3", "f32"):
            out["hip_" + math] = [run_hip(P, recipe, s, math, epochs, g) for s in seeds]
        res = [f.result() for f in futs]
    out["oracle_f64"] = res[: len(seeds)]
    out["oracle_f32"] = res[len(seeds):]
    return {arith: {k: np.array([run[k] for run in runs]) for k in runs[0]} for arith, runs in out.items()}


def summarize(tab):
    """text table: mean +- std over seeds per arithmetic and K, and the paired difference to the float64 oracle"""
    lines = []
    ref = tab["oracle_f64"]
    for k in sorted(ref, key=lambda s: int(s.split("@")[1])):
        lines.append(f"{k}  (percent; valid / test; mean +- std over {ref[k].shape[0]} seeds)")
        for arith in ("oracle_f64", "oracle_f32", "hip_f32", "hip_bf16x3"):
            v = tab[arith][k]
            d = v - ref[k]
            lines.append(f"  {arith:11s} {v[:, 0].mean():6.2f} +- {v[:, 0].std(ddof=1):5.2f} / {v[:, 1].mean():6.2f} +- "
                         f"{v[:, 1].std(ddof=1):5.2f}    mean - f64: {d[:, 0].mean():+6.3f} / {d[:, 1].mean():+6.3f}"
                         f"    paired std {d[:, 0].std(ddof=1):5.3f} / {d[:, 1].std(ddof=1):5.3f}")
    return "\n".join(lines)
