"""Hits@K parity in a TRAINED regime that can fail (BASELINE.json: "Hits@K within +-0.3 of reference").

Two legs, each one recipe of the reference with its OWN loss and schedule, each on the problem that makes its question
answerable:

  collab  README.md:35 -- SAGE x1 + DOT, WeightedHingeAUC on random-walk pairs with 1 / hop weights (main.py:241-253),
          one negative per pair, grad clip 1, linear lr decay (model.py:279-286), the recipe's own batch of 65 536;
          metric Hits@50 (README.md:9).  Problem: plnlp_amd.synthetic.geometric_graph, a soft random geometric graph
          on which NOTHING SATURATES -- a converged model ranks held-out edges against uniform non-edges at
          Hits@20 / 50 / 100 ~ 55 / 81 / 93 %, set by how well the embedding recovered the geometry.  Question: does the
          HIP arithmetic land on the same LEVEL (+-0.3) and get there at the same pace?
  ddi     README.md:24 -- SAGE x2 + MLP, AUC loss, three negatives, grad clip 2; metric Hits@20 (README.md:8).
          Problem: the stochastic block model of round 3 (community_graph, 10 % unrankable positives): every CONVERGED
          run sits on a plateau of 90.0 % -- and at Hits@20 one run in ten is still short of it after 60 epochs, in
          every arithmetic.  Round 3's ten seeds could not tell a lottery from a slower time-to-plateau on this path
          (the one with the MLP GEMMs).  Question here, over 48 seeds: is the distribution of EPOCHS-TO-PLATEAU that of
          the reference's arithmetic, and is the plateau the same?  (On the geometric problem this recipe's final
          Hits@20 scatters by +-7 points between the ORACLE's own float32 and float64 runs of one seed -- no +-0.3
          statement exists there for any arithmetic; recorded in profiles/r04_trained_parity.md.)
Four arithmetics train the SAME runs -- same initial weights, walks, negatives, batch permutations:
  HIP split-bf16 GEMMs (the product's default), HIP f32-MFMA GEMMs,
  the CPU oracle in float32 (the reference's arithmetic) and in float64.
Hits@20/50/100 on valid and test are recorded after EVERY epoch; compared over the seeds:
  * the final level  = the recipe's own Hits@K averaged over the last FINAL_EPOCHS epochs; mean over seeds (collab),
    median over seeds (ddi: robust against the one-in-ten straggler)                           (asserted at +-0.3)
  * epochs-to-level  = first epoch whose valid Hits@K reaches the level (collab: 90 % of the float64 oracle's mean final
    level; ddi: 88 %, two points under the plateau), epochs + 1 if never -- a distribution over seeds, HIP vs oracle
    float32 by a two-sided Mann-Whitney U test
and the same harness is run on a deliberately DEGRADED product, which must fail.
Test infrastructure: imports the oracle; used by tests/test_hip_round4.py and tests/golden/make_trained_curves.py only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PROBLEMS = {
    "geometric": dict(num_nodes=3000, avg_degree=30.0, softness=0.15, holdout=0.15, n_neg=10000, seed=3),
    "sbm": dict(num_nodes=2000, community=50, p_in=0.97, cross_per_node=1.0, seed=3, unlearnable=0.1),
    # the WIDE legs (round 5): the recipes at their own widths, on graphs big enough that every multi-step run goes through
    # the kernels the benchmark measures -- the stationary-weights split-bf16 GEMM (>= 16 384 rows of A), the F = 256 / 512
    # aggregation forms with hub rows beyond the long-row threshold (fused chunk pass), the touched-rows forward.
    "geometric_wide": dict(num_nodes=40000, avg_degree=20.0, softness=0.15, holdout=0.15, n_neg=10000, seed=3, n_hubs=24,
                           hub_degree=1500.0),
    # ddi's own size and density: 4 267 nodes, ~450 neighbours each (README.md:24's graph has 2.1 M entries): the block
    # model of the small ddi leg with eight communities of 500 (p_in 0.99: a dozen same-block non-edges among the 10 000
    # negatives, fewer than K = 20, so a converged run sits on the 90 % plateau the 10 % unrankable positives leave)
    "sbm_dense": dict(num_nodes=4267, community=500, p_in=0.99, cross_per_node=1.0, seed=3, unlearnable=0.1),
}
H = 64
KS = ("Hits@20", "Hits@50", "Hits@100")
RECIPES = {
    "collab": dict(problem="geometric", layers=1, predictor="DOT", loss="WeightedHingeAUC", k=1, lr=0.03, clip=1.0,
                   epochs=40, batch=65536, walk_length=3, decay=True, metric="Hits@50", level=None, center="mean",
                   seeds=64),
    "ddi": dict(problem="sbm", layers=2, predictor="MLP", loss="AUC", k=3, lr=0.005, clip=2.0, epochs=60, batch=2048,
                walk_length=0, decay=False, metric="Hits@20", level=88.0, center="median", seeds=48),
    # README.md:35 at h = 256: batches of 32 768 pairs (+ as many negatives) touch ~96 % of the 40 000 nodes -- inside the
    # row-sparse window (ops.SPARSE_BACKWARD), so the last conv runs at the ~38 000 touched rows, as on the real graph
    "collab_wide": dict(problem="geometric_wide", h=256, layers=1, predictor="DOT", loss="WeightedHingeAUC", k=1, lr=0.01,
                        clip=1.0, epochs=8, batch=32768, walk_length=1, decay=True, metric="Hits@50", level=None,
                        center="mean", seeds=16),
    # README.md:24 at h = 512: 8 192 positives + 3 x 8 192 negatives = 32 768 scorer rows per step; an epoch is the first
    # `train_edges` training edges = 12 steps (the encoder still aggregates over the whole graph every step).  30 epochs = 360
    # steps: a GPU pilot (scripts/pilot_wide_legs.py) put the run on its 90 % Hits@20 plateau after 11-19 epochs at this
    # learning rate and left it there (at 0.005 it arrives earlier and then decays slowly, at 0.01 it oscillates)
    "ddi_wide": dict(problem="sbm_dense", h=512, layers=2, predictor="MLP", loss="AUC", k=3, lr=0.002, clip=2.0,
                     epochs=30, batch=8192, walk_length=0, decay=False, metric="Hits@20", level=88.0, center="median",
                     seeds=8, train_edges=98304),
}


def width(recipe: str) -> int:
    return RECIPES[recipe].get("h", H)


WIDE = ("collab_wide", "ddi_wide")
KS_WIDE = KS + ("AUC",)


def metrics_of(recipe: str):
    """the wide legs record, beside Hits@20/50/100, the AUC of the positives against the negatives (the share of
    (positive, negative) pairs ranked the right way round, in percent): a metric that moves from the first step on -- the
    ddi recipe at h = 512 is still near chance at Hits@20 after the few dozen steps the CPU oracle can afford"""
    return KS_WIDE if recipe in WIDE else KS


def auc_percent(pos: torch.Tensor, neg: torch.Tensor) -> float:
    """100 * P(score(pos) > score(neg)) + 50 * P(equal), exact (sort + two binary searches), float64 counts"""
    neg_sorted, _ = torch.sort(neg.reshape(-1).double())
    p = pos.reshape(-1).double()
    below = torch.searchsorted(neg_sorted, p, right=False)
    upto = torch.searchsorted(neg_sorted, p, right=True)
    return float(100.0 * (below.double() + 0.5 * (upto - below).double()).mean() / neg_sorted.numel())
FINAL_EPOCHS = 3
LEVEL_FRACTION = 0.9
_problem = {}


GENERATOR_VERSION = 1          # bump when synthetic.geometric_graph_blocked changes what it returns for the same arguments


def _cache_path(name: str) -> str:
    """where the geometric_wide problem is kept between the session's processes (~1 minute of all-pairs blocks): keyed on the
    problem's arguments and the generator's version, in the repository's scratch directory or a directory only this user can
    write (a predictable path in a shared /tmp was unpickled)"""
    import hashlib
    key = hashlib.sha1(repr((sorted(PROBLEMS[name].items()), GENERATOR_VERSION)).encode()).hexdigest()[:16]
    base = os.path.join(ROOT, "gpurun_out")
    if not os.path.isdir(base):
        base = os.path.join(os.environ.get("TMPDIR", "/tmp"), "plnlp_parity_%d" % os.getuid())
        os.makedirs(base, mode=0o700, exist_ok=True)
        st = os.stat(base)
        if st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise RuntimeError(f"{base} is not a private directory of this user")
    return os.path.join(base, f"{name}_{key}.pt")


def _pack(g) -> dict:
    """the problem as plain tensors (loaded back with weights_only=True)"""
    out = {k: g[k] for k in ("train", "valid", "test", "valid_neg", "test_neg", "pos")}
    out["num_nodes"] = torch.tensor(g["num_nodes"])
    out["rowptr"], out["col"] = g["adj_t"].rowptr, g["adj_t"].col
    return out


def _unpack(d) -> dict:
    from plnlp_amd import Graph
    from plnlp_amd.synthetic import SyntheticData
    n = int(d["num_nodes"])
    adj = Graph(d["rowptr"], d["col"], None, n, n)
    rr, cc, _ = adj.coo()
    g = {k: d[k] for k in ("train", "valid", "test", "valid_neg", "test_neg", "pos")}
    g.update(num_nodes=n, adj_t=adj, data=SyntheticData(adj_t=adj, edge_index=torch.stack([cc, rr]).cpu(), num_nodes=n))
    return g


def problem(recipe: str):
    name = RECIPES[recipe]["problem"]
    if name not in _problem:
        from plnlp_amd import synthetic
        if name == "geometric_wide":
            cache = _cache_path(name)
            if os.path.exists(cache):
                g = _unpack(torch.load(cache, weights_only=True))
            else:
                g = synthetic.geometric_graph_blocked(**PROBLEMS[name])
                torch.save(_pack(g), cache + ".%d" % os.getpid())
                os.replace(cache + ".%d" % os.getpid(), cache)
            _problem[name] = g
        elif name.startswith("geometric"):
            _problem[name] = synthetic.geometric_graph(**PROBLEMS[name])
        else:
            _problem[name] = synthetic.community_graph(device="cpu", **PROBLEMS[name])
        te = RECIPES[recipe].get("train_edges")
        if te:                                # the epoch's training edges: a fixed prefix (the graph keeps every train edge)
            _problem[name] = dict(_problem[name], train=_problem[name]["train"][:te].clone())
    return _problem[name]


def initial_modules(recipe: str, seed: int):
    """oracle modules holding the run's initial weights (their state_dict keys are the product's)"""
    import oracle as O
    r = RECIPES[recipe]
    n = PROBLEMS[r["problem"]]["num_nodes"]
    torch.manual_seed(21 + 7919 * seed)
    h = width(recipe)
    # (the wide legs aggregate through torch's CSR product -- the formulation bench.py's cpu_baseline times: the
    # transparent gather + index_add form materialises [entries, h], 7 GB per pass in float64 on the dense graph)
    enc = O.GNNRef("SAGE", h, h, h, r["layers"], 0.0, spmm_impl="sparse_csr" if recipe in WIDE else "index_add")
    pred = O.MLPPredictorRef(h, h, 1, 2, 0.0) if r["predictor"] == "MLP" else O.DotPredictorRef()
    emb = torch.nn.Embedding(n, h)
    enc.reset_parameters()
    if r["predictor"] == "MLP":
        pred.reset_parameters()
    torch.nn.init.xavier_uniform_(emb.weight)
    return enc, pred, emb


def epoch_seed(epoch: int, seed: int) -> int:
    return 1000 + epoch + 100003 * seed


def walk_seed(epoch: int, seed: int) -> int:
    return 77 + epoch + 1000 * seed


def _eval_sets(g, r):
    return g["valid"], g["valid_neg"], g["test"], g["test_neg"]


def run_oracle(args):
    """(recipe, seed, 'f32' | 'f64') -> (hits [epochs, 3, 2] in percent, losses [epochs]).  Top-level so that a
    process pool can run the seeds side by side (one thread each)."""
    recipe, seed, dtype = args
    import oracle as O
    torch.set_num_threads(int(os.environ.get("PLNLP_ORACLE_THREADS", "1")))     # (results are thread-count independent only at 1)
    r = RECIPES[recipe]
    g = problem(recipe)
    n = g["num_nodes"]
    enc, pred, emb = initial_modules(recipe, seed)
    adj = g["adj_t"]
    if dtype == "f64":
        enc, pred, emb = enc.double(), pred.double(), emb.double()
    csr = O.CSR(adj.rowptr, adj.col.to(torch.int64), None, n)
    tr = O.TrainerRef(enc, pred, emb, csr, loss_name=r["loss"], lr=r["lr"], clip_norm=r["clip"])
    train = g["train"]
    pv, nv, pt, nt = _eval_sets(g, r)
    hits, losses = [], []
    for epoch in range(r["epochs"]):
        torch.manual_seed(epoch_seed(epoch, seed))
        pos, w = train, None
        if r["walk_length"]:                               # main.py:241-253
            walk = O.random_walk_ref(csr, train.reshape(-1), r["walk_length"], walk_seed(epoch, seed))
            pos, w = O.random_walk_pairs_ref(walk, r["walk_length"])
        _, neg = O.pos_neg_edges_ref("train", {"train": {"edge": pos}}, num_nodes=n, neg_sampler_name="local",
                                     num_neg=r["k"])
        losses.append(tr.train_epoch(pos, neg, r["batch"], r["k"], w))
        if r["decay"]:
            O.adjust_lr_ref(tr.optimizer, (epoch + 1) / r["epochs"], r["lr"])      # main.py:288-291
        hh = tr.embed_for_eval()
        B = r["batch"]
        spv, snv, spt, snt = tr.score(hh, pv, B), tr.score(hh, nv, B), tr.score(hh, pt, B), tr.score(hh, nt, B)
        res = O.evaluate_hits_ref(spv, snv, spt, snt)
        row = [[100.0 * res[k][0], 100.0 * res[k][1]] for k in KS]
        if recipe in WIDE:
            row.append([auc_percent(spv, snv), auc_percent(spt, snt)])
        hits.append(row)
    return np.array(hits), np.array(losses)


class Mutation:
    """a deliberately degraded product for the harness's power check -- patches plnlp_amd.ops in place (with-block)"""

    def __init__(self, P, kind: str):
        self.P, self.kind = P, kind

    def __enter__(self):
        ops = self.P.ops
        self.saved = {}
        if self.kind == "bf16_operands":          # every dense operand rounded to ONE bf16 term (single-term bf16 GEMM)
            self.saved["_f32c"] = ops._f32c
            orig = ops._f32c
            ops._f32c = lambda t: orig(t).to(torch.bfloat16).to(torch.float32)
        elif self.kind.startswith("agg_noise:"):  # relative noise on every aggregated row (a sloppy reduction)
            eps = float(self.kind.split(":")[1])
            self.saved["csr_aggregate"] = ops.csr_aggregate
            orig_agg = ops.csr_aggregate
            gen = torch.Generator(device="cuda").manual_seed(99)

            def noisy(*a, **k):
                y = orig_agg(*a, **k)
                y.mul_(1.0 + eps * torch.randn(y.shape, generator=gen, device=y.device, dtype=y.dtype))
                return y
            ops.csr_aggregate = noisy
        elif self.kind != "none":
            raise ValueError(self.kind)
        return self

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            setattr(self.P.ops, k, v)
        return False


def hip_model(P, recipe: str, seed: int, dropout: float = 0.0):
    """the HIP model of a run with the run's initial weights, and what its epochs need: (model, data, split)"""
    r = RECIPES[recipe]
    g = problem(recipe)
    n = g["num_nodes"]
    m = P.BaseModel(lr=r["lr"], dropout=dropout, grad_clip_norm=r["clip"], gnn_num_layers=r["layers"],
                    mlp_num_layers=2, emb_hidden_channels=width(recipe), gnn_hidden_channels=width(recipe),
                    mlp_hidden_channels=width(recipe),
                    num_nodes=n, num_node_feats=0, gnn_encoder_name="SAGE", predictor_name=r["predictor"],
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
    if "adj_cuda" not in g:
        g["adj_cuda"] = g["adj_t"].to("cuda")
        g["start_cuda"] = g["train"].reshape(-1).to("cuda")
    data.adj_t = g["adj_cuda"]
    data.edge_index = g["data"].edge_index
    pv, nv, pt, nt = _eval_sets(g, r)
    split = {"train": {"edge": g["train"]}, "valid": {"edge": pv, "edge_neg": nv}, "test": {"edge": pt, "edge_neg": nt}}
    return m, data, split


def hip_epoch(P, m, data, split, recipe: str, seed: int, epoch: int) -> float:
    """one training epoch of the run through BaseModel.train (fresh walks where the recipe has them, the lr decay after it)"""
    r = RECIPES[recipe]
    g = problem(recipe)
    torch.manual_seed(epoch_seed(epoch, seed))
    if r["walk_length"]:
        pairs, w = P.ops.random_walk_pairs(data.adj_t, g["start_cuda"], r["walk_length"], walk_seed(epoch, seed))
        split["train"] = {"edge": pairs.cpu(), "weight": w.cpu()}
    loss = float(m.train(data, split, r["batch"], "local", r["k"]))
    if r["decay"]:
        P.adjust_lr(m.optimizer, (epoch + 1) / r["epochs"], r["lr"])
    return loss


def run_hip(P, recipe: str, seed: int, math: str, mutation: str = "none", epochs: int = None):
    """the same run on the HIP path (BaseModel.train / BaseModel.test), dense products formed as `math`; `epochs`: stop early"""
    from plnlp_amd.utils import Evaluator
    r = RECIPES[recipe]
    old = P.ops.GEMM_MATH["mode"]
    P.ops.GEMM_MATH["mode"] = math
    try:
        with Mutation(P, mutation):
            m, data, split = hip_model(P, recipe, seed)
            pv, nv, pt, nt = (split["valid"]["edge"], split["valid"]["edge_neg"], split["test"]["edge"], split["test"]["edge_neg"])
            ev = Evaluator("ogbl-ddi")
            hits, losses = [], []
            for epoch in range(r["epochs"] if epochs is None else epochs):
                losses.append(hip_epoch(P, m, data, split, recipe, seed, epoch))
                res = m.test(data, split, r["batch"], ev, "hits")
                row = [[100.0 * res[k][0], 100.0 * res[k][1]] for k in KS]
                if recipe in WIDE:          # the scores themselves once more (BaseModel.test returns only Hits@K)
                    with torch.no_grad():
                        hh = m.encoder(m.create_input_feat(data), data.adj_t)
                        hh = torch.cat([hh, hh.mean(0, keepdim=True)], 0)
                        sc = [m.batch_predict(hh, e.to("cuda"), r["batch"], to_cpu=False) for e in (pv, nv, pt, nt)]
                    row.append([auc_percent(sc[0], sc[1]), auc_percent(sc[2], sc[3])])
                hits.append(row)
            return np.array(hits), np.array(losses)
    finally:
        P.ops.GEMM_MATH["mode"] = old


# ------------------------------------------------------------------ statistics --
def first_epoch_check(loss_hip1: np.ndarray, loss32_1: np.ndarray, loss64_1: np.ndarray):
    """epoch-1 loss, paired per seed with the float32 oracle's: the MEAN relative deviation over the seeds must not exceed the
    mean deviation of the oracle's own float64 run (+ 1e-4).  The one trained-regime statistic of the wide legs that sees
    single-term bf16 products (scripts/calibrate_wide_parity.py: ddi_wide clean 1.6e-3 / mutated 6.7e-3 / oracle gap 3.5e-3;
    collab_wide 6e-5 / 1.8e-3 / 7e-5) -- later epochs are the lottery's.  -> (passes, text)"""
    rel = np.abs(loss_hip1 - loss32_1) / loss32_1
    gap = np.abs(loss64_1 - loss32_1) / loss32_1
    return bool(rel.mean() <= gap.mean() + 1e-4), (f"epoch-1 loss vs oracle f32 over {rel.size} seeds: mean {rel.mean():.2e} max {rel.max():.2e}; "
                                                    f"oracle f64 vs f32: mean {gap.mean():.2e} max {gap.max():.2e}")



def final_level(hits: np.ndarray, recipe: str) -> np.ndarray:
    """[..., epochs, 3, 2] -> [..., 2]: mean of the recipe's own Hits@K over the last FINAL_EPOCHS epochs (valid, test)"""
    ki = metrics_of(recipe).index(RECIPES[recipe]["metric"])
    return hits[..., -FINAL_EPOCHS:, ki, :].mean(-2)


def epochs_to_level(hits: np.ndarray, recipe: str, level: float) -> np.ndarray:
    """[seeds, epochs, 3, 2] -> [seeds]: first epoch (1-based) whose VALID Hits@K reaches `level`; epochs + 1 if never"""
    ki = metrics_of(recipe).index(RECIPES[recipe]["metric"])
    reached = hits[:, :, ki, 0] >= level
    first = reached.argmax(1) + 1
    return np.where(reached.any(1), first, hits.shape[1] + 1)


def mann_whitney_p(x: np.ndarray, y: np.ndarray) -> float:
    from scipy.stats import mannwhitneyu
    if np.all(x == x[0]) and np.all(y == x[0]):
        return 1.0
    return float(mannwhitneyu(x, y, alternative="two-sided").pvalue)


def compare(hip: np.ndarray, ref32: np.ndarray, ref64: np.ndarray, recipe: str) -> dict:
    """all three [seeds, epochs, 3, 2] over the same seeds.  The verdict the test asserts on."""
    fh, f32, f64 = final_level(hip, recipe), final_level(ref32, recipe), final_level(ref64, recipe)
    r = RECIPES[recipe]
    level = r["level"] if r["level"] is not None else LEVEL_FRACTION * float(f64[:, 0].mean())
    th, t32 = epochs_to_level(hip, recipe, level), epochs_to_level(ref32, recipe, level)
    n = fh.shape[0]
    center = (lambda v: np.median(v, axis=0)) if r["center"] == "median" else (lambda v: v.mean(0))

    def se(x, y):          # of the difference of centres: paired s.e. for means, a bootstrap for medians
        if r["center"] == "mean":
            return (x - y).std(0, ddof=1) / np.sqrt(n)
        rng = np.random.default_rng(0)
        picks = rng.integers(0, n, (400, n))
        return np.array([np.median(x[i], 0) - np.median(y[i], 0) for i in picks]).std(0)
    return dict(n=n, level=level, center=r["center"],
                final_hip=center(fh), final_f32=center(f32), final_f64=center(f64),
                diff_f32=center(fh) - center(f32), diff_f32_se=se(fh, f32),
                diff_f64=center(fh) - center(f64), diff_f64_se=se(fh, f64),
                oracle_gap=center(f32) - center(f64), oracle_gap_se=se(f32, f64),
                reached_hip=float((th <= hip.shape[1]).mean()), reached_f32=float((t32 <= hip.shape[1]).mean()),
                epochs_hip=th, epochs_f32=t32, mw_p=mann_whitney_p(th, t32))


def describe(name: str, c: dict) -> str:
    f = lambda v: f"{v[0]:6.2f} / {v[1]:6.2f}"
    return (f"{name}: n = {c['n']} seeds; final level, {c['center']} over seeds (valid / test)  HIP {f(c['final_hip'])}   oracle f32 {f(c['final_f32'])}"
            f"   oracle f64 {f(c['final_f64'])}\n"
            f"    HIP - f32 {c['diff_f32'][0]:+.3f} / {c['diff_f32'][1]:+.3f}  (s.e. {c['diff_f32_se'][0]:.3f} / {c['diff_f32_se'][1]:.3f})"
            f"    HIP - f64 {c['diff_f64'][0]:+.3f} / {c['diff_f64'][1]:+.3f}"
            f"    f32 - f64 {c['oracle_gap'][0]:+.3f} / {c['oracle_gap'][1]:+.3f}  (s.e. {c['oracle_gap_se'][0]:.3f} / {c['oracle_gap_se'][1]:.3f})\n"
            f"    epochs to {c['level']:.1f} % valid: HIP median {np.median(c['epochs_hip']):.0f} mean {c['epochs_hip'].mean():.2f},"
            f" oracle f32 median {np.median(c['epochs_f32']):.0f} mean {c['epochs_f32'].mean():.2f};"
            f" reached by {100 * c['reached_hip']:.0f} % / {100 * c['reached_f32']:.0f} % of the seeds;"
            f" Mann-Whitney p = {c['mw_p']:.3f}")
