"""Generate tests/golden/*.npz by RUNNING the reference's own Python.

Run only in the build container (needs /root/reference, which never travels):

    python tests/golden/make_golden.py

The reference imports torch_geometric / torch_sparse / torch_cluster / ogb, none
of which is installed.  A sys.modules stub provides the *names*; the slots whose
arithmetic lives in those wheels (SAGEConv, GCNConv, negative_sampling) are
filled with the oracle's restatements where a fixture needs them (G8), so the
fixtures pin the REFERENCE's own code: losses, predictors, BaseGNN control
flow, samplers, get_pos_neg_edges, BaseModel.train loop order / clipping / loss
accounting, Logger.  Fixtures are data only (inputs + expected outputs).
"""
import io
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import reference_path as O  # noqa: E402


# ---------------------------------------------------------------- stub ------
class _ConvSlot(torch.nn.Module):
    """placeholder conv; G3 injects simple convs, G8 injects oracle convs."""

    def __init__(self, cin, cout, **kw):
        super().__init__()
        self.cin, self.cout, self.kw = cin, cout, kw

    def reset_parameters(self):
        pass


class _RefSAGEConv(O.SAGEConvRef):
    def __init__(self, cin, cout, **kw):
        super().__init__(cin, cout)


class _RefGCNConv(O.GCNConvRef):
    def __init__(self, cin, cout, normalize=False, **kw):
        super().__init__(cin, cout)


_state = {"negative_sampling": None}


def _negative_sampling(edge_index, num_nodes=None, num_neg_samples=None, method="sparse"):
    return _state["negative_sampling"](edge_index, num_nodes, num_neg_samples)


def _add_self_loops(edge_index, edge_weight=None, fill_value=1.0, num_nodes=None):
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    loops = torch.arange(n).repeat(2, 1)
    return torch.cat([edge_index, loops], dim=1), None


def install_stub():
    tg = types.ModuleType("torch_geometric")
    nn = types.ModuleType("torch_geometric.nn")
    ut = types.ModuleType("torch_geometric.utils")
    for name in ("GraphConv", "TransformerConv"):
        setattr(nn, name, type(name, (_ConvSlot,), {}))
    # arithmetic of these two lives in PyG (absent): the oracle restatements
    # stand in so BaseModel.train can run end to end (G8)
    nn.SAGEConv, nn.GCNConv = _RefSAGEConv, _RefGCNConv
    ut.negative_sampling = _negative_sampling
    ut.add_self_loops = _add_self_loops
    tg.nn, tg.utils = nn, ut
    sys.modules.update({"torch_geometric": tg, "torch_geometric.nn": nn,
                        "torch_geometric.utils": ut})
    sys.path.insert(0, REF)


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, len(out), "arrays")


# ---------------------------------------------------------------- G1 --------
def g1_losses():
    from plnlp import loss as L
    fns = {
        "auc": (L.auc_loss, False), "hinge_auc": (L.hinge_auc_loss, False),
        "weighted_auc": (L.weighted_auc_loss, True), "adaptive_auc": (L.adaptive_auc_loss, True),
        "weighted_hinge_auc": (L.weighted_hinge_auc_loss, True),
        "adaptive_hinge_auc": (L.adaptive_hinge_auc_loss, True),
        "log_rank": (L.log_rank_loss, False), "info_nce": (L.info_nce_loss, False),
    }
    arrays = {}
    case = 0
    for seed in (0, 1, 2):
        for B in (1, 5, 64):
            for k in (1, 3):
                g = torch.Generator().manual_seed(1000 * seed + 10 * B + k)
                pos = torch.randn(B, 1, generator=g)
                neg = torch.randn(B * k, 1, generator=g)
                w = torch.rand(B, generator=g) + 0.25
                arrays[f"c{case}_pos"], arrays[f"c{case}_neg"], arrays[f"c{case}_w"] = pos, neg, w
                arrays[f"c{case}_k"] = k
                for name, (fn, needs_w) in fns.items():
                    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
                        p = pos.to(dt).clone().requires_grad_(True)
                        n = neg.to(dt).clone().requires_grad_(True)
                        out = fn(p, n, k, w.to(dt)) if needs_w else fn(p, n, k)
                        out.backward()
                        arrays[f"c{case}_{name}_{tag}_loss"] = out
                        arrays[f"c{case}_{name}_{tag}_gpos"] = p.grad
                        arrays[f"c{case}_{name}_{tag}_gneg"] = n.grad
                p = pos.clone().requires_grad_(True)
                n = neg.clone().requires_grad_(True)
                out = L.ce_loss(p, n)
                out.backward()
                arrays[f"c{case}_ce_f32_loss"], arrays[f"c{case}_ce_f32_gpos"] = out, p.grad
                arrays[f"c{case}_ce_f32_gneg"] = n.grad
                case += 1
    arrays["num_cases"] = case
    npz("g1_losses", **arrays)


# ---------------------------------------------------------------- G2 --------
def g2_predictors():
    from plnlp import layer as Ly
    arrays = {}
    for L in (1, 2, 3):
        torch.manual_seed(40 + L)
        m = Ly.MLPPredictor(16, 16, 1, L, 0.0)
        xi = torch.randn(9, 16, requires_grad=True)
        xj = torch.randn(9, 16, requires_grad=True)
        out = m(xi, xj)
        out.sum().backward()
        for k, v in m.state_dict().items():
            arrays[f"mlp{L}_sd_{k}"] = v
        for k, p in m.named_parameters():
            arrays[f"mlp{L}_grad_{k}"] = p.grad
        arrays[f"mlp{L}_xi"], arrays[f"mlp{L}_xj"], arrays[f"mlp{L}_out"] = xi, xj, out
        arrays[f"mlp{L}_gxi"], arrays[f"mlp{L}_gxj"] = xi.grad, xj.grad
    torch.manual_seed(7)
    d = Ly.DotPredictor()
    xi = torch.randn(9, 16, requires_grad=True)
    xj = torch.randn(9, 16, requires_grad=True)
    out = d(xi, xj)
    (out * torch.arange(1.0, 10.0)).sum().backward()
    arrays.update(dot_xi=xi, dot_xj=xj, dot_out=out, dot_gxi=xi.grad, dot_gxj=xj.grad)
    # factories (model.py:252-276): class chosen per name
    from plnlp import model as M
    names = {}
    for n in ("DOT", "dot", "BIL", "MLP", "mlp", "MLPDOT", "MLPBIL", "MLPCAT", "nope"):
        p = M.create_predictor_layer(8, 2, 0.0, n)
        names[n] = "None" if p is None else type(p).__name__
    arrays["pred_factory_keys"] = np.array(list(names.keys()))
    arrays["pred_factory_vals"] = np.array(list(names.values()))
    enc = {}
    for n in ("SAGE", "sage", "GCN", "gcn", "WSAGE", "Transformer", "whatever"):
        enc[n] = type(M.create_gnn_layer(4, 8, 2, 0.0, n)).__name__
    arrays["enc_factory_keys"] = np.array(list(enc.keys()))
    arrays["enc_factory_vals"] = np.array(list(enc.values()))
    npz("g2_predictors", **arrays)


# ---------------------------------------------------------------- G3 --------
class _AffineConv(torch.nn.Module):
    """x -> x * a - b (sign-mixing so ReLU matters); ignores adj."""

    def __init__(self, a, b):
        super().__init__()
        self.a, self.b = a, b

    def reset_parameters(self):
        pass

    def forward(self, x, adj_t):
        return x * self.a - self.b


def g3_control_flow():
    from plnlp import layer as Ly
    arrays = {}
    torch.manual_seed(3)
    x = torch.randn(6, 5)
    arrays["x"] = x
    for L in (1, 2, 3):
        g = Ly.BaseGNN(0.0, L)
        for i in range(L):
            g.convs.append(_AffineConv(1.5 - i, 0.3 * (i + 1)))
        g.eval()
        arrays[f"out_L{L}"] = g(x, None)
    npz("g3_control_flow", **arrays)


# ---------------------------------------------------------------- G4/G5 -----
def g4_g5_samplers():
    from plnlp import negative_sample as NS
    arrays = {}
    torch.manual_seed(0)
    pos = torch.tensor([[0, 1], [2, 3]])
    arrays["local_s0"] = NS.local_neg_sample(pos, 10, 3)
    for seed, (E, N, k) in enumerate([(7, 13, 1), (33, 100, 3), (1, 5, 4)], start=11):
        g = torch.Generator().manual_seed(seed)
        pe = torch.randint(0, N, (E, 2), generator=g)
        torch.manual_seed(seed)
        arrays[f"local_{seed}_pos"] = pe
        arrays[f"local_{seed}_N"], arrays[f"local_{seed}_k"] = N, k
        arrays[f"local_{seed}_out"] = NS.local_neg_sample(pe, N, k)
    # G5a sample_perm_copy
    ei = torch.tensor([[0, 1, 2, 3, 4], [5, 6, 7, 8, 9]])
    torch.manual_seed(21)
    arrays["permcopy_in"] = ei
    arrays["permcopy_out_t8_c3"] = NS.sample_perm_copy(ei, 8, 3)
    torch.manual_seed(22)
    arrays["permcopy_out_t5_c2"] = NS.sample_perm_copy(ei, 5, 2)
    # G5b padding branch of global_neg_sample with a fixed short structured sampler
    short = torch.tensor([[9, 8, 7, 6], [1, 2, 3, 4]])
    _state["negative_sampling"] = lambda ei_, n_, m_: short
    edge_index = torch.tensor([[0, 1, 2], [1, 2, 0]])
    torch.manual_seed(23)
    arrays["globalpad_short"] = short
    arrays["globalpad_out"] = NS.global_neg_sample(edge_index, 10, 3, 2)   # wants 6, gets 4
    seen = {}
    _state["negative_sampling"] = lambda ei_, n_, m_: (seen.update(ei=ei_.clone(), n=n_, m=m_) or short)
    NS.global_neg_sample(edge_index, 10, 2, 2)
    arrays["globalcall_edge_index"], arrays["globalcall_n"], arrays["globalcall_m"] = seen["ei"], seen["n"], seen["m"]
    npz("g4_g5_samplers", **arrays)


# ---------------------------------------------------------------- G6 --------
def g6_dataloader():
    from torch.utils.data import DataLoader
    arrays = {}
    for seed, n, B in [(123, 10, 4), (5, 1000, 64), (77, 65, 65), (8, 3, 10)]:
        torch.manual_seed(seed)
        batches = [torch.as_tensor(b) for b in DataLoader(range(n), B, shuffle=True)]
        after = torch.randint(0, 1 << 30, (4,))
        arrays[f"s{seed}_n{n}_B{B}_perm"] = torch.cat(batches)
        arrays[f"s{seed}_n{n}_B{B}_sizes"] = np.array([b.numel() for b in batches])
        arrays[f"s{seed}_n{n}_B{B}_after"] = after
    torch.manual_seed(9)
    for _ in DataLoader(range(10), 4):
        pass
    arrays["noshuffle_after"] = torch.randint(0, 1 << 30, (4,))
    npz("g6_dataloader", **arrays)


# ---------------------------------------------------------------- G7 --------
def g7_pos_neg_edges():
    from plnlp import utils as U
    arrays = {}
    g = torch.Generator().manual_seed(70)
    S, K, N = 6, 5, 40
    se = {s: {"source_node": torch.randint(0, N, (S,), generator=g),
              "target_node": torch.randint(0, N, (S,), generator=g),
              "target_node_neg": torch.randint(0, N, (S, K), generator=g)} for s in ("train", "valid", "test")}
    for s in ("valid", "test"):
        pos, neg = U.get_pos_neg_edges(s, se)
        arrays[f"cit_{s}_pos"], arrays[f"cit_{s}_neg"] = pos, neg
    torch.manual_seed(71)
    pos, neg = U.get_pos_neg_edges("train", se, num_nodes=N, neg_sampler_name="local", num_neg=3)
    arrays["cit_train_pos"], arrays["cit_train_neg"] = pos, neg
    for s in ("train", "valid", "test"):
        for k, v in se[s].items():
            arrays[f"cit_in_{s}_{k}"] = v
    se2 = {"train": {"edge": torch.randint(0, N, (9, 2), generator=g)},
           "valid": {"edge": torch.randint(0, N, (4, 2), generator=g),
                     "edge_neg": torch.randint(0, N, (7, 2), generator=g)}}
    pos, neg = U.get_pos_neg_edges("valid", se2)
    arrays["edge_valid_pos"], arrays["edge_valid_neg"] = pos, neg
    arrays["edge_in_valid_edge"], arrays["edge_in_valid_edge_neg"] = se2["valid"]["edge"], se2["valid"]["edge_neg"]
    npz("g7_pos_neg_edges", **arrays)


# ---------------------------------------------------------------- G8 --------
def _toy_graph(n, e, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.randint(0, n, (e,), generator=g)
    b = torch.randint(0, n, (e,), generator=g)
    keep = a != b
    a, b = a[keep], b[keep]
    lo, hi = torch.minimum(a, b), torch.maximum(a, b)
    key = torch.unique(lo * n + hi)
    lo, hi = key // n, key % n
    w = (torch.rand(lo.numel(), generator=g) * 4 + 1).floor()
    return lo, hi, w


class _Data:
    pass


def g8_train_trajectory():
    """Full reference BaseModel.train (model.py:128-173) on a 200-node toy graph
    with the oracle's conv restatements in the PyG slots.  Pins loop order,
    per-group clipping, Adam, loss accounting, index streams."""
    from plnlp import model as M
    arrays = {}
    N = 200
    lo, hi, w = _toy_graph(N, 900, 80)
    row = torch.cat([lo, hi])
    col = torch.cat([hi, lo])
    val = torch.cat([w, w])
    adj = O.CSR.from_coo(row, col, val, N)
    arrays.update(N=N, lo=lo, hi=hi, w=w)
    configs = {
        # name: encoder, predictor, loss, L_gnn, L_mlp, h, num_neg, clip, weighted, B
        "sage_mlp_auc": ("SAGE", "MLP", "AUC", 2, 2, 16, 3, 2.0, False, 256),
        "sage1_dot_whinge": ("SAGE", "DOT", "WeightedHingeAUC", 1, 2, 16, 1, 1.0, True, 300),
        "gcn_mlp_auc": ("GCN", "MLP", "AUC", 2, 2, 12, 3, 1.0, False, 512),
        "sage_dot_hinge": ("SAGE", "DOT", "HingeAUC", 2, 2, 8, 2, -1.0, False, 200),
        "sage_mlp_whinge_noweight": ("SAGE", "MLP", "WeightedHingeAUC", 2, 3, 8, 2, 2.0, False, 400),
    }
    for name, (enc, pred, lossn, Lg, Lm, h, k, clip, weighted, B) in configs.items():
        torch.manual_seed(800 + len(name))
        m = M.BaseModel(lr=0.01, dropout=0.0, grad_clip_norm=clip, gnn_num_layers=Lg,
                        mlp_num_layers=Lm, emb_hidden_channels=h, gnn_hidden_channels=h,
                        mlp_hidden_channels=h, num_nodes=N, num_node_feats=0,
                        gnn_encoder_name=enc, predictor_name=pred, loss_func=lossn,
                        optimizer_name="Adam", device=torch.device("cpu"),
                        use_node_feats=False, train_node_emb=True)
        m.param_init()
        data = _Data()
        data.adj_t = O.gcn_norm_csr(adj) if enc == "GCN" else adj
        data.edge_index = torch.stack([col, row])
        split = {"train": {"edge": torch.stack([lo, hi], 1)}}
        if weighted:
            split["train"]["weight"] = (w / w.max()).to(torch.float32)
        init_sd = {}
        for k_, v in m.encoder.state_dict().items():
            init_sd["enc." + k_] = v.clone()
        for k_, v in m.predictor.state_dict().items():
            init_sd["pred." + k_] = v.clone()
        init_sd["emb.weight"] = m.emb.weight.detach().clone()
        torch.manual_seed(4242)
        losses = [m.train(data, split, B, "local", k) for _ in range(3)]
        for k_, v in init_sd.items():
            arrays[f"{name}_init_{k_}"] = v
        arrays[f"{name}_losses"] = np.array(losses, dtype=np.float64)
        arrays[f"{name}_final_emb"] = m.emb.weight.detach()
        for k_, v in m.encoder.state_dict().items():
            arrays[f"{name}_final_enc.{k_}"] = v
        for k_, v in m.predictor.state_dict().items():
            arrays[f"{name}_final_pred.{k_}"] = v
        arrays[f"{name}_cfg"] = np.array([enc, pred, lossn, str(Lg), str(Lm), str(h), str(k),
                                          str(clip), str(int(weighted)), str(B)])
    arrays["config_names"] = np.array(list(configs.keys()))
    npz("g8_train_trajectory", **arrays)


# ---------------------------------------------------------------- G9 --------
def g9_logger_and_lr():
    from plnlp.logger import Logger
    from plnlp.model import adjust_lr
    lg = Logger(3)
    g = torch.Generator().manual_seed(90)
    res = torch.rand(3, 5, 2, generator=g)
    res[1, 1, 0] = res[1, 3, 0] = 0.99      # tie -> last_best matters
    for r in range(3):
        for e in range(5):
            lg.add_result(r, (float(res[r, e, 0]), float(res[r, e, 1])))
    texts = {}
    for tag, kw in {"run1": dict(run=1), "run1_last": dict(run=1, last_best=True),
                    "all": dict(), "all_last": dict(last_best=True)}.items():
        buf = io.StringIO()
        lg.print_statistics(f=buf, **kw)
        texts[tag] = buf.getvalue()
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=0.01)
    lrs = [adjust_lr(opt, r, 0.01) for r in (0.0, 0.25, 0.5, 0.99995, 1.0)]
    npz("g9_logger", results=res, keys=np.array(list(texts.keys())),
        texts=np.array(list(texts.values())), lrs=np.array(lrs, dtype=np.float64))


if __name__ == "__main__":
    assert os.path.isdir(REF), "reference not mounted: fixtures can only be made in the build container"
    install_stub()
    g1_losses()
    g2_predictors()
    g3_control_flow()
    g4_g5_samplers()
    g6_dataloader()
    g7_pos_neg_edges()
    g9_logger_and_lr()
    g8_train_trajectory()
