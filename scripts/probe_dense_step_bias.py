"""why does the full-size ddi step's scorer-bias gradient depend on the dense aggregation's slice count?  The test's model and batch,
forward + backward per forced slice count: the scorer's hidden pre-activation sign pattern and lins.0.bias's gradient, compared"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import plnlp_amd as P
from plnlp_amd import _lib
import test_hip_round4 as T4
c = T4._c2_problem()
n, h, k = c["n"], c["h"], c["k"]
lib = _lib.load()
P.ops.GEMM_MATH["mode"] = "bf16x3"


class Data:
    pass


res = {}
for s in (4, 3, 5, 7):
    lib.plnlp_dense_aggregate_tuning(s)
    m = P.BaseModel(lr=1e-3, dropout=0.0, grad_clip_norm=2.0, gnn_num_layers=2, mlp_num_layers=2, emb_hidden_channels=h,
                    gnn_hidden_channels=h, mlp_hidden_channels=h, num_nodes=n, num_node_feats=0, gnn_encoder_name="SAGE",
                    predictor_name="MLP", loss_func="AUC", optimizer_name="Adam", device="cuda", use_node_feats=False, train_node_emb=True)
    m.encoder.load_state_dict(c["state"][0]); m.predictor.load_state_dict(c["state"][1])
    with torch.no_grad():
        m.emb.weight.copy_(c["state"][2])
    data = Data(); data.adj_t = c["g"]["adj_t"].to("cuda")
    m.encoder.train(); m.predictor.train()
    pos, ne = c["pos"], c["neg"].reshape(-1, 2)
    hh = m.encoder(m.create_input_feat(data), data.adj_t)
    src = torch.cat([pos[:, 0], ne[:, 0]]).cuda(); dst = torch.cat([pos[:, 1], ne[:, 1]]).cuda()
    out = m._score(hh, src, dst)
    B = pos.size(0)
    out.retain_grad()
    loss = m.calculate_loss(out[:B], out[B:], k)
    loss.backward()
    with torch.no_grad():
        z = (hh[src] * hh[dst]).double() @ m.predictor.lins[0].weight.double().t() + m.predictor.lins[0].bias.double()
        # the product's OWN hidden pre-activation (its split-bf16 GEMM, float32): the gate it applies is the sign of THIS
        zp = P.ops.gemm([((hh[src] * hh[dst]).detach(), m.predictor.lins[0].weight.detach())], False, True)
        # ... and the activation as the model's forward makes it (bias + relu in the product's epilogue), column 207 kept
        ap = P.ops.gemm([((hh[src] * hh[dst]).detach(), m.predictor.lins[0].weight.detach())], False, True,
                        epilogue=_lib.make_epilogue(bias=m.predictor.lins[0].bias.detach(), relu=True))
        a207 = ap[:, 207].clone()
        x207 = (hh[src] * hh[dst]).detach().abs().sum(1)
        del ap
    if s == 4:
        with torch.no_grad():
            xx = (hh[src] * hh[dst]).double()
            S = xx.abs() @ m.predictor.lins[0].weight.double().abs().t()
            gw = out.grad.detach().double().reshape(-1, 1).abs() * m.predictor.lins[1].weight.detach().double().reshape(1, -1).abs()
            for tau in (1e-10, 1e-9, 1e-8, 1e-7, 1e-6):
                near = z.abs() <= tau * S
                allow = (near * gw).sum(0)
                print(json.dumps({"kink_threshold_relative_to_sum_abs_terms": tau, "elements": int(near.sum()), "of": z.numel(),
                                  "max_column_allowance": float(allow.max()), "allowance_col_207": float(allow[207]),
                                  "columns_with_any": int((allow > 0).sum())}), flush=True)
            del xx, S, gw, near, allow
    res[s] = dict(h=hh.detach().clone(), z_sign=(z > 0), zabs=z.abs(), gb=m.predictor.lins[0].bias.grad.double().clone(),
                  zp=zp, a207=a207, x207=x207, g=out.grad.detach().double().reshape(-1).clone() if out.grad is not None else None,
                  w2=m.predictor.lins[1].weight.detach().double().reshape(-1).clone(), loss=float(loss.detach()))
    del m, hh, out, loss, z
    torch.cuda.empty_cache()
ref = res[4]
g64 = c["grads"]["f64"]["params"]
for s in (3, 5, 7):
    r = res[s]
    flips = (r["z_sign"] != ref["z_sign"])
    d = r["gb"] - ref["gb"]
    col = int(d.abs().argmax())
    print(json.dumps({"slices": s, "vs": 4, "h_max_abs_diff": float((r["h"] - ref["h"]).abs().max()), "loss_diff": r["loss"] - ref["loss"],
                      "hidden_sign_flips": int(flips.sum()), "flips_in_worst_column": int(flips[:, col].sum()),
                      "min_abs_z_at_flips": float(ref["zabs"][flips].min()) if bool(flips.any()) else None,
                      "bias_grad_max_diff": float(d.abs().max()), "worst_column": col, "columns_over_0.01": int((d.abs() > 0.01).sum()),
                      "bias_grad_scale": float(ref["gb"].abs().max())}), flush=True)
    pf = ((r["zp"] > 0) != (ref["zp"] > 0)).nonzero()
    for row, cc in pf[:8].tolist():
        print(json.dumps({"slices": s, "product_gate_flip_at": [row, cc], "z_product_this": float(r["zp"][row, cc]), "z_product_4": float(ref["zp"][row, cc]),
                          "g_row": float(ref["g"][row]), "w2_col": float(ref["w2"][cc]), "g_times_w2": float(ref["g"][row] * ref["w2"][cc]),
                          "bias_grad_diff_in_that_column": float(d[cc])}), flush=True)
    print(json.dumps({"slices": s, "product_gate_flips_total": int(pf.shape[0])}), flush=True)
    rows = ((r["a207"] > 0) != (ref["a207"] > 0)).nonzero().reshape(-1)
    print(json.dumps({"slices": s, "column_207_gate_flips": int(rows.numel()), "w2_207": float(ref["w2"][207]),
                      "sum_g_w2_over_flips": float((ref["g"][rows] * ref["w2"][207] * torch.where(r["a207"][rows] > 0, 1.0, -1.0).double()).sum()),
                      "rows": rows[:6].tolist(), "a_this": r["a207"][rows][:6].tolist(), "a_4": ref["a207"][rows][:6].tolist(),
                      "g": ref["g"][rows][:6].tolist(), "sum_abs_x_row": ref["x207"][rows][:6].tolist(),
                      "max_abs_a207_diff": float((r["a207"] - ref["a207"]).abs().max())}), flush=True)
lib.plnlp_dense_aggregate_tuning(0)
