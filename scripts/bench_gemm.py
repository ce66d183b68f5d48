"""Micro-benchmark of the f32-MFMA GEMM (K4) on the shapes of the hot path."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plnlp_amd as P
from plnlp_amd import _lib
from bench import time_kernel

SHAPES = {
    # name: (M, N, K-segments, a_trans, b_trans, epilogue)
    "collab_fwd": (235868, 256, [256, 256], False, True, True),
    "collab_fwd_bias_relu": (235868, 256, [256, 256], False, True, "bias_relu"),
    "collab_fwd_plain": (235868, 256, [256, 256], False, True, False),
    "collab_dgrad": (235868, 256, [256], False, False, False),
    "collab_wgrad": (256, 256, [235868], True, False, False),
    "ddi_pred_fwd": (262144, 512, [512], False, True, True),
    "ddi_pred_wgrad": (512, 512, [262144], True, False, False),
    "ddi_enc_fwd": (4267, 512, [512, 512], False, True, True),
    "ddi_enc_dgrad": (4267, 1024, [512], False, False, False),
    "ddi_enc_fwd_l1": (4267, 512, [512, 512], False, True, "bias_relu"),
    "square4k": (4096, 4096, [4096], False, True, False),
    # citation2 (GCN h=200): first layer over [A emb | A x] (52 + 128 = 180 columns), padded to 192, second layer
    "cit_in_fwd_k180": (2927963, 200, [180], False, True, "bias_relu"),
    "cit_in_fwd_k192": (2927963, 200, [192], False, True, "bias_relu"),
    "cit_l2_fwd_k200": (2927963, 200, [200], False, True, False),
    # its two weight gradients: dz^T [A emb | A x] (200 x 180) and (A^T dz)^T h (200 x 200) over all 2.9 M rows
    "cit_in_wgrad": (200, 180, [2927963], True, False, False),
    "cit_l2_wgrad": (200, 200, [2927963], True, False, False),
    "wgrad_224": (224, 224, [2927963], True, False, False),
    "collab_wgrad_256x512": (256, 512, [132224], True, False, False),
    # the same product with A's rows gathered from a 65 536-row table (50 MB: cache-resident) -- what the launch costs when
    # the activation matrix does not come from HBM
    "cit_l2_fwd_k200_cached": (2927963, 200, [200], False, True, "gather0"),
    # tail quantisation probes: 3584 tiles = exactly 7 rounds of the 512 workgroup slots, vs 7.2 rounds above
    "collab_fwd_7rounds": (229376, 256, [256, 256], False, True, True),
    "collab_fwd_plain_7rounds": (229376, 256, [256, 256], False, True, False),
    # the collab STEP's own launches: the conv at the ~132 K touched rows (root rows gathered from the 235 868-row table)
    "collab_step_fwd": (132224, 256, [256, 256], False, True, "gather"),
    "collab_step_dgrad": (132224, 512, [256], False, False, False),
    "ddi_pred_dgrad": (262144, 512, [512], False, False, False),
    "collab_dgrad_T": (131072, 512, [256], False, False, False),
    "collab_wgrad_T": (256, 512, [131072], True, False, False),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--min-rows", type=int, default=0, help="nb mode: rows of A from which the stationary-weights form applies (0: the default 16384)")
    ap.add_argument("--repeats", type=int, default=3, help="timed blocks per shape; the median is reported")
    ap.add_argument("--math", default="env", choices=["env", "f32", "bf16x3", "ab", "st", "nb", "wide", "blk"],
                    help="how the products are formed (ops.GEMM_MATH); ab = measure both, interleaved")
    ap.add_argument("--error", action="store_true",
                    help="also report max |C - C_fp64| / sum_k |a||b| over 64 sampled result rows (no epilogue)")
    args = ap.parse_args()
    dev = torch.device("cuda")
    # bring the clocks up first: the first shape of a cold process measured 10-15 % low
    wa, wb = torch.randn(4096, 4096, device=dev), torch.randn(4096, 4096, device=dev)
    for _ in range(60):
        P.ops.gemm([(wa, wb)], False, True)
    torch.cuda.synchronize()
    del wa, wb
    for name in args.shapes.split(","):
        m, n, ks, at, bt, epi = SHAPES[name]
        segs = []
        for k in ks:
            a = torch.randn((k, m) if at else (m, k), device=dev)
            b = torch.randn((n, k) if bt else (k, n), device=dev)
            segs.append((a, b))
        bias = torch.randn(n, device=dev)
        e = None
        if epi == "bias_relu":
            e = _lib.make_epilogue(bias=bias, relu=True)
        elif epi and epi != "gather0":
            e = _lib.make_epilogue(bias=bias, relu=True, dropout_p=0.3, dropout_seed=1)
        a_index = None
        if epi == "gather0":
            table = torch.randn(65536, ks[0], device=dev)
            rows = (torch.arange(m, device=dev) % 65536).to(torch.int32)
            segs[0] = (table, segs[0][1])
            a_index = [rows]
            e = None
        if epi == "gather":           # the second segment's A rows gathered from a larger table
            table = torch.randn(235868, ks[1], device=dev)
            rows = torch.randperm(235868, device=dev)[:m].sort().values.to(torch.int32)
            segs[1] = (table, segs[1][1])
            a_index = [None, rows]
        out = torch.empty(m, n, device=dev)
        flop = 2.0 * m * n * sum(ks)
        maths = {"env": [None], "f32": ["f32"], "bf16x3": ["bf16x3"], "ab": ["f32", "bf16x3"], "st": [], "nb": [], "wide": [], "blk": []}[args.math]
        modes = [(mt, None) for mt in maths]
        if args.math == "st":         # split-bf16 products: the 128 x 128 kernels vs the stationary-weights kernel
            modes = [("bf16x3", False), ("bf16x3", True)]
        if args.math == "nb":         # the stationary-weights kernel by column-tile width, tail launch on / off, vs the tile kernel
            modes = [("bf16x3", False)] + [("bf16x3", (nb, args.min_rows)) for nb in (0, 8, 4, 2, 1)]
        if args.math == "wide":       # weight gradients 129 .. 224 wide: the 128 x 128 kernels vs one workgroup per result
            modes = [("bf16x3", "tile"), ("bf16x3", "wide")]
        if args.math == "blk":        # the stationary-weights product: 128-row kernel (x3s) vs whole 256-row blocks (x3b), +- half blocks
            modes = [("bf16x3", "x3s"), ("bf16x3", "x3b"), ("bf16x3", "x3b_nohalf")]
        ts = {md: [] for md in modes}

        def arm(md):
            if md[0] is not None:
                P.ops.GEMM_MATH["mode"] = md[0]
            if md[1] in ("tile", "wide"):
                P.ops.GEMM_WIDE_WGRAD["enabled"] = md[1] == "wide"
            elif md[1] in ("x3s", "x3b", "x3b_nohalf"):
                P.ops.GEMM_BLOCK["mode"] = {"x3s": "off", "x3b": "all", "x3b_nohalf": "all-nolead"}[md[1]]
            elif md[1] is not None:
                P.ops.GEMM_STATIONARY_B["enabled"] = bool(md[1])
                nb, tail = md[1] if isinstance(md[1], tuple) else (0, 1)
                _lib.load().plnlp_gemm_stationary_tuning(nb, tail)
                if tail > 0:
                    P.ops.GEMM_STATIONARY_B["min_rows"] = tail
        for _ in range(args.repeats):                     # interleaved: every arm sees the same clocks
            for md in modes:
                arm(md)
                ts[md].append(time_kernel(lambda: P.ops.gemm(segs, at, bt, out=out, epilogue=e, a_index=a_index), iters=args.iters))
        if args.math == "blk":        # same bits, whichever kernel ran
            outs = []
            for md in modes:
                arm(md)
                outs.append(P.ops.gemm(segs, at, bt, epilogue=e, a_index=a_index).clone())
            torch.cuda.synchronize()
            same = [bool(torch.equal(outs[0], o)) for o in outs[1:]]
            print(json.dumps({"shape": name, "bits_equal_x3s": same,
                              "max_abs_diff": [float((outs[0] - o).abs().max()) for o in outs[1:]]}), flush=True)
            del outs
        for md in modes:
            t = sorted(ts[md])[len(ts[md]) // 2]
            rec = {"shape": name, "M": m, "N": n, "K": ks, "ms": round(t * 1e3, 4),
                   "TFLOPs": round(flop / t / 1e12, 2), "frac_of_157": round(flop / t / 157.3e12, 3),
                   "math": md[0] or P.ops.GEMM_MATH["mode"]}
            if md[1] is not None:
                rec["stationary_b"] = md[1]
            if md[0] == "bf16x3" or (md[0] is None and P.ops.GEMM_MATH["mode"] == "bf16x3"):
                rec["bf16_TFLOPs_executed"] = round(6 * flop / t / 1e12, 1)
                rec["frac_of_2500"] = round(6 * flop / t / 2.5e15, 3)
            if args.error:
                arm(md)
                got = P.ops.gemm(segs, at, bt, a_index=a_index)
                rows = torch.randperm(m, device=dev)[:64]
                ref = torch.zeros(rows.numel(), n, dtype=torch.float64, device=dev)
                mag = torch.zeros_like(ref)
                for a, b in segs:
                    if a_index is not None and a is segs[1][0]:
                        a = a[a_index[1].long()]
                    a64 = (a[:, rows].T if at else a[rows]).double()
                    b64 = (b.T if bt else b).double()
                    ref += a64 @ b64
                    mag += a64.abs() @ b64.abs()
                rec["max_err_over_sum_abs"] = float(((got[rows].double() - ref).abs() / mag).max())
                rec["rms_err_over_sum_abs"] = float((((got[rows].double() - ref) / mag) ** 2).mean().sqrt())
            print(json.dumps(rec), flush=True)
        del segs, out


if __name__ == "__main__":
    main()
