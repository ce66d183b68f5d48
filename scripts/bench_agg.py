"""Micro-benchmark of the CSR aggregation kernel (K1) against the HBM roofline.
python scripts/bench_agg.py [--cases collab,uniform,citation2] [--feat 256,512]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plnlp_amd as P
from plnlp_amd import synthetic
from bench import agg_bytes, time_kernel


def uniform_graph(n, m, device, seed=0):
    g = torch.Generator(device=device).manual_seed(seed)
    a = torch.randint(0, n, (m,), generator=g, device=device)
    b = torch.randint(0, n, (m,), generator=g, device=device)
    return P.Graph.from_coo(torch.cat([a, b]), torch.cat([b, a]), None, n, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="collab,uniform,ddi")
    ap.add_argument("--feat", default="256,512")
    ap.add_argument("--weighted", action="store_true")
    ap.add_argument("--tune", default="0", help="comma list of tuning flag sets (plnlp_hip.h: 4 = NT loads, 8 = fewer rows "
                                                "in flight, 12 = both; 'halves' = two launches over column halves)")
    ap.add_argument("--hub-order", default="none", help="comma list: none | <part_rows>:<max_len> (chunks of the long "
                                                        "rows cut by source range, graph.SourceOrderedSplit)")
    ap.add_argument("--threshold", default="0", help="comma list of long-row thresholds (0 = the default rule)")
    args = ap.parse_args()
    dev = torch.device("cuda")
    for case in args.cases.split(","):
        if case == "uniform":
            g = uniform_graph(235868, 1179052, dev)
        elif case == "uniform_big":
            g = uniform_graph(2927963, 30387995, dev)
        elif case.startswith("rmat"):          # rmat<scale>: 2**scale nodes, 16 edges per node (BASELINE config 5, scaled)
            raw = case.endswith("raw")
            sc = int(case[4:].replace("raw", "") or 23)
            g = synthetic.rmat_graph(sc, 16 << sc, dev, seed=1, permute=not raw)
        else:
            g = synthetic.make_graph(case, seed=2, device=dev, weighted=args.weighted)["adj_t"]
        deg = g.degree()
        for feat in [int(f) for f in args.feat.split(",")]:
            x = torch.randn(g.n_cols, feat, device=dev)
            out = torch.empty(g.n_rows, feat, device=dev)
            ref = None
            for thr in [int(v) for v in args.threshold.split(",")]:
              P.ops.SPLIT_THRESHOLD = thr
              for ho in args.hub_order.split(","):
                hub_bit = 0
                if ho != "none":
                    P.ops.HUB_RANGES["part_rows"], P.ops.HUB_RANGES["max_len"] = (int(v) for v in ho.split(":"))
                    P.ops.HUB_RANGES["max_ranges"] = 1 << 30            # the sweep's range size as given
                    hub_bit = P.ops.AGG_HUB_RANGES
                for tune in args.tune.split(","):
                    if tune == "halves":
                        hf = feat // 2
                        def run():
                            P.ops.csr_aggregate(g, x[:, :hf], "mean", args.weighted, out=out[:, :hf])
                            P.ops.csr_aggregate(g, x[:, hf:], "mean", args.weighted, out=out[:, hf:])
                    else:
                        def run(tv=int(tune) | hub_bit):
                            P.ops.csr_aggregate(g, x, "mean", args.weighted, out=out, tune=tv)
                    t = time_kernel(run, iters=10)
                    if ref is None:
                        ref = out.clone()
                    err = float((out - ref).abs().max() / ref.abs().max())
                    by = agg_bytes(g.nnz, g.n_rows, feat, args.weighted)
                    sp = g.row_split(P.ops.split_threshold(g.n_cols), None if ho == "none" else
                                     P.ops.hub_ranges(g.n_cols))
                    print(json.dumps({"case": case, "N": g.n_rows, "nnz": g.nnz, "max_deg": int(deg.max()), "feat": feat,
                                      "tune": tune, "hub_order": ho, "threshold": sp.threshold, "long_rows": sp.n_long,
                                      "chunks": sp.n_chunks, "ms": round(t * 1e3, 4),
                                      "GBps": round(by / t / 1e9, 1), "frac_of_8TBps": round(by / t / 8e12, 4),
                                      "max_dev_vs_first": err,
                                      "source_MiB": round(g.n_cols * feat * 4 / 2 ** 20, 1)}), flush=True)
            del x, out
        del g
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
