"""Measure -- once, on one MI355X -- which form of the aggregation kernel runs on each benchmark graph at each width, and
write plnlp_amd/agg_forms.json's content (ops.AGG_FORMS): from then on no box measures those shapes again, every box runs the
same kernels on them and two boxes give the same bits.  Three rounds per shape; a shape whose rounds disagree (a near-tie) is
pinned to the majority, ties to the lower form id (0 = the plain one-wave-per-row form).

python scripts/pin_agg_forms.py [--out gpurun_out/agg_forms.json] [--shapes collab,ddi,citation2,uniform_big,rmat23]"""
import argparse
import collections
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PLNLP_AGG_FORMS_FILE"] = "none"          # measure, whatever the shipped table says

import torch  # noqa: E402

import plnlp_amd as P  # noqa: E402
from plnlp_amd import ops, synthetic  # noqa: E402


def graphs(shapes, dev):
    for shape in shapes:
        if shape == "uniform_big":       # bench.py's `roofline` graph
            yield shape, synthetic.uniform_graph(2_927_963, 30_387_995, dev, seed=3), [512]
        elif shape.startswith("rmat"):
            sc = int(shape[4:])
            yield shape, synthetic.rmat_graph(sc, 16 << sc, dev, seed=1), [512]
        else:                            # bench.py's workload graphs (seed 2), at the widths their encoders aggregate
            g = synthetic.make_graph(shape, seed=2, device=dev, weighted=shape == "collab")["adj_t"]
            if shape == "citation2":
                g = P.gcn_normalization(g)
            yield shape, g, {"collab": [256], "ddi": [512], "citation2": [256, 512]}[shape]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/agg_forms.json")
    ap.add_argument("--shapes", default="collab,ddi,citation2,uniform_big,rmat23")
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda")
    forms, detail = {}, {}
    for shape, g, feats in graphs(args.shapes.split(","), dev):
        for feat in feats:
            votes, times = collections.Counter(), []
            for _ in range(args.rounds):
                g._agg_tune.clear()
                ops.AGG_FORMS["measured"].clear()
                picked = ops.tune_aggregation(g, [feat])
                votes[picked[feat]] += 1
                times.append(ops.AGG_FORMS["measured"].get(ops.agg_form_key(g, feat), {}).get("ms"))
            best = sorted(votes.items(), key=lambda kv: (-kv[1], kv[0]))[0][0]
            key = ops.agg_form_key(g, feat)
            forms[key] = int(best)
            detail[key] = {"shape": shape, "feat": feat, "form": ops.describe_form(best), "votes": dict(votes), "best_ms_per_form": times}
            print(key, shape, feat, dict(votes), "->", best, ops.describe_form(best), flush=True)
        del g
        torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump({"note": "aggregation form per (graph shape, width): key = min(rows, cols):max(rows, cols):entries:f<width> "
                           "(ops.agg_form_key); value = PLNLP_AGG_* flag bits (+ ops.AGG_HUB_RANGES); written by "
                           "scripts/pin_agg_forms.py on one MI355X",
                   "forms": forms, "measured": detail}, f, indent=1)


if __name__ == "__main__":
    main()
