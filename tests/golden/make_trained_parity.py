"""Generates tests/golden/g10_trained_parity.npz: Hits@K of the CPU oracle (float32 = the reference's arithmetic,
float64 = the arbiter) on the learnable problem of tests/trained_parity.py, 10 seeds per recipe.  The oracle is
deterministic given the seeds (torch CPU generator streams, CPU float arithmetic), so its per-seed results are data:
the GPU test (tests/test_hip_round3.py::test_trained_regime_hits_parity_over_seeds) trains the HIP path on the same
problem / seeds and is held to these numbers without re-running ~6 CPU-minutes of oracle training on the GPU box
(PLNLP_TRAINED_PARITY_LIVE=1 re-runs the oracle there instead).

    python tests/golden/make_trained_parity.py            (run in the build container; ~10 min on 8 CPUs)"""
import concurrent.futures as cf
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import trained_parity as T

SEEDS = list(range(10))
EPOCHS = {r: c["epochs"] for r, c in T.RECIPES.items()}
KS = ("Hits@20", "Hits@50", "Hits@100")


def main():
    out = {"seeds": np.array(SEEDS), "ks": np.array([20, 50, 100])}
    only = os.environ.get("ONLY")            # ONLY=ddi: recompute one recipe, keep the other from the existing file
    path = os.path.join(HERE, "g10_trained_parity.npz")
    if only:
        old = np.load(path)
        out.update({k: old[k] for k in old.files if not k.startswith(only) and k != "hyper"})
    jobs = [(recipe, s, dt, EPOCHS[recipe]) for recipe in EPOCHS if only in (None, recipe) for dt in ("f64", "f32") for s in SEEDS]
    with cf.ProcessPoolExecutor(max_workers=int(os.environ.get("WORKERS", "4")), mp_context=mp.get_context("spawn")) as pool:
        res = list(pool.map(T.run_oracle, jobs))
    for (recipe, s, dt, ep), r in zip(jobs, res):
        out.setdefault(f"{recipe}_{dt}", np.zeros((len(SEEDS), 3, 2)))[SEEDS.index(s)] = [r[k] for k in KS]
    for recipe, ep in EPOCHS.items():
        out[f"{recipe}_epochs"] = np.array(ep)
        out[f"{recipe}_hyper"] = np.array([T.RECIPES[recipe]["lr"], T.RECIPES[recipe]["clip"], ep])
    out["problem"] = np.array([T.PROBLEM["num_nodes"], T.PROBLEM["community"], T.PROBLEM["seed"]])
    out["hyper"] = np.array([T.H, T.B, T.PROBLEM["p_in"], T.PROBLEM["cross_per_node"], T.PROBLEM["unlearnable"]])
    np.savez_compressed(os.path.join(HERE, "g10_trained_parity.npz"), **out)
    for key in sorted(out):
        if key.endswith(("_f32", "_f64")):
            print(key, "mean over seeds (valid, test) per K:", out[key].mean(0).round(2).tolist())


if __name__ == "__main__":
    main()
