"""Generates tests/golden/g11_trained_curves.npz: per-seed, per-epoch Hits@20/50/100 (valid, test) and epoch losses of
the CPU oracle -- float32 (the reference's arithmetic) and float64 -- on the two recipes of tests/trained_parity.py,
64 / 48 seeds.  The oracle is deterministic given the seeds (CPU generator streams, counter-hash walks, one thread), so
its curves are data: tests/test_hip_round4.py trains the HIP path on the same problem / seeds on the GPU box and is held
to them without re-running ~35 CPU-minutes of oracle training there.

    python tests/golden/make_trained_curves.py          (build container; WORKERS=6 by default)"""
import concurrent.futures as cf
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import trained_parity as T

# WIDE=1: the wide legs of round 5 (collab_wide, ddi_wide: the recipes at h = 256 / 512 on graphs big enough for the
# benchmark's kernels) into g12_trained_curves_wide.npz -- ~7 CPU-hours of single-thread oracle runs, 7 workers
WIDE = os.environ.get("WIDE") == "1"
NAMES = [r for r in T.RECIPES if (r in T.WIDE) == WIDE]
OUT = "g12_trained_curves_wide.npz" if WIDE else "g11_trained_curves.npz"
N_SEEDS = {r: int(os.environ.get("SEEDS", T.RECIPES[r]["seeds"])) for r in NAMES}


def main():
    t0 = time.time()
    out = {"ks": np.array([20, 50, 100])}
    only = os.environ.get("ONLY")            # ONLY=ddi: recompute one recipe, keep the other from the existing file
    if only:
        old = np.load(os.path.join(HERE, OUT))
        out.update({k: old[k] for k in old.files if not k.startswith(only)})
    jobs = [(recipe, s, dt) for dt in ("f64", "f32") for recipe in NAMES if only in (None, recipe)
            for s in range(N_SEEDS[recipe])]          # (the float64 runs first: they are the long ones)
    if WIDE:
        for recipe in NAMES:                          # build (and cache) the problems once, before the workers race for them
            T.problem(recipe)
    with cf.ProcessPoolExecutor(max_workers=int(os.environ.get("WORKERS", "6")), mp_context=mp.get_context("spawn")) as pool:
        res = list(pool.map(T.run_oracle, jobs, chunksize=1))
    for (recipe, s, dt), (hits, losses) in zip(jobs, res):
        e = T.RECIPES[recipe]["epochs"]
        out.setdefault(f"{recipe}_{dt}", np.zeros((N_SEEDS[recipe], e, len(T.metrics_of(recipe)), 2), np.float32))[s] = hits
        out.setdefault(f"{recipe}_{dt}_loss", np.zeros((N_SEEDS[recipe], e), np.float64))[s] = losses
    for recipe in NAMES:
        r = T.RECIPES[recipe]
        if only in (None, recipe):
            out[f"{recipe}_hyper"] = np.array([r["lr"], r["clip"], r["epochs"], r["batch"], r["walk_length"], r["k"],
                                               float(r["decay"])])
            out[f"{recipe}_problem"] = np.array([float(v) for v in T.PROBLEMS[r["problem"]].values()])
    np.savez_compressed(os.path.join(HERE, OUT), **out)
    for recipe in NAMES:
        for dt in ("f32", "f64"):
            f = T.final_level(out[f"{recipe}_{dt}"].astype(np.float64), recipe)
            print(recipe, dt, "final level (valid, test) mean", f.mean(0).round(2), "std", f.std(0, ddof=1).round(2))
    print(f"{time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
