"""GPU pilot for the wide trained-parity legs (tests/trained_parity.py): which learning rate / epoch count brings the HIP
path to a non-degenerate Hits@K inside the oracle's CPU budget?  python scripts/pilot_wide_legs.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import plnlp_amd as P
import trained_parity as T

for recipe, lrs, epochs in (("ddi_wide", (0.005, 0.01, 0.02, 0.05), 10), ("collab_wide", (0.01, 0.02), 8)):
    for lr in lrs:
        T.RECIPES[recipe].update(lr=lr, epochs=epochs)
        c0 = P.ops.launch_counts()
        t = time.time()
        hits, losses = T.run_hip(P, recipe, 0, "bf16x3")
        torch.cuda.synchronize()
        c1 = P.ops.launch_counts()
        ki = T.KS.index(T.RECIPES[recipe]["metric"])
        print(recipe, "lr", lr, "s", round(time.time() - t, 1), "valid", T.RECIPES[recipe]["metric"], np.round(hits[:, ki, 0], 1).tolist(),
              "loss", np.round(losses, 1).tolist(), flush=True)
        print("   launches:", {k: c1[k] - c0[k] for k in c1 if c1[k] != c0[k]}, flush=True)
