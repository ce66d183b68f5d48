"""GPU pilot for the wide trained-parity legs (tests/trained_parity.py): how many steps / which learning rate bring the ddi
recipe at h = 512 to its plateau, so that the CPU oracle's budget can be planned?  python scripts/pilot_wide_legs.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import plnlp_amd as P
import trained_parity as T

recipe = "ddi_wide"
for lr, epochs, edges in ((0.005, 40, 98304), (0.01, 40, 98304), (0.005, 60, 49152), (0.002, 40, 98304)):
    T.RECIPES[recipe].update(lr=lr, epochs=epochs, train_edges=edges)
    T._problem.clear()
    for seed in (0, 1):
        t = time.time()
        hits, losses = T.run_hip(P, recipe, seed, "bf16x3")
        torch.cuda.synchronize()
        print(recipe, "lr", lr, "epochs", epochs, "edges/epoch", edges, "seed", seed, "s", round(time.time() - t, 1), flush=True)
        for name in ("Hits@20", "Hits@100", "AUC"):
            ki = T.metrics_of(recipe).index(name)
            print("   valid", name, np.round(hits[:, ki, 0], 1).tolist(), flush=True)
        print("   loss", np.round(losses, 0).tolist(), flush=True)
