"""What the trained-regime harness (tests/trained_parity.py) can and cannot see: the collab leg re-run on deliberately
degraded products of increasing severity, each compared with the oracle fixture exactly as the test compares the clean
product.  Output: one describe() block per mutation -> profiles/r04_trained_parity.md."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import plnlp_amd as P
import trained_parity as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g11 = np.load(os.path.join(ROOT, "tests", "golden", "g11_trained_curves.npz"))
ref32, ref64 = g11["collab_f32"][:n].astype(float), g11["collab_f64"][:n].astype(float)
for mutation in ("none", "bf16_operands", "agg_noise:0.001", "agg_noise:0.01", "agg_noise:0.03", "agg_noise:0.1"):
    runs = [T.run_hip(P, "collab", s, "bf16x3", mutation) for s in range(n)]
    hip = np.stack([h for h, _ in runs])
    c = T.compare(hip, ref32, ref64, "collab")
    verdict = "FAILS the +-0.3 check" if np.abs(c["diff_f32"]).max() > 0.3 else "passes"
    print(T.describe("collab recipe, mutation %-16s -> %s" % (mutation, verdict), c), flush=True)
