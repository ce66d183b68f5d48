"""What can a ddi_wide statistic see?  (VERDICT r5 #2c; provenance of the thresholds in tests/test_hip_round6.py)
The 8 seeds of the ddi_wide leg (tests/trained_parity.py; fixture g12 holds the oracle's float32 / float64 curves) on the HIP path,
clean and MUTATED (every dense operand rounded to ONE bf16 term): per epoch, over the seeds, the paired differences of AUC,
Hits@20 and the epoch loss against the float32 oracle, beside the oracle's own float32 - float64 differences."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import plnlp_amd as P
import trained_parity as T

recipe = sys.argv[1] if len(sys.argv) > 1 else "ddi_wide"
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
g12 = np.load(os.path.join(ROOT, "tests", "golden", "g12_trained_curves_wide.npz"))
ref32, ref64 = g12[f"{recipe}_f32"].astype(np.float64)[:seeds], g12[f"{recipe}_f64"].astype(np.float64)[:seeds]
l32, l64 = g12[f"{recipe}_f32_loss"][:seeds], g12[f"{recipe}_f64_loss"][:seeds]
out = {}
for arm in ("none", "bf16_operands"):
    runs = [T.run_hip(P, recipe, s, "bf16x3", mutation=arm) for s in range(seeds)]
    out[arm] = (np.stack([h for h, _ in runs]), np.stack([l for _, l in runs]))
ms = T.metrics_of(recipe)
ka, k20 = ms.index("AUC"), ms.index("Hits@20")
print(f"{recipe}: {seeds} seeds; per epoch: mean over seeds of |x - oracle f32| (valid) -- clean HIP | mutated HIP | oracle f64 (its own gap)")
print("epoch   AUC: clean  mutated  f64-gap     Hits@20: clean  mutated  f64-gap     rel. loss: clean  mutated  f64-gap    mean AUC f32")
for e in range(ref32.shape[1]):
    row = []
    for k in (ka, k20):
        row += [np.abs(out["none"][0][:, e, k, 0] - ref32[:, e, k, 0]).mean(), np.abs(out["bf16_operands"][0][:, e, k, 0] - ref32[:, e, k, 0]).mean(),
                np.abs(ref64[:, e, k, 0] - ref32[:, e, k, 0]).mean()]
    row += [(np.abs(out["none"][1][:, e] - l32[:, e]) / l32[:, e]).mean(), (np.abs(out["bf16_operands"][1][:, e] - l32[:, e]) / l32[:, e]).mean(),
            (np.abs(l64[:, e] - l32[:, e]) / l32[:, e]).mean()]
    print(f"{e + 1:5d}   {row[0]:10.3f} {row[1]:8.3f} {row[2]:8.3f}   {row[3]:14.3f} {row[4]:8.3f} {row[5]:8.3f}   {row[6]:16.2e} {row[7]:8.2e} {row[8]:8.2e}   {ref32[:, e, ka, 0].mean():8.2f}")
np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"calibrate_{recipe}.npz"), clean=out["none"][0], clean_loss=out["none"][1],
                    mutated=out["bf16_operands"][0], mutated_loss=out["bf16_operands"][1])
