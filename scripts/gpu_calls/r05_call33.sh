#!/bin/bash
# citation2: the 50-wide embedding table padded to 52 (PLNLP_EMB_PAD=4) vs 64 columns (16), same box, interleaved
O=gpurun_out/r05w; mkdir -p $O
for rep in 1 2 3; do
  for g in 16 4; do
    PLNLP_EMB_PAD=$g timeout 900 python bench.py --workload citation2 --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/pad_${g}_$rep.json 2> $O/pad_${g}_$rep.err
    python - <<PY
import json
try:
    d=json.loads(open("$O/pad_${g}_$rep.json").read().strip().splitlines()[-1]); print("pad=$g rep=$rep", round(d["ms_per_step"],4), "ms | epoch", round(d.get("train_epoch",{}).get("ms_per_step",0),4), "loss", d.get("final_loss"))
except Exception as e:
    print("pad=$g rep=$rep failed", e); print(open("$O/pad_${g}_$rep.err").read()[-800:])
PY
  done
done
