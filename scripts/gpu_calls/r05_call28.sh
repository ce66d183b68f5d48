#!/bin/bash
# wide weight gradient after the pair reduce + branch-free gather: its tests, then collab / ddi on / off, same box
O=gpurun_out/r05w; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round5.py -q -x -k "wide_weight" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for rep in 1 2 3; do
  for wl in collab ddi; do
    for w in 1 0; do
      PLNLP_GEMM_WIDE_WGRAD=$w timeout 900 python bench.py --workload $wl --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/ab2_${wl}_wide${w}_$rep.json 2> $O/ab2_${wl}_wide${w}_$rep.err
      python - <<PY
import json
try:
    d=json.loads(open("$O/ab2_${wl}_wide${w}_$rep.json").read().strip().splitlines()[-1]); print("$wl wide=$w rep=$rep", round(d["ms_per_step"],4), "ms | epoch", round(d.get("train_epoch",{}).get("ms_per_step",0),4))
except Exception as e:
    print("$wl wide=$w rep=$rep failed", e); print(open("$O/ab2_${wl}_wide${w}_$rep.err").read()[-600:])
PY
    done
  done
done
