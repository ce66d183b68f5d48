#!/bin/bash
mkdir -p gpurun_out/c29
timeout 900 python -m pytest tests/test_hip_round2.py -q -m gpu -x -k "pair_gathered or fused_edge_mlp" 2>&1 | tail -30 > gpurun_out/c29/tests.log
cat gpurun_out/c29/tests.log
for f in 1 0; do
  PLNLP_FUSE_EDGE_MLP=$f timeout 600 python bench.py --workload ddi --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > gpurun_out/c29/bench_ddi_fuse$f.json 2> gpurun_out/c29/bench_ddi_fuse$f.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/c29/bench_ddi_fuse$f.json").read().strip().splitlines()[-1])
print("ddi fuse=$f", round(d["ms_per_step"],3), "ms", round(d["value"]/1e6,2), "M edges/s")
PY
done
