#!/bin/bash
mkdir -p gpurun_out/c21
for m in f32 bf16x3; do
 for w in collab ddi citation2; do
  PLNLP_GEMM_MATH=$m timeout 900 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > gpurun_out/c21/bench_${w}_$m.json 2> gpurun_out/c21/bench_${w}_$m.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/c21/bench_${w}_$m.json").read().strip().splitlines()[-1])
print("$w $m", round(d["ms_per_step"],3), "ms", round(d["value"]/1e6,2), "M edges/s")
PY
 done
done
