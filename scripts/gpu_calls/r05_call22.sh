#!/bin/bash
# wide weight gradient with its own slice reduce: tests, microbench, then the citation2 step with the form on / off (same box)
mkdir -p gpurun_out/r05w
timeout 900 python -m pytest tests/test_hip_round5.py -q -x -k "wide_weight" > gpurun_out/r05w/tests.txt 2>&1
tail -3 gpurun_out/r05w/tests.txt
timeout 600 python scripts/bench_gemm.py --shapes cit_in_wgrad,cit_l2_wgrad --math wide --iters 5 2>&1 | grep shape | cut -c1-140
for rep in 1 2; do
  for w in 1 0; do
    PLNLP_GEMM_WIDE_WGRAD=$w timeout 900 python bench.py --workload citation2 --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > gpurun_out/r05w/cit_wide${w}_$rep.json 2> gpurun_out/r05w/cit_wide${w}_$rep.err
    python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05w/cit_wide${w}_$rep.json").read().strip().splitlines()[-1]); print("wide=$w rep=$rep", d["ms_per_step"], d["value"])
except Exception as e:
    print("wide=$w rep=$rep failed", e); print(open("gpurun_out/r05w/cit_wide${w}_$rep.err").read()[-800:])
PY
  done
done
