#!/bin/bash
# round 5, call 16: the stationary GEMMs' weight images built ahead of the product on the second side stream
# (plnlp_gemm_operand.b_terms_phase, ops.STEP_WEIGHTS): determinism / step tests, then collab and ddi on / off on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c16; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_round2.py tests/test_hip_round3.py tests/test_hip_round4.py -q -m gpu -x -k "step or trajectory or deterministic or stale or stationary or side_stream or capture or pipeline or driver" > $O/steps.txt 2>&1; tail -4 $O/steps.txt | cut -c1-300
run() {  # name workload on
  AHEAD=$3 timeout 600 python - <<PY > $O/ab_$1.json 2> $O/ab_$1.err
import os, sys, runpy
sys.argv = ["bench.py", "--workload", "$2", "--steps", "40", "--warmup", "10", "--no-cpu-baseline", "--no-parity", "--no-stress", "--no-roofline"]
import plnlp_amd
plnlp_amd.ops.STEP_WEIGHTS["enabled"] = os.environ["AHEAD"] == "1"
runpy.run_path("bench.py", run_name="__main__")
PY
  python -c "
import json; r = json.loads(open('$O/ab_$1.json').read().strip().splitlines()[-1]); print('$2 image_ahead=$3:', round(r['ms_per_step'], 4), 'ms', 'epoch', round(r.get('train_epoch', {}).get('ms_per_step', 0), 4))"
}
for rep in 1 2 3; do
  run c_on$rep collab 1
  run c_off$rep collab 0
done
for rep in 1 2; do
  run d_on$rep ddi 1
  run d_off$rep ddi 0
done
