#!/bin/bash
# round 5, call 13: the head in the hidden product's epilogue (PLNLP_EPI_ROWDOT): its tests, the predictor parity tests, then ddi
# with the forward fusion on / off on the same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c13; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round5.py -q -m gpu -x -k "head_in_the_hidden or fused_head" > $O/round5.txt 2>&1; tail -12 $O/round5.txt | cut -c1-400
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_round2.py tests/test_hip_round4.py -q -m gpu -x -k "mlp or MLP or predictor or ddi or single_step or trajectory or stationary" > $O/parity.txt 2>&1; tail -4 $O/parity.txt | cut -c1-300
for rep in 1 2 3; do
  for on in 1 0; do
    FWD=$on timeout 600 python - <<PY > $O/ddi_fwd${on}_$rep.json 2> $O/ddi_fwd${on}_$rep.err
import os, sys, runpy
sys.argv = ["bench.py", "--workload", "ddi", "--steps", "30", "--warmup", "8", "--no-cpu-baseline", "--no-parity", "--no-stress", "--no-roofline"]
import plnlp_amd
plnlp_amd.ops.FUSE_HEAD_FORWARD["enabled"] = os.environ["FWD"] == "1"
runpy.run_path("bench.py", run_name="__main__")
PY
    python -c "
import json; r = json.loads(open('$O/ddi_fwd${on}_$rep.json').read().strip().splitlines()[-1]); print('ddi head_in_epilogue=$on rep $rep', round(r['ms_per_step'], 4), 'ms')"
  done
done
