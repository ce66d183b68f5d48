#!/bin/bash
# round 5, call 6: the fused head backward -- its tests, then the ddi bench with the fusion on and off (same box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c06; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round5.py -q -m gpu -x -s -k "not trained_regime_parity" > $O/round5.txt 2>&1; grep -v amdgpu.ids $O/round5.txt | tail -12 | cut -c1-400
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "mlp or MLP or predictor" > $O/parity_mlp.txt 2>&1; tail -3 $O/parity_mlp.txt
for rep in 1 2; do
  for on in 1 0; do
    PLNLP_TEST_FUSE_HEAD=$on timeout 600 python - <<PY > $O/ddi_fuse${on}_$rep.json 2> $O/ddi_fuse${on}_$rep.err
import os, sys, runpy
sys.argv = ["bench.py", "--workload", "ddi", "--steps", "30", "--warmup", "8", "--no-cpu-baseline", "--no-parity", "--no-stress", "--no-roofline"]
import plnlp_amd
plnlp_amd.ops.FUSE_HEAD_BACKWARD["enabled"] = os.environ["PLNLP_TEST_FUSE_HEAD"] == "1"
runpy.run_path("bench.py", run_name="__main__")
PY
    python -c "
import json; r = json.loads(open('$O/ddi_fuse${on}_$rep.json').read().strip().splitlines()[-1]); print('ddi fuse_head=$on rep $rep', round(r['ms_per_step'], 4), 'ms')"
  done
done
