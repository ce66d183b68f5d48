#!/bin/bash
# round 5, call 15: the GCN layers' bias column sums on the second side stream: GCN parity tests, then citation2 on / off (same box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c15; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_round2.py tests/test_hip_round4.py tests/test_hip_round5.py -q -m gpu -x -k "gcn or GCN or citation2 or deterministic or stale or padded or unaligned" > $O/gcn.txt 2>&1; tail -4 $O/gcn.txt | cut -c1-300
for rep in 1 2; do
  for on in 1 0; do
    CSUM=$on timeout 600 python - <<PY > $O/cit_csum${on}_$rep.json 2> $O/cit_csum${on}_$rep.err
import os, sys, runpy
sys.argv = ["bench.py", "--workload", "citation2", "--steps", "12", "--warmup", "5", "--no-cpu-baseline", "--no-parity", "--no-stress", "--no-roofline"]
import plnlp_amd
plnlp_amd.ops.COLSUM_SIDE_STREAM["enabled"] = os.environ["CSUM"] == "1"
runpy.run_path("bench.py", run_name="__main__")
PY
    python -c "
import json; r = json.loads(open('$O/cit_csum${on}_$rep.json').read().strip().splitlines()[-1]); print('citation2 colsum_side=$on rep $rep', round(r['ms_per_step'], 3), 'ms')"
  done
done
