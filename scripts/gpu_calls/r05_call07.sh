#!/bin/bash
# round 5, call 7: the padded embedding table -- its test, the multirank file (citation2 case now with a 42-wide table), the
# whole suite, then citation2 with the table padded / plain on the same box, and the ddi / citation2 step breakdowns
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c07; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round5.py -q -m gpu -x -k "not trained_regime_parity" > $O/round5.txt 2>&1; tail -5 $O/round5.txt | cut -c1-300
timeout 900 python -m pytest tests/test_hip_multirank.py -q -m gpu > $O/multirank.txt 2>&1; tail -3 $O/multirank.txt | cut -c1-300
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_hip_multirank.py --deselect tests/test_hip_round5.py > $O/gpu_suite.txt 2>&1; tail -8 $O/gpu_suite.txt | cut -c1-300
for rep in 1 2; do
  for on in 1 0; do
    PLNLP_TEST_PAD=$on timeout 600 python - <<PY > $O/cit_pad${on}_$rep.json 2> $O/cit_pad${on}_$rep.err
import os, sys, runpy
sys.argv = ["bench.py", "--workload", "citation2", "--steps", "12", "--warmup", "5", "--no-cpu-baseline", "--no-parity", "--no-stress", "--no-roofline"]
import plnlp_amd
from plnlp_amd import model
model.PAD_EMBEDDING_TABLE["enabled"] = os.environ["PLNLP_TEST_PAD"] == "1"
runpy.run_path("bench.py", run_name="__main__")
PY
    python -c "
import json; r = json.loads(open('$O/cit_pad${on}_$rep.json').read().strip().splitlines()[-1]); print('citation2 padded_table=$on rep $rep', round(r['ms_per_step'], 3), 'ms')"
  done
done
for w in ddi citation2; do
  rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o p -- python3 bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
  f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 5 40 > $O/step_breakdown_$w.txt
  rm -rf $O/prof
  head -32 $O/step_breakdown_$w.txt | cut -c1-150
done
