#!/bin/bash
# round 4, call 7: GEMM tests + sweep after the epilogue batching; the collab step's kernel breakdown
O=$GRAFT_REPO_ROOT/gpurun_out/r04c07; mkdir -p $O
timeout 1800 python -m pytest tests/test_hip_round4.py -q -k "stationary" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/tests.log
timeout 900 python scripts/bench_gemm.py --math nb --shapes collab_step_fwd,collab_step_dgrad,ddi_pred_fwd,ddi_pred_dgrad,cit_in_fwd_k192,cit_l2_fwd_k200 > $O/gemm_nb.jsonl 2> $O/gemm_nb.err
cat $O/gemm_nb.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['shape'], r.get('stationary_b'), r['ms'], 'ms', r['TFLOPs'], 'TF', r.get('frac_of_2500'))
"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o step -- python3 bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 6 45 sequence > $O/step_breakdown_collab.txt
rm -rf $O/prof
head -75 $O/step_breakdown_collab.txt
