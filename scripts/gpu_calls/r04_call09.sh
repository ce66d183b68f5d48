#!/bin/bash
# round 4, call 9: the whole round-4 test file (trained-regime legs included), then deep vs shallow K loop by tile width
O=$GRAFT_REPO_ROOT/gpurun_out/r04c09; mkdir -p $O
rm -f gpurun_out/trained_parity_r04.txt
( time timeout 2400 python -m pytest tests/test_hip_round4.py -q --durations=8 ) > $O/tests.log 2>&1
echo "tests rc=$?"; tail -30 $O/tests.log
cat gpurun_out/trained_parity_r04.txt
timeout 900 python scripts/bench_gemm.py --math nb --shapes collab_step_fwd,collab_step_dgrad,ddi_pred_fwd,ddi_pred_dgrad,cit_l2_fwd_k200,collab_fwd_plain > $O/gemm_nb.jsonl 2> $O/gemm_nb.err
cat $O/gemm_nb.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['shape'], r.get('stationary_b'), r['ms'], 'ms', r['TFLOPs'], 'TF', r.get('frac_of_2500'))
"
tail -3 $O/gemm_nb.err
