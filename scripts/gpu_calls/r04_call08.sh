#!/bin/bash
# round 4, call 8: the deep K loop (two steps of load look-ahead, counted vmcnt) of the stationary-weights GEMM: tests, then
# deep vs shallow vs the tile kernel by tile width
O=$GRAFT_REPO_ROOT/gpurun_out/r04c08; mkdir -p $O
timeout 1800 python -m pytest tests/test_hip_round4.py -q -k "stationary" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -6 $O/tests.log
timeout 900 python scripts/bench_gemm.py --math nb --shapes collab_step_fwd,collab_step_dgrad,ddi_pred_fwd,ddi_pred_dgrad,cit_l2_fwd_k200,collab_fwd_plain > $O/gemm_nb.jsonl 2> $O/gemm_nb.err
cat $O/gemm_nb.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['shape'], r.get('stationary_b'), r['ms'], 'ms', r['TFLOPs'], 'TF', r.get('frac_of_2500'))
"
tail -3 $O/gemm_nb.err
