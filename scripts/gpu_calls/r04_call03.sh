#!/bin/bash
# round 4, call 3: tests again; the stationary-weights kernel by column-tile width x tail launch
O=$GRAFT_REPO_ROOT/gpurun_out/r04c03; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_round4.py -q -k "stationary or random_walk_pairs or driver or full_size" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -15 $O/tests.log
timeout 900 python scripts/bench_gemm.py --math nb --shapes collab_step_fwd,collab_step_dgrad,ddi_pred_fwd,ddi_pred_dgrad,collab_fwd_plain,cit_l2_fwd_k200 > $O/gemm_nb.jsonl 2> $O/gemm_nb.err
cat $O/gemm_nb.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['shape'], r.get('stationary_b'), r['ms'], 'ms', r['TFLOPs'], 'TF', r.get('frac_of_2500'))
"
tail -3 $O/gemm_nb.err
