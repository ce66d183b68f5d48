#!/bin/bash
# round 4, call 1: the new round-4 tests that need no oracle fixture (stationary-weights GEMM vs the tile kernel, random-walk
# pairs, driver loop vs oracle, full-size steps), then the GEMM A/B on the step's shapes and a default bench line
O=$GRAFT_REPO_ROOT/gpurun_out/r04c01; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_round4.py -q -x -k "stationary or random_walk_pairs or driver or full_size" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -25 $O/tests.log
timeout 600 python scripts/bench_gemm.py --math st --error --shapes collab_step_fwd,collab_step_dgrad,ddi_pred_fwd,ddi_pred_dgrad,ddi_enc_fwd,cit_in_fwd_k192,cit_l2_fwd_k200,collab_fwd_plain > $O/gemm_st.jsonl 2> $O/gemm_st.err
cat $O/gemm_st.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['shape'], 'stationary' if r.get('stationary_b') else 'tile      ', r['ms'], 'ms', r['TFLOPs'], 'TF', r.get('frac_of_2500'), 'err', r.get('max_err_over_sum_abs'))
"
tail -3 $O/gemm_st.err
timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress > $O/bench_collab.json 2> $O/bench_collab.err
python -c "
import json; r = json.loads(open('$O/bench_collab.json').read().strip().splitlines()[-1]); print('collab', r['ms_per_step'], 'ms', r['value'] / 1e6, 'M edges/s', r.get('roofline_mfma'))
"
