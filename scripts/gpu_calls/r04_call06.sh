#!/bin/bash
# round 4, call 6: round-4 tests (all but the trained-regime ones), tile-width sweep after the epilogue fix, the bench
# through a one-rank RCCL group (cost-model inputs measured in the run, phase timers)
O=$GRAFT_REPO_ROOT/gpurun_out/r04c06; mkdir -p $O
timeout 1800 python -m pytest tests/test_hip_round4.py -q -k "not trained" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -8 $O/tests.log
timeout 900 python scripts/bench_gemm.py --math nb --shapes collab_step_fwd,collab_step_dgrad,ddi_pred_fwd,cit_in_fwd_k192,cit_l2_fwd_k200 > $O/gemm_nb.jsonl 2> $O/gemm_nb.err
cat $O/gemm_nb.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['shape'], r.get('stationary_b'), r['ms'], 'ms', r['TFLOPs'], 'TF', r.get('frac_of_2500'))
"
tail -3 $O/gemm_nb.err
for ex in grads shard; do
  timeout 900 python bench.py --force-dist --dp-exchange $ex --steps 20 --warmup 6 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_collab_${ex}_1rank.json 2> $O/bench_collab_${ex}.err
  tail -2 $O/bench_collab_${ex}.err
  python -c "
import json; r = json.loads(open('$O/bench_collab_${ex}_1rank.json').read().strip().splitlines()[-1]); print('$ex', r['ms_per_step'], 'ms'); print(json.dumps(r.get('dp_prediction'))); print(json.dumps(r.get('dp_phases')))
"
done
for w in collab ddi citation2; do
  timeout 900 python bench.py --workload $w --steps 20 --warmup 6 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_$w.json 2> $O/bench_$w.err
  python -c "
import json; r = json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', r['ms_per_step'], 'ms', r['value'] / 1e6, 'M edges/s')
"
done
