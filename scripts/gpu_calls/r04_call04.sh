#!/bin/bash
# round 4, call 4: tests; tile-width rule of the stationary-weights GEMM; time of one trained-parity run; the three step benches
O=$GRAFT_REPO_ROOT/gpurun_out/r04c04; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_round4.py -q -k "stationary or random_walk_pairs or driver or full_size" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -12 $O/tests.log
timeout 900 python scripts/bench_gemm.py --math nb --shapes collab_step_fwd,collab_step_dgrad,ddi_pred_fwd,ddi_pred_dgrad,cit_in_fwd_k192,cit_l2_fwd_k200 > $O/gemm_nb.jsonl 2> $O/gemm_nb.err
cat $O/gemm_nb.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['shape'], r.get('stationary_b'), r['ms'], 'ms', r['TFLOPs'], 'TF', r.get('frac_of_2500'))
"
tail -3 $O/gemm_nb.err
python - <<'PY'
import sys, time, torch
sys.path.insert(0, "tests")
import plnlp_amd as P, trained_parity as T
for recipe in ("collab", "ddi"):
    for i in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        h, l = T.run_hip(P, recipe, i, "bf16x3")
        torch.cuda.synchronize(); print(recipe, "seed", i, "run_hip %.2f s" % (time.time() - t0), "final", h[-1].round(2).tolist(), "loss0 %.4f" % l[0])
PY
for w in collab ddi citation2; do
  timeout 900 python bench.py --workload $w --steps 20 --warmup 6 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_$w.json 2> $O/bench_$w.err
  python -c "
import json; r = json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', r['ms_per_step'], 'ms', r['value'] / 1e6, 'M edges/s')
"
done
