#!/bin/bash
# round 3, call 25: new aggregation forms under test, the default bench with the widened tuner, whole suite
O=gpurun_out/r03c25; mkdir -p $O
python -m pytest tests/test_hip_round3.py -x -q -m gpu -k "xcd_pinned" > $O/new_tests.log 2>&1; echo "rc=$?" >> $O/new_tests.log
tail -n 5 $O/new_tests.log
python bench.py --steps 20 --warmup 5 > $O/bench_collab.json 2> $O/bench_collab.err; echo "rc=$?" >> $O/bench_collab.err
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r03c25/bench_collab.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "ms_per_step_f32_mfma", "ms_per_step_full_forward", "train_epoch", "eval_scoring", "host_busy_ms_per_step", "step_capture"):
    print(k, r.get(k))
for k in ("roofline", "roofline_workload_agg", "roofline_agg_adam", "roofline_mfma"):
    v = r.get(k) or {}
    print(k, {a: v.get(a) for a in ("kernel_ms", "achieved", "frac", "kernel_form")})
PY
PLNLP_AGG_AUTOTUNE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_collab_notune.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_collab_tuned.json 2>/dev/null
PLNLP_AGG_AUTOTUNE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_collab_notune2.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_collab_tuned2.json 2>/dev/null
for f in notune tuned notune2 tuned2; do python -c "
import json,sys; r=json.loads(open('$O/bench_collab_$f.json').read().strip().splitlines()[-1]); print('$f', r['ms_per_step'], r['value'])"; done
python -m pytest tests -x -q -m gpu --durations=6 > $O/suite.log 2>&1; echo "rc=$?" >> $O/suite.log
tail -n 12 $O/suite.log
cp gpurun_out/trained_parity_table.txt $O/ 2>/dev/null
