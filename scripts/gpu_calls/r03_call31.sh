#!/bin/bash
# round 3, call 31: whole GPU suite + default bench after the aggregation changes
O=gpurun_out/r03c31; mkdir -p $O
python -m pytest tests -x -q -m gpu --durations=8 > $O/suite.log 2>&1; echo "rc=$?" >> $O/suite.log
tail -n 14 $O/suite.log
cp gpurun_out/trained_parity_table.txt $O/ 2>/dev/null
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err; tail -n 3 $O/bench_default.err
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r03c31/bench_default.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "ms_per_step_f32_mfma", "ms_per_step_full_forward", "host_busy_ms_per_step"):
    print(k, r.get(k))
print("train_epoch", r["train_epoch"]["value"], "eval", r["eval_scoring"]["value"])
print("capture other", r["step_capture"]["other_path"])
for k in ("roofline", "roofline_workload_agg", "roofline_agg_adam", "roofline_mfma"):
    v = r.get(k) or {}
    print(k, {a: v.get(a) for a in ("kernel_ms", "achieved", "frac", "kernel_form")})
print("cpu", r["cpu_baseline"]["value"], r["config"])
PY
