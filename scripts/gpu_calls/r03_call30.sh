#!/bin/bash
# round 3, call 30: Adam operands requested at the start of the row: tests, the step's launches, step time
O=gpurun_out/r03c30; mkdir -p $O
python -m pytest tests -x -q -m gpu -k "adam or aggregate or fused or captured or trajectory" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 4 $O/tests.log
python scripts/bench_step_launches.py > $O/step_launches.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03c30/step_launches.json"))
for k, v in d.items():
    if isinstance(v, dict):
        print(k, v.get("kernel_ms"), v.get("frac"))
PY
for i in 1 2 3; do
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_collab_$i.json 2>/dev/null
python -c "
import json,sys; r=json.loads(open('$O/bench_collab_$i.json').read().strip().splitlines()[-1]); print('collab', r['ms_per_step'], r['value'], r.get('ms_per_step_full_forward'), r.get('ms_per_step_f32_mfma'))"
done
for w in ddi citation2; do python bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', r['ms_per_step'], r['value'])"; done
