#!/bin/bash
# round 3, call 39: tuner with warm-up / interleaved rounds; PMC of the step's launches in the tuned form; default bench
R=gpurun_out/r03p; mkdir -p $R
for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$R/pmc_s/$(echo $pass | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $pass -f csv -d $d -o s -- python3 scripts/bench_step_launches.py > /dev/null 2>&1
done
python3 scripts/pmc_collect.py csr_agg $R/agg_pmc_step_launches.json "$R/pmc_s/**/*counter_collection.csv" > /dev/null
rm -rf $R/pmc_s
python3 scripts/bench_step_launches.py > $R/step_launches.json 2>/dev/null
python - <<'PY'
import json
s = json.load(open("gpurun_out/r03p/agg_pmc_step_launches.json"))
for k, v in s.items():
    print(k[:62], v.get("launches"), round(v.get("kernel_us_under_pmc", 0), 1), round(v.get("l2_hit_rate", 0), 3), v.get("fetch_bytes_corrected"), v.get("write_bytes"))
d = json.load(open("gpurun_out/r03p/step_launches.json"))
for k, v in d.items():
    if isinstance(v, dict):
        print(k, v.get("kernel_ms"), v.get("frac"), v.get("kernel_form"))
PY
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress > $R/bench_quick_$i.json 2>/dev/null; python -c "
import json; r=json.loads(open('$R/bench_quick_$i.json').read().strip().splitlines()[-1]); print(r['ms_per_step'], r['value'], r['roofline_workload_agg']['kernel_ms'], r['roofline_workload_agg'].get('kernel_form'))"; done
python bench.py > $R/bench_collab.json 2> $R/bench_collab.err; tail -c 200 $R/bench_collab.json
