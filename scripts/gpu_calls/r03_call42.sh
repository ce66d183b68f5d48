#!/bin/bash
# round 3, call 42: PMC of the step's launches with the tuned form pinned
R=gpurun_out/r03p; mkdir -p $R
for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$R/pmc_s/$(echo $pass | tr ' ' '_')
  PLNLP_AGG_FORM=65664 rocprofv3 --kernel-trace --pmc $pass -f csv -d $d -o s -- python3 scripts/bench_step_launches.py > /dev/null 2>&1
done
python3 scripts/pmc_collect.py csr_agg $R/agg_pmc_step_launches.json "$R/pmc_s/**/*counter_collection.csv" > /dev/null
rm -rf $R/pmc_s
python - <<'PY'
import json
s = json.load(open("gpurun_out/r03p/agg_pmc_step_launches.json"))
for k, v in s.items():
    print(k[:62], v.get("launches"), round(v.get("kernel_us_under_pmc", 0), 1), round(v.get("l2_hit_rate", 0), 3), v.get("fetch_bytes_corrected"), v.get("write_bytes"))
PY
