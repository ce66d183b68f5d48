#!/bin/bash
# round 3, call 28: PMC (L2 hit rate, fabric bytes) of the hub chunk pass per form
O=gpurun_out/r03c28; mkdir -p $O
for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$O/pmc/$(echo $pass | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $pass -f csv -d $d -o a -- python3 scripts/bench_agg.py --cases collab --feat 256 --tune 0,128 --hub-order none,65536:256,32768:256,16384:128 > /dev/null 2>&1
done
python3 scripts/pmc_collect.py csr_agg $O/agg_pmc_hub_forms.json "$O/pmc/**/*counter_collection.csv" > /dev/null
rm -rf $O/pmc
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03c28/agg_pmc_hub_forms.json"))
for k, v in d.items():
    print(k[:70], {a: (round(v[a], 3) if isinstance(v[a], float) else v[a]) for a in ("launches", "kernel_us_under_pmc", "l2_hit_rate", "fetch_bytes_corrected", "write_bytes") if a in v})
PY
