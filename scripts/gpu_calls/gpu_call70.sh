#!/bin/bash
# HBM traffic of the aggregation kernel in its current forms (separate --pmc passes; FETCH_SIZE doubled per MI355X_MICROARCH.md)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_a; mkdir -p gpurun_out/c70
for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  d=gpurun_out/pmc_a/$(echo $pass | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $pass -f csv -d $d -o a -- python3 scripts/bench_agg.py --cases collab,uniform_big --feat 256,512 --tune 0,16 > /dev/null 2>&1
done
python3 scripts/pmc_collect.py csr_agg gpurun_out/c70/pmc_agg.json "gpurun_out/pmc_a/**/*counter_collection.csv" > /dev/null
rm -rf gpurun_out/pmc_a
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/c70/pmc_agg.json"))
for k,v in d.items():
    print(k[:80], {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ("fetch_bytes_corrected","write_bytes","l2_hit_rate","kernel_us_under_pmc","launches")})
PY
