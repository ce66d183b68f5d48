#!/bin/bash
# round 5, call 5: the new round-5 GPU tests that need no fixture, the multirank file, and FRESH counters for the collab
# step's own two aggregation launches (r04's line quoted round 3's): separate --pmc passes over scripts/bench_step_launches.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c05; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round5.py -q -m gpu -x -s -k "not trained_regime_parity" > $O/round5.txt 2>&1; grep -v amdgpu.ids $O/round5.txt | tail -25 | cut -c1-400
timeout 900 python -m pytest tests/test_hip_multirank.py -q -m gpu > $O/multirank.txt 2>&1; tail -4 $O/multirank.txt | cut -c1-300
for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$O/pmc_s/$(echo $pass | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $pass -f csv -d $d -o s -- python3 scripts/bench_step_launches.py > /dev/null 2>&1
done
python3 scripts/pmc_collect.py csr_agg $O/agg_pmc_step_launches.json "$O/pmc_s/**/*counter_collection.csv" > /dev/null
rm -rf $O/pmc_s
python3 scripts/bench_step_launches.py > $O/step_launches.json 2>/dev/null
python3 -c "
import json
d = json.load(open('$O/agg_pmc_step_launches.json'))
for k, v in d.items():
    print(k[:90], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items() if a in ('fetch_bytes_corrected', 'write_bytes', 'l2_hit_rate', 'kernel_us_under_pmc', 'launches')})
"
