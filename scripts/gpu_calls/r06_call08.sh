#!/bin/bash
# round 6, call 8: (a) remaining new tests; (b) one locality experiment on the collab step's two aggregation launches (degree relabelling);
# (c) ddi's aggregation as a dense product on the existing kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round6.py -q -m gpu -k "asked_for or padded_table or block" 2>&1 | tail -4 > $O/call08_tests.txt; cat $O/call08_tests.txt
for lab in none degree; do
  export PLNLP_PROBE_RELABEL=$lab
  for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    d=$O/pmc_l/$lab/$(echo $pass | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $pass -f csv -d $d -o s -- python3 scripts/bench_step_launches.py > /dev/null 2>&1
  done
  python3 scripts/pmc_collect.py csr_agg $O/call08_agg_pmc_relabel_$lab.json "$O/pmc_l/$lab/**/*counter_collection.csv" > /dev/null
  for rep in 1 2 3; do python3 scripts/bench_step_launches.py 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lab rep$rep', {k: round(v['kernel_ms'], 4) for k, v in d.items() if isinstance(v, dict) and 'kernel_ms' in v})"; done
done > $O/call08_relabel_times.txt 2>&1
unset PLNLP_PROBE_RELABEL
rm -rf $O/pmc_l
cat $O/call08_relabel_times.txt
python3 - <<PY
import json
for lab in ("none", "degree"):
    d = json.load(open("$O/call08_agg_pmc_relabel_%s.json" % lab))
    for k, v in d.items():
        print(lab, k[:80], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items() if a in ('fetch_bytes_corrected', 'write_bytes', 'l2_hit_rate', 'kernel_us_under_pmc', 'launches')})
PY
timeout 600 python3 scripts/probe_ddi_dense_agg.py 2>&1 | grep -v amdgpu.ids > $O/call08_ddi_dense.jsonl; cat $O/call08_ddi_dense.jsonl
