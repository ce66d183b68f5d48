#!/bin/bash
# round 6, call 51: the north-star kernel's fabric-side traffic on the round-6 tree (separate --pmc passes: TCC hit / miss, FETCH_SIZE,
# WRITE_SIZE) -- uniform 2.9 M-node graph, F = 512, the pinned form (128-column slabs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O; rm -rf $O/pmc51
for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$O/pmc51/$(echo $pass | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $pass -f csv -d $d -o a -- python3 scripts/bench_agg.py --cases uniform_big --feat 512 --tune 16 > $O/call51_bench_agg.txt 2>/dev/null
done
python3 scripts/pmc_collect.py csr_agg $O/call51_northstar_pmc.json "$O/pmc51/**/*counter_collection.csv" > /dev/null
rm -rf $O/pmc51
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r06/call51_northstar_pmc.json'))
for k, v in d.items(): print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
PY
tail -3 $O/call51_bench_agg.txt | cut -c1-300
