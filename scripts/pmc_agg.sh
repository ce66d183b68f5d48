#!/bin/bash
# cache behaviour of the aggregation kernel: L2 hit rate and fabric-side traffic (separate --pmc passes),
# on the cache-resident collab graph and on the graph that does not fit (uniform_big)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_a
for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  d=gpurun_out/pmc_a/$(echo $pass | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $pass -f csv -d $d -o a -- python3 scripts/bench_agg.py --cases collab,uniform_big --feat 256,512 --tune 0 > /dev/null 2>&1
done
python3 scripts/pmc_collect.py csr_agg gpurun_out/r02/pmc_agg.json "gpurun_out/pmc_a/**/*counter_collection.csv"
rm -rf gpurun_out/pmc_a
