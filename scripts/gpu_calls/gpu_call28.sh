#!/bin/bash
# step breakdowns with the split-bf16 GEMMs (collab, ddi, citation2)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/c28; mkdir -p $R
for w in collab ddi citation2; do
  rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o p -- python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-stress --no-roofline > $R/bench_${w}_under_rocprof.json 2>/dev/null
  f=$(find $R/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 5 45 > $R/step_breakdown_$w.txt
  f=$(find $R/prof -name "*kernel_stats.csv" | head -1); cp $f $R/kernel_stats_$w.csv
  rm -rf $R/prof
  head -32 $R/step_breakdown_$w.txt | cut -c1-150
done
