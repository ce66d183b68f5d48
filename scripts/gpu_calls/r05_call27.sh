#!/bin/bash
# collab / ddi step breakdowns with the wide weight gradient
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r05q; mkdir -p $R
for w in collab ddi; do
  rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o step -- python3 bench.py --workload $w --steps 12 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
  f=$(find $R/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 6 45 sequence > $R/step_breakdown_$w.txt
  rm -rf $R/prof
  head -8 $R/step_breakdown_$w.txt
done
sed -n '/launch sequence/,$p' $R/step_breakdown_collab.txt | head -24
