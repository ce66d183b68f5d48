#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/c67; mkdir -p $R
rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o p -- python3 bench.py --force-dist --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
f=$(find $R/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 5 45 > $R/step_breakdown_shard1.txt
rm -rf $R/prof
head -40 $R/step_breakdown_shard1.txt | cut -c1-140
