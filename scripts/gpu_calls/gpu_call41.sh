#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/c41; mkdir -p $R
rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o p -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-stress --no-roofline > $R/bench_collab_under_rocprof.json 2>/dev/null
f=$(find $R/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 10 45 sequence > $R/step_breakdown_collab.txt
rm -rf $R/prof
grep -A200 "launch sequence" $R/step_breakdown_collab.txt | grep " s  0 " | cut -c1-120
