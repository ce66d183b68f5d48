#!/bin/bash
# round 3, call 37: launch sequence of the collab step after the launch fusions
O=gpurun_out/r03c37; mkdir -p $O
rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o step -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench.json 2>$O/bench.err
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 10 45 sequence > $O/step_breakdown_collab.txt
rm -rf $O/prof
sed -n 1,3p $O/step_breakdown_collab.txt; grep -n "launch sequence" -A70 $O/step_breakdown_collab.txt | cut -c1-150
tail -n 3 $O/bench.err
