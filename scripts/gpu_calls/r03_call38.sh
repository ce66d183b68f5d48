#!/bin/bash
O=gpurun_out/r03c38; mkdir -p $O
rocprofv3 --kernel-trace -f csv -d $O/prof -o step -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench.json 2>$O/bench.err
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); head -2 $f | cut -c1-400; python scripts/overlap_profile.py $f 3 > $O/overlap.txt; cat $O/overlap.txt | cut -c1-260
rm -rf $O/prof
