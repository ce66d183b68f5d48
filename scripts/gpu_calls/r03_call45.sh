#!/bin/bash
# round 3, call 45: where the 1-rank sharded step's extra 0.86 ms goes
O=gpurun_out/r03c45; mkdir -p $O
rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o step -- python3 bench.py --force-dist --dp-exchange shard --steps 20 --warmup 6 --no-cpu-baseline --no-parity --no-stress --no-roofline --no-strong > $O/bench.json 2>$O/bench.err
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 8 60 sequence > $O/step_breakdown_shard.txt
rm -rf $O/prof
head -50 $O/step_breakdown_shard.txt | cut -c1-150
grep -n "launch sequence" -A90 $O/step_breakdown_shard.txt | cut -c1-140 | tail -75
