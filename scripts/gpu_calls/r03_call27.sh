#!/bin/bash
# round 3, call 27: chunk-pass slicing fix, step breakdown with / without the tuned hub form
O=gpurun_out/r03c27; mkdir -p $O
timeout 600 python scripts/debug_agg_forms.py 2>&1 | tail -12
for t in 0 1; do
PLNLP_AGG_AUTOTUNE=$t rocprofv3 --kernel-trace --stats -f csv -d $O/prof$t -o step -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_under_prof_$t.json 2>/dev/null
f=$(find $O/prof$t -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 10 45 > $O/step_breakdown_tune$t.txt
rm -rf $O/prof$t
done
head -40 $O/step_breakdown_tune0.txt; echo ======; head -40 $O/step_breakdown_tune1.txt
