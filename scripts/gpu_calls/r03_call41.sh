#!/bin/bash
# round 3, call 41: data-gradient GEMM last among the backward GEMMs (its output is what the transposed aggregation gathers)
O=gpurun_out/r03c41; mkdir -p $O
for i in 1 2 3; do for t in 0 1; do
PLNLP_DGRAD_LAST=$t python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_collab_d${t}_$i.json 2>/dev/null
python -c "
import json; r=json.loads(open('$O/bench_collab_d${t}_$i.json').read().strip().splitlines()[-1]); print('collab dgrad_last=$t', r['ms_per_step'], r['value'], r['train_epoch']['value'])"
done; done
python -m pytest tests -x -q -m gpu -k "captured or fused or trajectory or row_restricted or sparse" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 3 $O/tests.log
PLNLP_DGRAD_LAST=1 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o step -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 10 45 > $O/step_breakdown.txt; head -12 $O/step_breakdown.txt | cut -c1-120
rm -rf $O/prof
