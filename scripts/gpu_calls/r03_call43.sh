#!/bin/bash
# round 3, call 43: the batch's endpoint lists in one launch each: tests, step time, launches per step
O=gpurun_out/r03c43; mkdir -p $O
python -m pytest tests/test_hip_round3.py -x -q -m gpu -k "endpoint_lists" > $O/new.log 2>&1; echo "rc=$?" >> $O/new.log; tail -n 3 $O/new.log
python -m pytest tests -x -q -m gpu --deselect tests/test_hip_round3.py::test_trained_regime_hits_parity_over_seeds > $O/suite.log 2>&1; echo "rc=$?" >> $O/suite.log; grep -n "passed\|failed\|FAILED" $O/suite.log | tail -3
for i in 1 2 3; do python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_$i.json 2>/dev/null; python -c "
import json; r=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1]); print('collab', r['ms_per_step'], r['value'], 'host_busy', r['host_busy_ms_per_step'], 'epoch', r['train_epoch']['value'], 'capt', r['step_capture']['other_path']['ms_per_step'], r['step_capture']['other_path']['host_busy_ms_per_step'])"; done
rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o step -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 10 45 > $O/step_breakdown.txt; head -3 $O/step_breakdown.txt | cut -c1-120
rm -rf $O/prof
