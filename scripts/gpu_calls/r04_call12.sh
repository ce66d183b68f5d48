#!/bin/bash
# round 4, call 12: the fused edge-lists builder: equality tests, the collab step's breakdown and bench with / without it
O=$GRAFT_REPO_ROOT/gpurun_out/r04c12; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_round4.py -q -k "edge_lists" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -12 $O/tests.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o step -- python3 bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 6 45 sequence > $O/step_breakdown_collab.txt
rm -rf $O/prof
head -42 $O/step_breakdown_collab.txt
for w in collab ddi citation2; do
  timeout 900 python bench.py --workload $w --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_$w.json 2> $O/bench_$w.err
  python -c "
import json; r = json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', r['ms_per_step'], 'ms', r['value'] / 1e6, 'M edges/s', r.get('train_epoch', {}).get('value'))
"
done
