#!/bin/bash
# round 4, call 22: the sort-free lists for the non-compact backward too (ddi): whole GPU suite, ddi breakdown + bench
O=$GRAFT_REPO_ROOT/gpurun_out/r04c22; mkdir -p $O
timeout 2400 python -m pytest tests -q -x -m gpu > $O/tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/tests.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o step -- python3 bench.py --workload ddi --steps 12 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 6 45 sequence > $O/step_breakdown_ddi.txt
rm -rf $O/prof
head -3 $O/step_breakdown_ddi.txt; grep -n "el::\|rocprim\|fillBuffer" $O/step_breakdown_ddi.txt | head
for i in 1 2; do
timeout 900 python bench.py --workload ddi --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_ddi_$i.json 2> $O/bench_ddi_$i.err
python -c "
import json; r = json.loads(open('$O/bench_ddi_$i.json').read().strip().splitlines()[-1]); print('ddi', r['ms_per_step'], 'ms', r['value'] / 1e6, 'M edges/s', r.get('train_epoch', {}).get('value'))
"
done
