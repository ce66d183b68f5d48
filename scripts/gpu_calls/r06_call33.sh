#!/bin/bash
# round 6, call 33: the ddi line and breakdown on the final tree (dense aggregation's slice rule), the whole GPU suite + smoke again
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; R=gpurun_out/r06p; mkdir -p $O $R
python bench.py --workload ddi --steps 10 --warmup 5 --no-parity --no-stress --cpu-steps 1 > $R/bench_ddi.json 2>/dev/null
rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o step -- python3 bench.py --workload ddi --steps 12 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
f=$(find $R/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 6 45 sequence > $R/step_breakdown_ddi.txt
rm -rf $R/prof
head -12 $R/step_breakdown_ddi.txt | cut -c1-130
SECONDS=0
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 | tee $O/call33_suite.txt
echo "suite: ${SECONDS}s" | tee -a $O/call33_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $O/call33_suite.txt
