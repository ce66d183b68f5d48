#!/bin/bash
# round 6, call 29: the whole GPU suite + smoke on the final tree; the ddi / citation2 lines and the three step breakdowns again
# (kernel names from the launch counters; the slab forms)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; R=gpurun_out/r06p; mkdir -p $O $R
SECONDS=0
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 | tee $O/call29_suite.txt
echo "suite: ${SECONDS}s" | tee -a $O/call29_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $O/call29_suite.txt
for w in ddi citation2; do
  python bench.py --workload $w --steps 10 --warmup 5 --no-parity --no-stress --cpu-steps 1 > $R/bench_$w.json 2>/dev/null
done
for w in collab ddi citation2; do
  rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o step -- python3 bench.py --workload $w --steps 12 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
  f=$(find $R/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 6 45 sequence > $R/step_breakdown_$w.txt
  rm -rf $R/prof
done
python bench.py --steps 20 --warmup 3 > $R/bench_collab_final.json 2>/dev/null; tail -c 400 $R/bench_collab_final.json
