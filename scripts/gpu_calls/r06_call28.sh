#!/bin/bash
# round 6, call 28: the Hadamard forward in XCD-pinned column slabs (ddi): test, ddi step x 3 with / without, its time in the trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round6.py -q -m gpu -x -k "hadamard_forward or segment_backward" 2>&1 | tail -4
for rep in 1 2 3; do
  for form in noslab auto; do
  PLNLP_EDGE_SEGMENT=$form python bench.py --workload ddi --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('ddi $form rep$rep', round(r['ms_per_step'], 4))"
  done
done | tee $O/call28_steps.txt
rocprofv3 --kernel-trace --stats -f csv -d $O/prof28 -o step -- python3 bench.py --workload ddi --steps 12 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
f=$(find $O/prof28 -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 6 45 > $O/call28_step_breakdown_ddi.txt
rm -rf $O/prof28
grep -n "edge_\|steady" $O/call28_step_breakdown_ddi.txt | cut -c1-150
