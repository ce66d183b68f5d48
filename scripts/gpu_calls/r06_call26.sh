#!/bin/bash
# round 6, call 26: groups of four waves / four segments against eight / eight (many segments): the kernel alone, citation2's step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
PROBE_FORMS=wave,auto,group4 PROBE_SHAPES=citation2,citation2_hubs python scripts/probe_segment_bwd.py | tee $O/call26_times.txt
for rep in 1 2; do
  for form in auto group4; do
  PLNLP_EDGE_SEGMENT=$form python bench.py --workload citation2 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('citation2 $form rep$rep', round(r['ms_per_step'], 4))"
  done
done | tee $O/call26_steps.txt
PLNLP_EDGE_SEGMENT=group4 rocprofv3 --kernel-trace --stats -f csv -d $O/prof26 -o step -- python3 bench.py --workload citation2 --steps 12 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
f=$(find $O/prof26 -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 6 45 | grep -n "edge_segment\|steady" | cut -c1-150
rm -rf $O/prof26
