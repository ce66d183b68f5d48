#!/bin/bash
# round 6, call 25: the segment backward with long segments shared by a workgroup: tests, the kernel alone (uniform / hubs / ddi),
# same-box A/B of the citation2 and ddi steps (PLNLP_EDGE_SEGMENT=wave | auto), the kernel's time in each step's trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round6.py -q -m gpu -x -k "segment_backward" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_round2.py tests/test_hip_round4.py -q -m gpu -x -k "edge or scorer or hadamard or mlp or predictor" 2>&1 | tail -4
python scripts/probe_segment_bwd.py | tee $O/call25_times.txt
for rep in 1 2; do
  for form in wave auto; do
    for w in citation2 ddi; do
  PLNLP_EDGE_SEGMENT=$form python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$w $form rep$rep', round(r['ms_per_step'], 4))"
    done
  done
done | tee $O/call25_steps.txt
for w in citation2 ddi; do
  rocprofv3 --kernel-trace --stats -f csv -d $O/prof25 -o step -- python3 bench.py --workload $w --steps 12 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
  f=$(find $O/prof25 -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 6 45 > $O/call25_step_breakdown_$w.txt
  rm -rf $O/prof25
  grep -n "edge_segment\|steady" $O/call25_step_breakdown_$w.txt | cut -c1-150
done
