#!/bin/bash
# round 6, call 32: same-box A/B of the ddi step, dense aggregation with four K slices (the first rule) against the round-aware rule
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for rep in 1 2 3; do
  for s in 4 0; do
  python -c "
import sys, runpy
from plnlp_amd import _lib
_lib.load().plnlp_dense_aggregate_tuning($s)
sys.argv = 'bench.py --workload ddi --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline'.split()
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('ddi slices=$s rep$rep', round(r['ms_per_step'], 4))"
  done
done | tee $O/call32_steps.txt
