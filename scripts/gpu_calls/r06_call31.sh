#!/bin/bash
# round 6, call 31: the dense aggregation's K-slice count against whole rounds of workgroups; tests; the ddi step x 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
python scripts/probe_dense_agg_slices.py 2>/dev/null | tee $O/call31_slices.txt
timeout 900 python -m pytest tests/test_hip_round6.py tests/test_hip_round5.py -q -m gpu -x -k "dense or ddi" 2>&1 | tail -4
for rep in 1 2 3; do
  python bench.py --workload ddi --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('ddi rep$rep', round(r['ms_per_step'], 4))"
done | tee $O/call31_steps.txt
