#!/bin/bash
# round 6, call 19: (a) the default bench twice on a fresh box (cold / warm); (b) host time per step, this tree vs the round-5 switches;
# (c) the 1-column colsum test; ddi step after it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for rep in cold warm warm2; do
  python bench.py --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('collab default $rep', round(r['ms_per_step'], 4), 'host_busy', round(r['host_busy_ms_per_step'], 3), 'host_enqueue', round(r['host_enqueue_ms_per_step'], 3))"
done > $O/call19_cold_warm.txt 2>&1; cat $O/call19_cold_warm.txt
timeout 600 python -m pytest tests/test_hip_round6.py tests/test_hip_round5.py -q -m gpu -k "one_column or fused_head_backward" 2>&1 | tail -3
for rep in 1 2; do
  python bench.py --workload ddi --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('ddi rep$rep', round(r['ms_per_step'], 4), 'host_busy', round(r['host_busy_ms_per_step'], 3))"
done
python scripts/host_profile.py 2>/dev/null | tail -25
