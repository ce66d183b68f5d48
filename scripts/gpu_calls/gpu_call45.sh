#!/bin/bash
for d in 2 1 3 0; do echo "== PLNLP_STEP_THROTTLE=$d"; PLNLP_STEP_THROTTLE=$d python scripts/probe_gap.py 2>&1 | grep -E "ms/step|num_device_alloc"; done
for w in collab ddi citation2; do
python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['ms_per_step'], 'host enqueue ms/step', d['host_enqueue_ms_per_step'])"
done
