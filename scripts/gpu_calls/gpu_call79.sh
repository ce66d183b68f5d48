#!/bin/bash
for r in 1 2; do python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('collab', round(d['ms_per_step'],4), d['host_enqueue_ms_per_step'])"; done
python bench.py --workload ddi --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ddi', round(d['ms_per_step'],4))"
