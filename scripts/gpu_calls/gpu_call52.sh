#!/bin/bash
for r in 1 2; do
for cfg in "20 3" "40 5" "20 10" "100 5"; do set -- $cfg
python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps $1 warmup $2:', round(d['ms_per_step'],4))"
done; done
