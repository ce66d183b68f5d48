#!/bin/bash
for th in 256 1024; do
echo "== PLNLP_SPLIT_THRESHOLD=$th"
PLNLP_SPLIT_THRESHOLD=$th python bench.py --workload rmat --scale 0.25 --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  rmat fwd', round(d['ms_per_step'],3), 'agg ms', round(d['roofline']['kernel_ms'],2))"
PLNLP_SPLIT_THRESHOLD=$th python bench.py --workload citation2 --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  citation2 step', round(d['ms_per_step'],4))"
done
