#!/bin/bash
for th in 256 128 64 512; do
echo "== PLNLP_SPLIT_THRESHOLD=$th"
PLNLP_SPLIT_THRESHOLD=$th python scripts/bench_agg.py --cases collab,ddi --feat 256,512 --tune 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(' ', d['case'], d['feat'], d['ms'])"
PLNLP_SPLIT_THRESHOLD=$th python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  collab step', round(d['ms_per_step'],4))"
PLNLP_SPLIT_THRESHOLD=$th python bench.py --workload ddi --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ddi step', round(d['ms_per_step'],4))"
done
