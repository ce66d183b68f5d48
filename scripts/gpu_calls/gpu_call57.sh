#!/bin/bash
for th in 256 512 1024 4096; do
echo "== PLNLP_SPLIT_THRESHOLD=$th"
PLNLP_SPLIT_THRESHOLD=$th python bench.py --workload citation2 --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  citation2 step', round(d['ms_per_step'],4))"
PLNLP_SPLIT_THRESHOLD=$th python scripts/bench_agg.py --cases citation2 --feat 200 --weighted --tune 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(' ', d['case'], d['feat'], d['ms'])"
done
