#!/bin/bash
# round 3, call 46: long-row threshold again, now that the chunk pass shares the main pass's launch
for rep in 1 2; do for t in 128 192 256 384 512; do
PLNLP_SPLIT_THRESHOLD=$t python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('collab threshold=$t', r['ms_per_step'], r['train_epoch']['value'])"
done; done
for t in 128 256 512; do
PLNLP_SPLIT_THRESHOLD=$t python bench.py --workload ddi --steps 10 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ddi threshold=$t', r['ms_per_step'])"
done
