#!/bin/bash
# round 3, call 44 (probe): cost of a device-scope release fence + ticket per chunk wave
for p in 0 1 0 1; do
echo "probe=$p"
PLNLP_PROBE_DONE=$p python scripts/bench_agg.py --cases collab --feat 256 --tune 0,128 --hub-order none,65536:256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('  tune', r['tune'], 'hub', r['hub_order'], 'ms', r['ms'])"
done
for p in 0 1 0 1; do
PLNLP_PROBE_DONE=$p python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step probe=$p', r['ms_per_step'])"
done
