#!/bin/bash
timeout 600 python -m pytest tests/test_hip_round2.py -q -m gpu -k "rccl" 2>&1 | tail -3
python bench.py --force-dist --steps 30 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('collab shard 1-rank', d['ms_per_step'])"
python bench.py --force-dist --dp-exchange scores --steps 30 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('collab scores 1-rank', d['ms_per_step'])"
