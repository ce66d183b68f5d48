#!/bin/bash
python bench.py --force-dist --dp-exchange scores --steps 30 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('collab scores 1-rank', d['ms_per_step'], d.get('replicas_in_sync'))"
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_round2.py -q -m gpu -k "data_parallel or scores or rccl or replica" 2>&1 | grep -E "passed|failed"
