#!/bin/bash
for r in 1 2; do for e in 1 0; do
echo "== collab PLNLP_FUSED_ADAM_SIDE_STREAM=$e"; PLNLP_FUSED_ADAM_SIDE_STREAM=$e python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done; done
PLNLP_FUSED_ADAM_SIDE_STREAM=1 timeout 600 python -m pytest tests/test_hip_round2.py -q -m gpu -k "embedding_adam" 2>&1 | grep -E "passed|failed"
