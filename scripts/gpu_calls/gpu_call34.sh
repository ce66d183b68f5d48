#!/bin/bash
mkdir -p gpurun_out/c34
timeout 900 python -m pytest tests/test_hip_round2.py -q -m gpu -k "early_embedding" 2>&1 | tail -12 > gpurun_out/c34/tests.log
cat gpurun_out/c34/tests.log
{
for r in 1 2 3; do
for e in 1 0; do
echo "== collab PLNLP_EARLY_EMBEDDING_STEP=$e"; PLNLP_EARLY_EMBEDDING_STEP=$e python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done; done
for e in 1 0; do
echo "== ddi PLNLP_EARLY_EMBEDDING_STEP=$e"; PLNLP_EARLY_EMBEDDING_STEP=$e python bench.py --workload ddi --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done
} > gpurun_out/c34/early.txt 2>&1
cat gpurun_out/c34/early.txt
