#!/bin/bash
mkdir -p gpurun_out/c36
timeout 900 python -m pytest tests/test_hip_round2.py tests/test_hip_parity.py -q -m gpu -k "embedding_adam or row_sparse_backward or adam or optim or trajectory or smoke" 2>&1 | tail -15 > gpurun_out/c36/tests.log
cat gpurun_out/c36/tests.log
{
for r in 1 2 3; do
for e in 1 0; do
echo "== collab PLNLP_FUSE_EMBEDDING_ADAM=$e"; PLNLP_FUSE_EMBEDDING_ADAM=$e python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done; done
} > gpurun_out/c36/fuse_adam.txt 2>&1
cat gpurun_out/c36/fuse_adam.txt
