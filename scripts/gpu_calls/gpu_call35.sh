#!/bin/bash
mkdir -p gpurun_out/c35
{
for r in 1 2; do
for e in 1 0; do
echo "== collab PLNLP_OVERLAP_BACKWARD=$e"; PLNLP_OVERLAP_BACKWARD=$e python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done; done
for w in ddi citation2; do
for e in 1 0; do
echo "== $w PLNLP_OVERLAP_BACKWARD=$e"; PLNLP_OVERLAP_BACKWARD=$e python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done; done
} > gpurun_out/c35/overlap.txt 2>&1
cat gpurun_out/c35/overlap.txt
