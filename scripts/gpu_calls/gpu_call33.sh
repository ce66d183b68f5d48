#!/bin/bash
mkdir -p gpurun_out/c33
{
for r in 1 2 3; do
echo "== bench collab: streaming hints in Adam"; python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
echo "== bench collab: plain accesses in Adam"; PLNLP_HIP_LIB=$PWD/plnlp_amd/build/abl/libplnlp_hip_adamplain.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done
} > gpurun_out/c33/adam_nt.txt 2>&1
cat gpurun_out/c33/adam_nt.txt
timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -k "adam or optim or trajectory" 2>&1 | tail -3
