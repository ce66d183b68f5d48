#!/bin/bash
# round 6, call 49: the panel kernel's tail round in 32-column tiles: tests, the collab step x 3 with / without, the launches in the trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_round6.py tests/test_hip_round4.py tests/test_hip_round5.py -q -m gpu -x -k "tail_round or block_kernel or stationary or collab or pair" 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
for rep in 1 2 3; do
  for mode in notail auto; do
  PLNLP_GEMM_BLOCK=$mode python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('collab $mode rep$rep', round(r['ms_per_step'], 4))"
  done
done | tee $O/call49_steps.txt
rocprofv3 --kernel-trace --stats -f csv -d $O/prof49 -o step -- python3 bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
f=$(find $O/prof49 -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 6 45 sequence > $O/call49_step_breakdown_collab.txt
rm -rf $O/prof49
grep -n "gemm_x3s\|split_b\|steady" $O/call49_step_breakdown_collab.txt | cut -c1-150
