#!/bin/bash
# round 6, call 35: bias / row-dot weights of the next chunk requested before this chunk's stores (panel kernel, wide tiles):
# tests, ddi + collab steps, the forward launches in ddi's trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_round6.py tests/test_hip_round4.py tests/test_hip_round5.py tests/test_hip_parity.py -q -m gpu -x -k "block_kernel or gate or stationary or head or epilogue or rowdot or ddi" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
for rep in 1 2; do
  for w in ddi collab; do
  python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$w rep$rep', round(r['ms_per_step'], 4))"
  done
done | tee $O/call35_steps.txt
rocprofv3 --kernel-trace --stats -f csv -d $O/prof35 -o step -- python3 bench.py --workload ddi --steps 12 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
f=$(find $O/prof35 -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 6 45 > $O/call35_step_breakdown_ddi.txt
rm -rf $O/prof35
grep -n "gemm_x3\|steady" $O/call35_step_breakdown_ddi.txt | cut -c1-150
