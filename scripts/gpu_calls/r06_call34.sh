#!/bin/bash
# round 6, call 34: the panel kernel's wide-tile write-back with the gate rows of the next batch requested before this batch's
# stores: tests (gate / accumulate epilogues, step forms, citation2 parity), citation2 step x 2, the gated launch in the trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_round6.py tests/test_hip_round4.py tests/test_hip_round5.py tests/test_hip_parity.py -q -m gpu -x -k "block_kernel or gate or citation or gcn or stationary or head or epilogue" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
for rep in 1 2; do
  python bench.py --workload citation2 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('citation2 rep$rep', round(r['ms_per_step'], 4))"
done | tee $O/call34_steps.txt
rocprofv3 --kernel-trace --stats -f csv -d $O/prof34 -o step -- python3 bench.py --workload citation2 --steps 12 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
f=$(find $O/prof34 -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 6 45 > $O/call34_step_breakdown_citation2.txt
rm -rf $O/prof34
grep -n "gemm_x3\|steady" $O/call34_step_breakdown_citation2.txt | cut -c1-150
