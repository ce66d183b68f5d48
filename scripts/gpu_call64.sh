#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/c64; mkdir -p $R
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_round2.py -q -m gpu -k "mlp or predictor or hadamard or edge or outer or matvec or trajectory or single_step" 2>&1 | grep -E "passed|failed|^FAILED"
python bench.py --workload ddi --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ddi', round(d['ms_per_step'],4))"
rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o p -- python3 bench.py --workload ddi --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
f=$(find $R/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 5 40 > $R/step_breakdown_ddi.txt
rm -rf $R/prof
grep -E "hadamard|outer_vec|matvec|steady" $R/step_breakdown_ddi.txt | cut -c1-110
