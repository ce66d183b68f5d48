#!/bin/bash
# round 6, call 48: ddi's encoder products (4 267 rows) on the stationary-weights kernel: row threshold 16 384 (default) | 4 096, same box x 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for rep in 1 2 3; do
  for rows in 16384 4096; do
  python -c "
import sys, runpy
from plnlp_amd import _lib
_lib.load().plnlp_gemm_stationary_tuning(0, $rows)
sys.argv = 'bench.py --workload ddi --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline'.split()
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('ddi stationary_min_rows=$rows rep$rep', round(r['ms_per_step'], 4))"
  done
done | tee $O/call48_steps.txt
