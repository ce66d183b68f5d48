#!/bin/bash
# round 6, call 21: the table's Adam step in citation2's transposed 64-wide aggregation (GCN input conv): bits, the citation2 parity
# tests, same-box A/B of the step (model.FUSE_EMBEDDING_ADAM set by a two-line driver around bench.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round6.py -q -m gpu -x -k "table_adam or padded_table" 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_hip_round2.py tests/test_hip_round4.py tests/test_hip_round5.py tests/test_hip_parity.py -q -m gpu -x -k "citation or gcn or adam or capture or graph" 2>&1 | tail -4
for rep in 1 2; do
  for fuse in 1 0; do
  python -c "
import sys, runpy
import plnlp_amd.model as M
M.FUSE_EMBEDDING_ADAM['enabled'] = bool($fuse)
sys.argv = 'bench.py --workload citation2 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline'.split()
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('citation2 fuse=$fuse rep$rep', round(r['ms_per_step'], 4), 'host_busy', round(r['host_busy_ms_per_step'], 3))"
  done
done | tee $O/call21_steps.txt
rocprofv3 --kernel-trace --stats -f csv -d $O/prof21 -o step -- python3 bench.py --workload citation2 --steps 12 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
f=$(find $O/prof21 -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 6 45 sequence > $O/call21_step_breakdown_citation2.txt
rm -rf $O/prof21
head -24 $O/call21_step_breakdown_citation2.txt | cut -c1-130
