#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c30
timeout 900 python -m pytest tests/test_hip_round2.py -q -m gpu -k "pair_gathered or fused_edge_mlp" 2>&1 | tail -12 > gpurun_out/c30/tests.log
cat gpurun_out/c30/tests.log
R=gpurun_out/c30
for f in 1 0; do
  PLNLP_FUSE_EDGE_MLP=$f rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o p -- python3 bench.py --workload ddi --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
  f2=$(find $R/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f2 5 45 > $R/step_breakdown_ddi_fuse$f.txt
  rm -rf $R/prof
  echo "== fuse=$f"; head -16 $R/step_breakdown_ddi_fuse$f.txt | cut -c1-140
done
