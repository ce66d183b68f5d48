#!/bin/bash
# round 4, call 30: the hub bitmap pass as its own launch (only its workgroups ask for the bitmap's LDS): edge-lists tests,
# then fused vs sort-based (PLNLP_EDGE_LISTS=0) vs the measurement-commit tree
O=$GRAFT_REPO_ROOT/gpurun_out/r04c30; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round4.py -q -x -k "edge_lists" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -3 $O/tests.log
run() {  # name dir workload env
  ( cd $2 && env $4 timeout 600 python bench.py --workload $3 --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/$1_$3.json 2> $O/$1_$3.err )
  python -c "
import json; r = json.loads(open('$O/$1_$3.json').read().strip().splitlines()[-1]); print('$1', '$3', round(r['ms_per_step'], 4), 'ms', 'epoch', r.get('train_epoch', {}).get('ms_per_step'))
"
}
for rep in 1 2 3; do
  for w in collab; do
    run old$rep $GRAFT_REPO_ROOT/ab_old $w A=1
    run fused$rep $GRAFT_REPO_ROOT $w A=1
    run sort$rep $GRAFT_REPO_ROOT $w PLNLP_EDGE_LISTS=0
  done
done
for w in citation2 ddi; do
    run fused1 $GRAFT_REPO_ROOT $w A=1
    run sort1 $GRAFT_REPO_ROOT $w PLNLP_EDGE_LISTS=0
done
