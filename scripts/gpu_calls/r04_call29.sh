#!/bin/bash
# round 4, call 29: this tree with the sort-free edge lists (default) and with the sort-based entry points (PLNLP_EDGE_LISTS=0),
# the measurement-commit tree (./ab_old) beside them: is the remaining +1.5 % of the collab step the preparation's?
O=$GRAFT_REPO_ROOT/gpurun_out/r04c29; mkdir -p $O
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
    run old1 $GRAFT_REPO_ROOT/ab_old $w A=1
    run fused1 $GRAFT_REPO_ROOT $w A=1
    run sort1 $GRAFT_REPO_ROOT $w PLNLP_EDGE_LISTS=0
done
