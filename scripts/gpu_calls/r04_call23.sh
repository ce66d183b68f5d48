#!/bin/bash
# round 4, call 23: same-box A/B of the bench: the tree of the round-4 measurement commit (ba426cf, ./ab_old), the tree before the
# lists-only commit (aeec4ea, ./ab_mid) and this tree, interleaved
O=$GRAFT_REPO_ROOT/gpurun_out/r04c23; mkdir -p $O
run() {  # name dir workload
  ( cd $2 && timeout 600 python bench.py --workload $3 --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/$1_$3.json 2> $O/$1_$3.err )
  python -c "
import json; r = json.loads(open('$O/$1_$3.json').read().strip().splitlines()[-1]); print('$1', '$3', round(r['ms_per_step'], 4), 'ms', 'epoch', r.get('train_epoch', {}).get('ms_per_step'))
"
}
for rep in 1 2 3; do
  for w in collab; do
    run old$rep $GRAFT_REPO_ROOT/ab_old $w
    run mid$rep $GRAFT_REPO_ROOT/ab_mid $w
    run new$rep $GRAFT_REPO_ROOT $w
  done
done
for w in citation2; do
  run old1 $GRAFT_REPO_ROOT/ab_old $w
  run mid1 $GRAFT_REPO_ROOT/ab_mid $w
  run new1 $GRAFT_REPO_ROOT $w
done
