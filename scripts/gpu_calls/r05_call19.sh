#!/bin/bash
# round 5, call 19: round over round on ONE box: the round-4 tree (6ae4e84, ./ab_r4) against this tree, interleaved -- the three
# workloads' steps (2 repetitions) and the north-star aggregation kernel (the `roofline` object of the collab line, WITH the roofline)
O=$GRAFT_REPO_ROOT/gpurun_out/r05c19; mkdir -p $O
run() {  # name dir workload extra
  ( cd $2 && timeout 900 python bench.py --workload $3 --steps 30 --warmup 8 --no-cpu-baseline --no-parity $4 > $O/$1_$3.json 2> $O/$1_$3.err )
  python -c "
import json; r = json.loads(open('$O/$1_$3.json').read().strip().splitlines()[-1]); ro = r.get('roofline') or {}
print('$1', '$3', round(r['ms_per_step'], 4), 'ms | epoch', round(r.get('train_epoch', {}).get('ms_per_step', 0), 4), '| roofline kernel', round(ro.get('kernel_ms') or 0, 3), 'ms', ro.get('frac'))
"
}
for rep in 1 2; do
  run r4_$rep $GRAFT_REPO_ROOT/ab_r4 collab ""
  run r5_$rep $GRAFT_REPO_ROOT collab ""
  for w in ddi citation2; do
    run r4_$rep $GRAFT_REPO_ROOT/ab_r4 $w "--no-stress --no-roofline"
    run r5_$rep $GRAFT_REPO_ROOT $w "--no-stress --no-roofline"
  done
done
