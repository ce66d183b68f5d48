#!/bin/bash
# round 4, call 24: bisecting the collab regression of commit 58c7781 on one box: mid (aeec4ea), HEAD, v1 (HEAD csrc + the ops.py of aeec4ea; earlier: HEAD with
# segments_kernel's old statement order), v2 (HEAD ops.py + the edge_lists.hip of aeec4ea; earlier: HEAD with the old once-per-process LDS attribute code)
O=$GRAFT_REPO_ROOT/gpurun_out/r04c24; mkdir -p $O
run() {  # name dir workload
  ( cd $2 && timeout 600 python bench.py --workload $3 --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/$1_$3.json 2> $O/$1_$3.err )
  python -c "
import json; r = json.loads(open('$O/$1_$3.json').read().strip().splitlines()[-1]); print('$1', '$3', round(r['ms_per_step'], 4), 'ms', 'epoch', r.get('train_epoch', {}).get('ms_per_step'))
"
}
for rep in 1 2; do
    run mid$rep $GRAFT_REPO_ROOT/ab_mid collab
    run new$rep $GRAFT_REPO_ROOT collab
    run v1_$rep $GRAFT_REPO_ROOT/ab_v1 collab
    run v2_$rep $GRAFT_REPO_ROOT/ab_v2 collab
done
