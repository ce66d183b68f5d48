#!/bin/bash
# round 4, call 25: same box: the round-4 measurement commit (./ab_old), the tree before the lists-only commit (./ab_mid) and this
# tree (wave-aggregated list appends written by hand): bench A/B, and per-kernel breakdowns of citation2 and collab
O=$GRAFT_REPO_ROOT/gpurun_out/r04c25; mkdir -p $O
run() {  # name dir workload
  ( cd $2 && timeout 600 python bench.py --workload $3 --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/$1_$3.json 2> $O/$1_$3.err )
  python -c "
import json; r = json.loads(open('$O/$1_$3.json').read().strip().splitlines()[-1]); print('$1', '$3', round(r['ms_per_step'], 4), 'ms', 'epoch', r.get('train_epoch', {}).get('ms_per_step'))
"
}
for rep in 1 2; do
  for w in collab citation2; do
    run old$rep $GRAFT_REPO_ROOT/ab_old $w
    run mid$rep $GRAFT_REPO_ROOT/ab_mid $w
    run new$rep $GRAFT_REPO_ROOT $w
  done
done
cd /tmp && export TMPDIR=/tmp
for w in citation2 collab; do
for v in old new; do
  d=$GRAFT_REPO_ROOT; [ $v = old ] && d=$GRAFT_REPO_ROOT/ab_old
  cd $d
  rocprofv3 --kernel-trace --stats -f csv -d $O/prof_$v -o step -- python3 bench.py --workload $w --steps 12 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
  f=$(find $O/prof_$v -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/scripts/step_profile.py $f 6 45 sequence > $O/step_breakdown_${w}_$v.txt
  rm -rf $O/prof_$v
done
done
