#!/bin/bash
# round 4, call 28: the F = 256 aggregation forms at 7 waves per SIMD (this tree) vs 8 (./ab_v1: __launch_bounds__(256, 8),
# ~30 SGPRs kept in VGPR lanes): micro-benchmark and the collab / citation2 bench, old tree beside them
O=$GRAFT_REPO_ROOT/gpurun_out/r04c28; mkdir -p $O
for v in new v1; do
  d=$GRAFT_REPO_ROOT; [ $v = v1 ] && d=$GRAFT_REPO_ROOT/ab_v1
  ( cd $d && timeout 600 python scripts/bench_agg.py --cases collab,citation2 --feat 200,256 --tune 0 > $O/agg_$v.jsonl 2> $O/agg_$v.err; timeout 600 python scripts/bench_agg.py --cases collab,citation2 --feat 200,256 --weighted --tune 0 >> $O/agg_$v.jsonl 2>> $O/agg_$v.err )
  python -c "
import json
for l in open('$O/agg_$v.jsonl'):
    r = json.loads(l); print('$v', r['case'], r['feat'], r['ms'], 'ms')
"
done
run() {  # name dir workload
  ( cd $2 && timeout 600 python bench.py --workload $3 --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/$1_$3.json 2> $O/$1_$3.err )
  python -c "
import json; r = json.loads(open('$O/$1_$3.json').read().strip().splitlines()[-1]); print('$1', '$3', round(r['ms_per_step'], 4), 'ms', 'epoch', r.get('train_epoch', {}).get('ms_per_step'))
"
}
for rep in 1 2 3; do
  for w in collab citation2; do
    run old$rep $GRAFT_REPO_ROOT/ab_old $w
    run new$rep $GRAFT_REPO_ROOT $w
    run v1_$rep $GRAFT_REPO_ROOT/ab_v1 $w
  done
done
