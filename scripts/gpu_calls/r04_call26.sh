#!/bin/bash
# round 4, call 26: the narrow weighted aggregation form (<1, 16>, citation2's F = 52 launches) at 4 / 5 / 6 waves per SIMD:
# ./ab_mid (97 VGPRs, 4 waves), this tree (no epilogue prefetch in the narrow forms: 84 VGPRs, 5 waves), ./ab_v1 (the same +
# __launch_bounds__(256, 6): 80 VGPRs, 2 spilled); then the citation2 / collab bench of old, this tree, v1
O=$GRAFT_REPO_ROOT/gpurun_out/r04c26; mkdir -p $O
for v in mid new v1; do
  d=$GRAFT_REPO_ROOT; [ $v = mid ] && d=$GRAFT_REPO_ROOT/ab_mid; [ $v = v1 ] && d=$GRAFT_REPO_ROOT/ab_v1
  ( cd $d && timeout 600 python scripts/bench_agg.py --cases citation2 --feat 52,64 --weighted --tune 0 > $O/agg_$v.jsonl 2> $O/agg_$v.err )
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
for rep in 1 2; do
  for w in citation2 collab; do
    run old$rep $GRAFT_REPO_ROOT/ab_old $w
    run new$rep $GRAFT_REPO_ROOT $w
    run v1_$rep $GRAFT_REPO_ROOT/ab_v1 $w
  done
done
