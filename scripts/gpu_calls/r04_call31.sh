#!/bin/bash
# round 4, call 31: two more kernels moved off their occupancy cliffs (same sums, same order): the segmented edge-gradient gather with
# one float4 slot per lane for F <= 256 (85 -> 63 VGPRs, 5 -> 7 waves), the weighted narrow chunk pass with 4 groups in flight
# (83 -> 5x VGPRs): tests of those kernels, then citation2 / ddi / collab against the measurement-commit tree (./ab_old)
O=$GRAFT_REPO_ROOT/gpurun_out/r04c31; mkdir -p $O
timeout 1500 python -m pytest tests -q -x -m gpu -k "edge or segment or hadamard or mlp or MLP or aggregat or gcn or GCN or citation2 or full_size" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -3 $O/tests.log
run() {  # name dir workload
  ( cd $2 && timeout 600 python bench.py --workload $3 --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/$1_$3.json 2> $O/$1_$3.err )
  python -c "
import json; r = json.loads(open('$O/$1_$3.json').read().strip().splitlines()[-1]); print('$1', '$3', round(r['ms_per_step'], 4), 'ms', 'epoch', r.get('train_epoch', {}).get('ms_per_step'))
"
}
for rep in 1 2; do
  for w in citation2 ddi collab; do
    run old$rep $GRAFT_REPO_ROOT/ab_old $w
    run new$rep $GRAFT_REPO_ROOT $w
  done
done
