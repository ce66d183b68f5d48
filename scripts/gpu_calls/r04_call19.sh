#!/bin/bash
# round 4, call 19: the deferred [emb | x] copy: GCN tests, citation2 bench
O=$GRAFT_REPO_ROOT/gpurun_out/r04c19; mkdir -p $O
timeout 1500 python -m pytest tests -q -x -m gpu -k "gcn or GCN or citation2 or input or sparse_vs_dense or stale or full_size_steps" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -8 $O/tests.log
for i in 1 2; do
timeout 900 python bench.py --workload citation2 --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_citation2_$i.json 2> $O/bench_citation2_$i.err
python -c "
import json; r = json.loads(open('$O/bench_citation2_$i.json').read().strip().splitlines()[-1]); print('citation2', r['ms_per_step'], 'ms', r['value'] / 1e6, 'M edges/s', r.get('train_epoch', {}).get('value'))
"
done
