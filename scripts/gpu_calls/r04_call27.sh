#!/bin/bash
# round 4, call 27: the narrow weighted form with 8 (6 waves) vs 4 (7 waves) neighbour rows in flight per wave
O=$GRAFT_REPO_ROOT/gpurun_out/r04c27; mkdir -p $O
timeout 600 python scripts/bench_agg.py --cases citation2 --feat 52,64 --weighted --tune 0,8 > $O/agg.jsonl 2> $O/agg.err
timeout 600 python scripts/bench_agg.py --cases citation2,collab --feat 52,64 --tune 0,8 >> $O/agg.jsonl 2>> $O/agg.err
python -c "
import json
for l in open('$O/agg.jsonl'):
    r = json.loads(l); print(r['case'], r['feat'], 'tune', r['tune'], r['ms'], 'ms', r['max_dev_vs_first'])
"
