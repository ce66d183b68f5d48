#!/bin/bash
# round 4, call 17: the cost of the side stream's batch preparation: alone, absent, and overlapped
O=$GRAFT_REPO_ROOT/gpurun_out/r04c17; mkdir -p $O
for w in collab citation2 ddi; do
  timeout 600 python scripts/prologue_cost.py $w 40 > $O/cost_$w.json 2> $O/cost_$w.err; tail -2 $O/cost_$w.err; cat $O/cost_$w.json
done
