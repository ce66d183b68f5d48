#!/bin/bash
# round 5, call 10: (a) bench.py --share-gpu after the sub-group broadcast fix; (b) same-box A/B of the collab step: the side
# stream's priority (high = -1, the default since round 2; normal 0; low 1), the sort-free edge lists, the bias column sums
# on the second side stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c10; mkdir -p $O
for cfg in "collab 2" "collab 4" "ddi 2" "citation2 2"; do
  set -- $cfg
  timeout 900 python bench.py --workload $1 --gpus $2 --share-gpu --steps 8 --warmup 3 --no-strong > $O/bench_$1_w$2.json 2> $O/bench_$1_w$2.err
  python -c "
import json
try:
    r = json.loads(open('$O/bench_$1_w$2.json').read().strip().splitlines()[-1])
    print('$1 world $2 shared-gpu:', round(r['ms_per_step'], 3), 'ms/step, value', round(r['value']), '| choice', (r.get('dp_prediction') or {}).get('choice'), '| phases', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in (r.get('dp_phases') or {}).items() if k != 'note'})
except Exception as e:
    print('$1 world $2 FAILED', e); print(open('$O/bench_$1_w$2.err').read()[-1500:])
"
done
run() {  # name prio lists colsum
  PRIO=$2 LISTS=$3 CSUM=$4 timeout 600 python - <<PY > $O/ab_$1.json 2> $O/ab_$1.err
import os, sys, runpy
sys.argv = ["bench.py", "--steps", "40", "--warmup", "10", "--no-cpu-baseline", "--no-parity", "--no-stress", "--no-roofline"]
import plnlp_amd
from plnlp_amd import ops
ops.SIDE_STREAM_PRIORITY["value"] = int(os.environ["PRIO"])
ops.EDGE_LISTS_FUSED["enabled"] = os.environ["LISTS"] == "1"
ops.COLSUM_SIDE_STREAM["enabled"] = os.environ["CSUM"] == "1"
runpy.run_path("bench.py", run_name="__main__")
PY
  python -c "
import json; r = json.loads(open('$O/ab_$1.json').read().strip().splitlines()[-1]); print('$1 prio=$2 lists=$3 colsum_side=$4:', round(r['ms_per_step'], 4), 'ms', 'epoch', round(r.get('train_epoch', {}).get('ms_per_step', 0), 4))"
}
for rep in 1 2 3; do
  run base$rep -1 0 0
  run csum$rep -1 0 1
  run low$rep 1 0 1
  run norm$rep 0 0 1
  run lists$rep -1 1 1
  run listslow$rep 1 1 1
done
