#!/bin/bash
# round 5, call 9: bench.py's N-rank path EXECUTED on one GPU (--share-gpu: gloo, every rank on cuda:0) for the three workloads;
# the multirank file with its new bench test; the teacher-forced tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c09; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_multirank.py -q -m gpu > $O/multirank.txt 2>&1; tail -4 $O/multirank.txt | cut -c1-300
timeout 900 python -m pytest tests/test_hip_round5.py -q -m gpu -s -k "teacher_forced" > $O/teacher.txt 2>&1; grep "teacher-forced\|passed\|failed" $O/teacher.txt | cut -c1-500
for cfg in "collab 2" "collab 4" "ddi 2" "citation2 2"; do
  set -- $cfg
  timeout 900 python bench.py --workload $1 --gpus $2 --share-gpu --steps 8 --warmup 3 --no-strong > $O/bench_$1_w$2.json 2> $O/bench_$1_w$2.err
  python -c "
import json
try:
    r = json.loads(open('$O/bench_$1_w$2.json').read().strip().splitlines()[-1])
    print('$1 world $2 shared-gpu:', round(r['ms_per_step'], 3), 'ms/step, value', round(r['value']), 'exchange', r['config']['parallelism'][:60], '| choice', (r.get('dp_prediction') or {}).get('choice'), '| phases', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in (r.get('dp_phases') or {}).items() if k != 'note'})
except Exception as e:
    print('$1 world $2 FAILED', e); print(open('$O/bench_$1_w$2.err').read()[-1500:])
"
done
