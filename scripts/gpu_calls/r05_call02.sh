#!/bin/bash
# round 5, call 2: (a) tests/test_hip_multirank.py with per-case watchdogs (call 1's world-2 run hung silently);
# (b) scripts/pin_agg_forms.py -> the shipped aggregation-form table; (c) the whole -m gpu suite on the round-5 tree;
# (d) the collab bench line as the round's same-box reference (WITH the roofline)
O=$GRAFT_REPO_ROOT/gpurun_out/r05c02; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_multirank.py -x -q -m gpu > $O/multirank.txt 2>&1; tail -60 $O/multirank.txt | cut -c1-300
timeout 900 python scripts/pin_agg_forms.py --out $O/agg_forms.json > $O/pin_agg_forms.txt 2>&1; tail -12 $O/pin_agg_forms.txt
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_hip_multirank.py > $O/gpu_suite.txt 2>&1; tail -15 $O/gpu_suite.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench_collab.json 2> $O/bench_collab.err; python -c "
import json; r = json.loads(open('$O/bench_collab.json').read().strip().splitlines()[-1]); print('collab', r['ms_per_step'], 'ms', r['value'], 'roofline', r['roofline']['kernel_ms'], r['roofline']['frac'])"
