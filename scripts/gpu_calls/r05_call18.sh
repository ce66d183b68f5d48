#!/bin/bash
# round 5, call 18: what the driver runs at round end, on the final tree: the whole -m gpu suite (-x), smoke(), the default bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c18; mkdir -p $O
( time timeout 3000 python -m pytest tests/ -x -q -m gpu ) > $O/gpu_suite.txt 2>&1; tail -8 $O/gpu_suite.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err; python -c "
import json; r = json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print('collab', round(r['ms_per_step'], 4), 'ms', round(r['value']/1e6, 2), 'M; roofline', round(r['roofline']['frac'], 3), 'cpu', round(r['cpu_baseline']['value']), 'families', r['kernel_families_per_step'])"
