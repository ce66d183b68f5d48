#!/bin/bash
# round 5, call 11: the whole -m gpu suite on the tree without the edge-lists builder (ordering as the driver runs it), then
# the three bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c11; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/gpu_suite.txt 2>&1; tail -6 $O/gpu_suite.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for w in collab ddi citation2; do
  timeout 900 python bench.py --workload $w --no-cpu-baseline --no-parity > $O/bench_$w.json 2> $O/bench_$w.err
  python -c "
import json; r = json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', round(r['ms_per_step'], 4), 'ms', round(r['value']/1e6, 2), 'M edges/s', 'f32', round(r.get('ms_per_step_f32_mfma', 0), 3), 'roofline', round(r['roofline']['kernel_ms'], 3), r['roofline'].get('frac'))"
done
