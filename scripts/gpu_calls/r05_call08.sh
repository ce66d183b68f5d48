#!/bin/bash
# round 5, call 8: the whole round-5 test file (wide legs with the g12 fixture, teacher-forced steps at width), then the
# three tests call 7 flagged (head-bias association restored), then the ddi bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c08; mkdir -p $O
timeout 2400 python -m pytest tests/test_hip_round5.py -q -m gpu -s > $O/round5.txt 2>&1; grep -v amdgpu.ids $O/round5.txt | grep -v "^$" | tail -40 | cut -c1-600
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "single_step_gradients or training_trajectory or row_sparse_backward" > $O/parity3.txt 2>&1; tail -4 $O/parity3.txt | cut -c1-300
timeout 600 python bench.py --workload ddi --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_ddi.json 2> $O/bench_ddi.err; python -c "
import json; r = json.loads(open('$O/bench_ddi.json').read().strip().splitlines()[-1]); print('ddi', round(r['ms_per_step'], 4), 'ms')"
