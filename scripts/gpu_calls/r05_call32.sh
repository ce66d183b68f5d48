#!/bin/bash
# final driver-style check: the new edge cases, smoke(), the default bench line
mkdir -p gpurun_out/r05final
timeout 900 python -m pytest tests/test_hip_round5.py -q -x -k "wide_weight" > gpurun_out/r05final/tests.txt 2>&1; tail -2 gpurun_out/r05final/tests.txt
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/r05final/bench_default.json 2> gpurun_out/r05final/bench_default.err; tail -c 600 gpurun_out/r05final/bench_default.json; echo
python -c "
import json; d=json.loads(open('gpurun_out/r05final/bench_default.json').read().strip().splitlines()[-1]); print(d['metric'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
