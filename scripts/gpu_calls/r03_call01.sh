#!/bin/bash
# round 3, call 1: config-5 tests, the whole -m gpu suite, the default bench line (baseline for this round)
mkdir -p gpurun_out/r03c01
python -m pytest tests/test_hip_round3.py -x -q -m gpu --durations=8 > gpurun_out/r03c01/round3.log 2>&1
echo "round3 rc=$?" >> gpurun_out/r03c01/round3.log
python -m pytest tests -x -q -m gpu --durations=10 > gpurun_out/r03c01/gpu_suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/r03c01/gpu_suite.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r03c01/bench_default.json 2> gpurun_out/r03c01/bench_default.err
python bench.py --workload rmat --scale 0.25 --steps 5 --warmup 2 > gpurun_out/r03c01/bench_rmat.json 2> gpurun_out/r03c01/bench_rmat.err
tail -5 gpurun_out/r03c01/round3.log gpurun_out/r03c01/gpu_suite.log
