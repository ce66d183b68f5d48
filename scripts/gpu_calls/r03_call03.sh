#!/bin/bash
# round 3, call 3: streamed permutation tests; bench line again (train_epoch with the streamed permutation)
mkdir -p gpurun_out/r03c03
python -m pytest tests/test_hip_round3.py -x -q -m gpu -k "streamed" > gpurun_out/r03c03/round3.log 2>&1
python -m pytest tests/test_hip_parity.py tests/test_hip_round2.py -x -q -m gpu > gpurun_out/r03c03/suite.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress > gpurun_out/r03c03/bench_default.json 2> gpurun_out/r03c03/bench_default.err
echo "bench rc=$?" >> gpurun_out/r03c03/bench_default.err
PLNLP_STREAM_PERMUTATION=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > gpurun_out/r03c03/bench_nostream.json 2> gpurun_out/r03c03/bench_nostream.err
python bench.py --workload citation2 --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-stress --epoch-steps 20 > gpurun_out/r03c03/bench_cit.json 2> gpurun_out/r03c03/bench_cit.err
echo "bench rc=$?" >> gpurun_out/r03c03/bench_cit.err
tail -n 3 gpurun_out/r03c03/round3.log gpurun_out/r03c03/suite.log
tail -n 4 gpurun_out/r03c03/*.err
