#!/bin/bash
# round 3, call 2: the bench line with train_epoch / eval_scoring / step-shaped rooflines; config-5 property test
mkdir -p gpurun_out/r03c02
python -m pytest tests/test_hip_round3.py -x -q -m gpu -k full_size > gpurun_out/r03c02/round3.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r03c02/bench_default.json 2> gpurun_out/r03c02/bench_default.err
echo "bench rc=$?" >> gpurun_out/r03c02/bench_default.err
python bench.py --workload ddi --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-stress > gpurun_out/r03c02/bench_ddi.json 2> gpurun_out/r03c02/bench_ddi.err
echo "bench rc=$?" >> gpurun_out/r03c02/bench_ddi.err
python bench.py --workload citation2 --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-stress --epoch-steps 20 > gpurun_out/r03c02/bench_cit.json 2> gpurun_out/r03c02/bench_cit.err
echo "bench rc=$?" >> gpurun_out/r03c02/bench_cit.err
tail -n 3 gpurun_out/r03c02/round3.log
tail -n 4 gpurun_out/r03c02/*.err
