#!/bin/bash
# round 3, call 21: row-sparse sharded step (tests + 1-rank RCCL timings), trained parity, whole suite
mkdir -p gpurun_out/r03c21
python -m pytest tests/test_hip_round3.py -x -q -m gpu -k "sharded_step_row_sparse or trained_regime" -s > gpurun_out/r03c21/new_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r03c21/new_tests.log
cp gpurun_out/trained_parity_table.txt gpurun_out/r03c21/ 2>/dev/null
for mode in shard grads scores; do
python bench.py --force-dist --dp-exchange $mode --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > gpurun_out/r03c21/bench_1rank_$mode.json 2> gpurun_out/r03c21/bench_1rank_$mode.err
echo "rc=$?" >> gpurun_out/r03c21/bench_1rank_$mode.err
done
PLNLP_SHARD_SPARSE=0 python bench.py --force-dist --dp-exchange shard --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > gpurun_out/r03c21/bench_1rank_shard_dense.json 2> gpurun_out/r03c21/bench_1rank_shard_dense.err
python bench.py --workload ddi --force-dist --dp-exchange shard --steps 20 --warmup 6 --no-cpu-baseline --no-parity --no-stress --no-roofline > gpurun_out/r03c21/bench_1rank_ddi_shard.json 2> gpurun_out/r03c21/bench_1rank_ddi_shard.err
python -m pytest tests -x -q -m gpu --durations=6 > gpurun_out/r03c21/suite.log 2>&1
echo "rc=$?" >> gpurun_out/r03c21/suite.log
tail -n 30 gpurun_out/r03c21/new_tests.log | cut -c1-200; tail -n 3 gpurun_out/r03c21/*.err; tail -n 8 gpurun_out/r03c21/suite.log
