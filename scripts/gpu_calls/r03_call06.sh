#!/bin/bash
# round 3, call 6: captured step -- bit-identity tests, bench with and without capture; trained-regime parity; drift probe
mkdir -p gpurun_out/r03c06
python -m pytest tests/test_hip_round3.py -x -q -m gpu -k "captured" > gpurun_out/r03c06/capture_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r03c06/capture_tests.log
python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > gpurun_out/r03c06/bench_capture.json 2> gpurun_out/r03c06/bench_capture.err
echo "rc=$?" >> gpurun_out/r03c06/bench_capture.err
PLNLP_CAPTURE=0 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > gpurun_out/r03c06/bench_eager.json 2> gpurun_out/r03c06/bench_eager.err
python -m pytest tests/test_hip_round3.py -x -q -m gpu -k "trained_regime" -s > gpurun_out/r03c06/trained.log 2>&1
echo "rc=$?" >> gpurun_out/r03c06/trained.log
for c in sage_mlp_whinge_noweight sage_mlp_auc; do
  python scripts/probe_drift.py $c > gpurun_out/r03c06/probe_$c.txt 2>&1
done
tail -n 25 gpurun_out/r03c06/capture_tests.log; tail -n 3 gpurun_out/r03c06/*.err; tail -n 30 gpurun_out/r03c06/trained.log
