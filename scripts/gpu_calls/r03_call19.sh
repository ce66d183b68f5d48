#!/bin/bash
# round 3, call 19: captured vs eager step (bucket 512, host work measured), twice each, interleaved
mkdir -p gpurun_out/r03c19
for r in 1 2; do
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-stress --no-roofline > gpurun_out/r03c19/bench_capture_$r.json 2> gpurun_out/r03c19/bench_capture_$r.err
PLNLP_CAPTURE=0 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-stress --no-roofline > gpurun_out/r03c19/bench_eager_$r.json 2> gpurun_out/r03c19/bench_eager_$r.err
done
python bench.py --workload ddi --steps 20 --warmup 10 --no-cpu-baseline --no-parity --no-stress --no-roofline > gpurun_out/r03c19/bench_ddi_capture.json 2> gpurun_out/r03c19/bench_ddi_capture.err
PLNLP_CAPTURE=0 python bench.py --workload ddi --steps 20 --warmup 10 --no-cpu-baseline --no-parity --no-stress --no-roofline > gpurun_out/r03c19/bench_ddi_eager.json 2> gpurun_out/r03c19/bench_ddi_eager.err
python bench.py --workload citation2 --steps 10 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline --epoch-steps 20 > gpurun_out/r03c19/bench_cit_capture.json 2> gpurun_out/r03c19/bench_cit_capture.err
PLNLP_CAPTURE=0 python bench.py --workload citation2 --steps 10 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline --epoch-steps 20 > gpurun_out/r03c19/bench_cit_eager.json 2> gpurun_out/r03c19/bench_cit_eager.err
tail -n 2 gpurun_out/r03c19/*.err
