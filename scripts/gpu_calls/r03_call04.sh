#!/bin/bash
# round 3, call 4: drift probe (which elements flip), math-parametrised parity tests, non-finite GEMM test, suite after the tidy
mkdir -p gpurun_out/r03c04
for c in sage_mlp_whinge_noweight sage_mlp_auc; do
  python scripts/probe_drift.py $c > gpurun_out/r03c04/probe_$c.txt 2>&1
done
python -m pytest tests -x -q -m gpu --deselect tests/test_hip_round3.py::test_trained_regime_hits_parity_over_seeds --durations=8 > gpurun_out/r03c04/suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/r03c04/suite.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress > gpurun_out/r03c04/bench_default.json 2> gpurun_out/r03c04/bench_default.err
tail -n 15 gpurun_out/r03c04/suite.log
