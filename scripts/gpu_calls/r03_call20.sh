#!/bin/bash
# round 3, call 20: trained-regime parity (final fixture); GEMM register fix A/B; full suite
mkdir -p gpurun_out/r03c20
python -m pytest tests/test_hip_round3.py -x -q -m gpu -k "trained_regime" -s > gpurun_out/r03c20/trained.log 2>&1
echo "rc=$?" >> gpurun_out/r03c20/trained.log
cp gpurun_out/trained_parity_table.txt gpurun_out/r03c20/ 2>/dev/null
for r in 1 2; do
python scripts/bench_gemm.py --shapes collab_fwd,collab_dgrad_T,collab_wgrad_T,collab_wgrad --math bf16x3 --repeats 5 > gpurun_out/r03c20/gemm_default_$r.jsonl 2>&1
PLNLP_HIP_LIB=$PWD/plnlp_amd/build/abl/libplnlp_hip_wgrad2.so python scripts/bench_gemm.py --shapes collab_fwd,collab_dgrad_T,collab_wgrad_T,collab_wgrad --math bf16x3 --repeats 5 > gpurun_out/r03c20/gemm_wgrad2_$r.jsonl 2>&1
done
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-stress > gpurun_out/r03c20/bench_default.json 2> gpurun_out/r03c20/bench_default.err
python -m pytest tests -x -q -m gpu --durations=6 > gpurun_out/r03c20/suite.log 2>&1
echo "rc=$?" >> gpurun_out/r03c20/suite.log
tail -n 30 gpurun_out/r03c20/trained.log | cut -c1-200; tail -n 12 gpurun_out/r03c20/suite.log; cat gpurun_out/r03c20/gemm_*.jsonl | grep -v amdgpu | cut -c1-220
