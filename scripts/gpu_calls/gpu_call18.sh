#!/bin/bash
mkdir -p gpurun_out/c18
timeout 900 python -m pytest tests/test_hip_round2.py tests/test_hip_parity.py -q -m gpu -s -k "split_bf16 or gemm or trajectory or reference_style or encoder or mlp_predictor" 2>&1 | grep -E "a_trans=|passed|failed|FAILED|Error" > gpurun_out/c18/tests.log
cat gpurun_out/c18/tests.log
bash scripts/pmc_gemm3.sh > gpurun_out/c18/pmc.log 2>&1
cat gpurun_out/c18/pmc.log | cut -c1-1500
