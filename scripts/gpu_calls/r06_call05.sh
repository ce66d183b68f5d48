#!/bin/bash
# round 6, call 5: gemm_x3b's tests after the clean-up
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_hip_round6.py tests/test_hip_round4.py tests/test_hip_round5.py -q -m gpu -k "block or stationary or head or counters" 2>&1 | tail -12 > gpurun_out/r06/call05_tests.txt
cat gpurun_out/r06/call05_tests.txt
