#!/bin/bash
# round 6, call 7: the new parity tests (dropout-on teacher-forced, reverse teacher-forced, mutation at width), the reworked wide legs
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests/test_hip_round6.py -q -m gpu -s 2>&1 | grep -v amdgpu.ids | tail -40 > gpurun_out/r06/call07_tests_r6.txt
cat gpurun_out/r06/call07_tests_r6.txt
timeout 2400 python -m pytest tests/test_hip_round5.py -q -m gpu -s -k "trained_regime or teacher" 2>&1 | grep -v amdgpu.ids | tail -30 > gpurun_out/r06/call07_tests_r5.txt
cat gpurun_out/r06/call07_tests_r5.txt
