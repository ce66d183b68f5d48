#!/bin/bash
# round 6, call 13: why the ddi parity tests fail on the dense aggregation
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_hip_round4.py tests/test_hip_round6.py tests/test_hip_round5.py -q -m gpu -x -k "full_size_ddi or (teacher_forced_steps_with and ddi_wide)" 2>&1 | grep -v amdgpu.ids | tail -60 > gpurun_out/r06/call13.txt; cut -c1-400 gpurun_out/r06/call13.txt
