#!/bin/bash
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_hip_round4.py tests/test_hip_round6.py tests/test_hip_round5.py -q -m gpu -s -k "dense or full_size_ddi or (teacher_forced and ddi_wide) or (trained_regime and ddi_wide)" 2>&1 | grep -v amdgpu.ids | grep -E "^E  |assert|passed|failed|FAILED|teacher-forced|ddi_wide" | cut -c1-420 > gpurun_out/r06/call14.txt; cat gpurun_out/r06/call14.txt
