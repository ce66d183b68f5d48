#!/bin/bash
# round 6, call 6: block-kernel tests again (row-dot bits); what a ddi_wide / collab_wide statistic can see (clean vs single-term bf16)
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_hip_round6.py tests/test_hip_round5.py -q -m gpu -k "block or head" 2>&1 | tail -5 > gpurun_out/r06/call06_tests.txt
cat gpurun_out/r06/call06_tests.txt
timeout 1500 python scripts/calibrate_wide_parity.py ddi_wide 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/call06_calibrate_ddi_wide.txt
cat gpurun_out/r06/call06_calibrate_ddi_wide.txt
timeout 900 python scripts/calibrate_wide_parity.py collab_wide 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/call06_calibrate_collab_wide.txt
cat gpurun_out/r06/call06_calibrate_collab_wide.txt
