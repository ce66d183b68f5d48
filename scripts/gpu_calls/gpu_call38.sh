#!/bin/bash
mkdir -p gpurun_out/c38
timeout 1800 python -m pytest tests/test_hip_parity.py tests/test_hip_round2.py -q -m gpu --deselect tests/test_hip_parity.py::test_hits20_training_parity_ddi_recipe --deselect tests/test_hip_round2.py::test_hits20_ddi_recipe_parity_over_seeds 2>&1 | grep -E "^FAILED|^E  |Error|assert" | head -40 > gpurun_out/c38/fail.log
cat gpurun_out/c38/fail.log
