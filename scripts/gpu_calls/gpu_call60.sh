#!/bin/bash
mkdir -p gpurun_out/c60
timeout 2400 python -m pytest tests/test_hip_parity.py::test_hits20_training_parity_ddi_recipe tests/test_hip_round2.py::test_hits20_ddi_recipe_parity_over_seeds -q -m gpu -s 2>&1 | grep -E "Hits@20|passed|failed|Error|assert" > gpurun_out/c60/hits20.log
cat gpurun_out/c60/hits20.log
timeout 1800 python -m pytest tests/test_hip_parity.py tests/test_hip_round2.py -q -m gpu --deselect tests/test_hip_parity.py::test_hits20_training_parity_ddi_recipe --deselect tests/test_hip_round2.py::test_hits20_ddi_recipe_parity_over_seeds > gpurun_out/c60/all.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/c60/all.log | tail -5
