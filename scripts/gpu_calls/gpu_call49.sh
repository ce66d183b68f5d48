#!/bin/bash
mkdir -p gpurun_out/c49
timeout 1800 python -m pytest tests/test_hip_parity.py tests/test_hip_round2.py -q -m gpu --deselect tests/test_hip_parity.py::test_hits20_training_parity_ddi_recipe --deselect tests/test_hip_round2.py::test_hits20_ddi_recipe_parity_over_seeds > gpurun_out/c49/all.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/c49/all.log | tail -8
python __graft_entry__.py smoke 2>&1 | tail -1
