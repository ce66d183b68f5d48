#!/bin/bash
mkdir -p gpurun_out/c17
for m in f32 bf16x3; do
  PLNLP_GEMM_MATH=$m timeout 600 python -m pytest tests/test_hip_round2.py -q -m gpu -s -k "reference_style" 2>&1 | grep -E "teacher|passed|failed" > gpurun_out/c17/teacher_$m.log
  echo "== $m"; cat gpurun_out/c17/teacher_$m.log
done
PLNLP_GEMM_MATH=bf16x3 timeout 2400 python -m pytest tests/test_hip_parity.py::test_hits20_training_parity_ddi_recipe tests/test_hip_round2.py::test_hits20_ddi_recipe_parity_over_seeds -q -m gpu -s 2>&1 | grep -E "Hits@20|passed|failed|Error|assert" > gpurun_out/c17/hits20_x3.log
cat gpurun_out/c17/hits20_x3.log
