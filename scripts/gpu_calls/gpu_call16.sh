#!/bin/bash
mkdir -p gpurun_out/c16
for m in f32 bf16x3; do
  PLNLP_GEMM_MATH=$m timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -s -k "trajectory_matches" 2>&1 | grep -E "per epoch|passed|failed" > gpurun_out/c16/traj_$m.log
  echo "== $m"; cat gpurun_out/c16/traj_$m.log
done
PLNLP_GEMM_MATH=bf16x3 timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_round2.py -q -m gpu --deselect tests/test_hip_parity.py::test_hits20_training_parity_ddi_recipe --deselect tests/test_hip_round2.py::test_hits20_ddi_recipe_parity_over_seeds 2>&1 | tail -40 > gpurun_out/c16/all_x3.log
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/c16/all_x3.log
