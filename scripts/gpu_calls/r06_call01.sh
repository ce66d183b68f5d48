#!/bin/bash
# round 6, call 1: the whole-block stationary GEMM (csrc/gemm_x3b.hip) -- parity of the GEMM tests, then x3s vs x3b per shape
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_hip_round4.py tests/test_hip_round5.py -x -q -m gpu -k "stationary or head or counters or wide or weight_grad" 2>&1 | tail -15 > gpurun_out/r06/call01_tests.txt
cat gpurun_out/r06/call01_tests.txt
timeout 900 python scripts/bench_gemm.py --math blk --iters 10 --repeats 3 \
  --shapes collab_step_fwd,collab_step_dgrad,collab_fwd_plain,collab_fwd,ddi_pred_fwd,ddi_pred_dgrad,cit_l2_fwd_k200,cit_in_fwd_k192 2>&1 | grep -v "^$" > gpurun_out/r06/call01_blk.jsonl
cat gpurun_out/r06/call01_blk.jsonl
