#!/bin/bash
# round 6, call 2: phase times inside gemm_x3b_kernel (scratch trace build)
mkdir -p gpurun_out/r06
PLNLP_HIP_LIB=$PWD/ab_x3b/lib_trace.so timeout 600 python ab_x3b/trace.py collab_step_fwd collab_fwd_plain ddi_pred_fwd ddi_pred_dgrad collab_step_dgrad cit_l2_fwd_k200 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/call02_trace.jsonl
cat gpurun_out/r06/call02_trace.jsonl
