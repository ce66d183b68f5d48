#!/bin/bash
# split-bf16 GEMM: accuracy + speed beside the f32 MFMA form, then the whole parity suite on it
mkdir -p gpurun_out/c15
timeout 900 python scripts/bench_gemm.py --math ab --error --shapes collab_fwd,collab_fwd_plain,collab_dgrad,collab_wgrad,ddi_pred_fwd,ddi_pred_wgrad,ddi_enc_fwd,square4k,cit_in_fwd_k180,cit_l2_fwd_k200,collab_dgrad_T,collab_wgrad_T > gpurun_out/c15/gemm_ab.jsonl 2> gpurun_out/c15/gemm_ab.err
cat gpurun_out/c15/gemm_ab.jsonl | cut -c1-400
tail -3 gpurun_out/c15/gemm_ab.err
timeout 1200 python -m pytest tests/test_hip_parity.py -q -m gpu -x --deselect tests/test_hip_parity.py::test_hits20_training_parity_ddi_recipe 2>&1 | tail -25 > gpurun_out/c15/parity_x3.log
tail -25 gpurun_out/c15/parity_x3.log
