#!/bin/bash
# stationary-weights GEMM: 4 vs 8 waves per workgroup on the wide-tile shapes (same box, interleaved)
mkdir -p gpurun_out/r05q
timeout 900 python scripts/bench_gemm.py --math wb --error --iters 10 --shapes cit_l2_fwd_k200,cit_in_fwd_k192,collab_fwd_plain,collab_dgrad,collab_step_fwd,collab_step_dgrad,ddi_pred_fwd,ddi_pred_dgrad 2>/dev/null | tee gpurun_out/r05q/gemm_wb.jsonl | cut -c1-330
