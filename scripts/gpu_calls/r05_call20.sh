#!/bin/bash
# the wide weight-gradient kernel (csrc/gemm_wgw.hip): its tests, then the citation2 shapes against the tile kernel
mkdir -p gpurun_out/r05w
timeout 900 python -m pytest tests/test_hip_round5.py -q -x -k "wide_weight" -s > gpurun_out/r05w/tests.txt 2>&1
echo "tests exit $?" >> gpurun_out/r05w/tests.txt
tail -15 gpurun_out/r05w/tests.txt
timeout 600 python scripts/bench_gemm.py --shapes cit_l2_wgrad,collab_wgrad,collab_wgrad_256x512,ddi_pred_wgrad --math wide --error --iters 10 > gpurun_out/r05w/bench_wide.jsonl 2>&1
cat gpurun_out/r05w/bench_wide.jsonl
