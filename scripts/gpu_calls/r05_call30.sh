#!/bin/bash
# stationary-weights GEMM with A through LDS-DMA + counted waits: its tests, then the shapes (A/B against the committed build in ab_head/)
mkdir -p gpurun_out/r05w
timeout 1200 python -m pytest tests/test_hip_round4.py tests/test_hip_round5.py tests/test_hip_parity.py -q -x -k "stationary or x3s or gemm or head or rowdot or linear or mlp" > gpurun_out/r05w/tests_x3s.txt 2>&1; tail -3 gpurun_out/r05w/tests_x3s.txt
for rep in 1 2; do
for v in new head; do
  if [ $v = new ]; then unset PLNLP_HIP_LIB; else export PLNLP_HIP_LIB=$PWD/ab_head/libplnlp_hip.so; fi
  echo "== $v"
  timeout 300 python scripts/bench_gemm.py --shapes collab_fwd_plain,collab_step_fwd,collab_step_dgrad,ddi_pred_fwd,ddi_pred_dgrad,cit_l2_fwd_k200,cit_in_fwd_k192 --math bf16x3 --iters 10 2>&1 | grep '"shape"' | python -c "import sys,json; [print('  ', json.loads(l)['shape'], json.loads(l)['ms']) for l in sys.stdin]"
done; done
