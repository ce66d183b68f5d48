#!/bin/bash
# stationary-weights GEMM (csrc/gemm_x3s.hip): what its time is made of (ablated builds in ab_x3s/, scratch)
mkdir -p gpurun_out/r05w
for rep in 1 2; do
for v in base c1 c2 c3 c4 c5 c6; do
  if [ $v = base ]; then unset PLNLP_HIP_LIB; else export PLNLP_HIP_LIB=$PWD/ab_x3s/lib_$v.so; fi
  echo "== $v"
  timeout 300 python scripts/bench_gemm.py --shapes collab_fwd_plain,ddi_pred_dgrad,cit_l2_fwd_k200 --math bf16x3 --iters 10 2>&1 | grep '"shape"' | python -c "import sys,json; [print('  ', json.loads(l)['shape'], json.loads(l)['ms']) for l in sys.stdin]"
done; done > gpurun_out/r05w/ablate_x3s.txt 2>&1
cat gpurun_out/r05w/ablate_x3s.txt
