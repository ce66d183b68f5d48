#!/bin/bash
# wide weight gradient: what the step time is made of (ablated builds of csrc/gemm_wgw.hip in ab_wgw/, scratch)
mkdir -p gpurun_out/r05w
for rep in 1 2; do
for v in base b1; do
  if [ $v = base ]; then unset PLNLP_HIP_LIB; else export PLNLP_HIP_LIB=$PWD/ab_wgw/lib_$v.so; fi
  echo "== $v" 
  timeout 300 python scripts/bench_gemm.py --shapes cit_l2_wgrad --math wide --iters 5 2>&1 | grep '"wide"' | python -c "import sys,json; [print(json.loads(l)['shape'], json.loads(l)['ms']) for l in sys.stdin]"
done; done > gpurun_out/r05w/ablate.txt 2>&1
cat gpurun_out/r05w/ablate.txt
