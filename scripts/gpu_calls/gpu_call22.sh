#!/bin/bash
mkdir -p gpurun_out/c22
SH=collab_fwd_plain,ddi_pred_fwd,collab_dgrad_T,collab_wgrad_T
echo "== baseline"; python scripts/bench_gemm.py --math bf16x3 --shapes $SH 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], d['ms'], d['TFLOPs'])"
for v in NOSPLIT NOSTORE NOGLOAD WG2; do
  echo "== $v"
  PLNLP_HIP_LIB=$PWD/plnlp_amd/build/abl/libplnlp_hip_x3_$v.so python scripts/bench_gemm.py --math bf16x3 --shapes $SH 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], d['ms'], d['TFLOPs'])"
done > gpurun_out/c22/abl.txt 2>&1
cat gpurun_out/c22/abl.txt
timeout 1800 python -m pytest tests/test_hip_parity.py tests/test_hip_round2.py -q -m gpu --deselect tests/test_hip_parity.py::test_hits20_training_parity_ddi_recipe --deselect tests/test_hip_round2.py::test_hits20_ddi_recipe_parity_over_seeds 2>&1 | tail -8 > gpurun_out/c22/all.log
cat gpurun_out/c22/all.log
