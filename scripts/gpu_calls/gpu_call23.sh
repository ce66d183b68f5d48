#!/bin/bash
mkdir -p gpurun_out/c23
SH=collab_fwd_plain,ddi_pred_fwd,collab_dgrad_T,collab_wgrad_T,cit_l2_fwd_k200,ddi_enc_fwd
{
echo "== baseline"; python scripts/bench_gemm.py --math bf16x3 --shapes $SH 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], d['ms'], d['TFLOPs'])"
for v in SETS4 SETS6; do
  echo "== $v"
  PLNLP_HIP_LIB=$PWD/plnlp_amd/build/abl/libplnlp_hip_x3_$v.so python scripts/bench_gemm.py --math bf16x3 --shapes $SH 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print(d['shape'], d['ms'], d['TFLOPs'])
    except Exception: print(l.strip()[:200])"
done; } > gpurun_out/c23/abl.txt 2>&1
cat gpurun_out/c23/abl.txt
