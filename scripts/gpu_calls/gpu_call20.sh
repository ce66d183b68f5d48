#!/bin/bash
mkdir -p gpurun_out/c20
timeout 900 python -m pytest tests/test_hip_round2.py tests/test_hip_parity.py -q -m gpu -k "split_bf16 or gemm or encoder or mlp_predictor or wgrad" 2>&1 | tail -8 > gpurun_out/c20/tests.log
cat gpurun_out/c20/tests.log
timeout 900 python scripts/bench_gemm.py --math ab --shapes collab_fwd,collab_fwd_plain,collab_dgrad,collab_wgrad,ddi_pred_fwd,ddi_pred_wgrad,ddi_enc_fwd,square4k,cit_in_fwd_k180,cit_l2_fwd_k200,collab_dgrad_T,collab_wgrad_T > gpurun_out/c20/gemm_ab.jsonl 2> gpurun_out/c20/gemm_ab.err
python - <<'PY'
import json
for l in open("gpurun_out/c20/gemm_ab.jsonl"):
    d=json.loads(l); print(d["shape"], d["math"], d["ms"], d["TFLOPs"], d.get("frac_of_2500"))
PY
SH=ddi_pred_fwd,collab_wgrad_T OUT=gpurun_out/c20/pmc.json bash scripts/pmc_gemm3.sh 2>&1 | cut -c1-1200
