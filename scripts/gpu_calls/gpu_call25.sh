#!/bin/bash
mkdir -p gpurun_out/c25
timeout 900 python -m pytest tests/test_hip_round2.py tests/test_hip_parity.py -q -m gpu -k "split_bf16 or gemm or encoder or mlp_predictor or wgrad or conv" 2>&1 | tail -5 > gpurun_out/c25/tests.log
cat gpurun_out/c25/tests.log
timeout 900 python scripts/bench_gemm.py --math bf16x3 --shapes collab_fwd,collab_fwd_plain,collab_dgrad,collab_wgrad,ddi_pred_fwd,ddi_pred_wgrad,ddi_enc_fwd,square4k,cit_in_fwd_k180,cit_l2_fwd_k200,collab_dgrad_T,collab_wgrad_T > gpurun_out/c25/gemm.jsonl 2> gpurun_out/c25/gemm.err
python - <<'PY'
import json
for l in open("gpurun_out/c25/gemm.jsonl"):
    d=json.loads(l); print(d["shape"], d["math"], d["ms"], d["TFLOPs"], d.get("frac_of_2500"))
PY
