#!/bin/bash
mkdir -p gpurun_out/c27
{
for sl in 512 768 1024; do
  echo "== PLNLP_SPLIT_K_SLOTS=$sl"
  PLNLP_SPLIT_K_SLOTS=$sl python scripts/bench_gemm.py --math bf16x3 --repeats 5 --shapes collab_wgrad,ddi_pred_wgrad,collab_wgrad_T,ddi_enc_fwd 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], d['ms'], d['TFLOPs'])"
done; } > gpurun_out/c27/splitk.txt
cat gpurun_out/c27/splitk.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress > gpurun_out/c27/bench_collab.json 2> gpurun_out/c27/bench_collab.err; tail -2 gpurun_out/c27/bench_collab.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c27/bench_collab.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"]); print(json.dumps(d["roofline_mfma"], indent=1))
PY
