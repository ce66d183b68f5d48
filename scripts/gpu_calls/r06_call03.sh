#!/bin/bash
# round 6, call 3: the continuous-pipeline x3b: bits vs x3s, times, phase trace
mkdir -p gpurun_out/r06
timeout 600 python scripts/bench_gemm.py --math blk --iters 10 --repeats 3 \
  --shapes collab_step_fwd,collab_step_dgrad,collab_fwd_plain,ddi_pred_fwd,ddi_pred_dgrad,cit_l2_fwd_k200,cit_in_fwd_k192 2>&1 | grep -v "^$" | grep -v amdgpu.ids > gpurun_out/r06/call03_blk.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r06/call03_blk.jsonl'):
    try: r=json.loads(l)
    except Exception: print(l.strip()); continue
    if 'bits_equal_x3s' in r: print(r)
    else: print(r['shape'], r['stationary_b'], r['ms'], r['frac_of_2500'])
PY
PLNLP_HIP_LIB=$PWD/ab_x3b/lib_trace.so timeout 600 python ab_x3b/trace.py collab_step_fwd ddi_pred_fwd ddi_pred_dgrad collab_step_dgrad cit_l2_fwd_k200 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/call03_trace.jsonl
cat gpurun_out/r06/call03_trace.jsonl
