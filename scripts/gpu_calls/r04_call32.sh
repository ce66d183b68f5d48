#!/bin/bash
# round 4, call 32: the stationary-weights GEMM on the ddi ENCODER's shapes (M = 4 267 rows): tile kernel vs x3s by tile width
O=$GRAFT_REPO_ROOT/gpurun_out/r04c32; mkdir -p $O
timeout 600 python scripts/bench_gemm.py --math nb --min-rows 4096 --shapes ddi_enc_fwd,ddi_enc_dgrad,ddi_enc_fwd_l1 --error > $O/gemm_enc.jsonl 2> $O/gemm_enc.err
python -c "
import json
for l in open('$O/gemm_enc.jsonl'):
    r = json.loads(l); print(r['shape'], r.get('stationary_b'), r['ms'], 'ms', r.get('frac_of_2500'), r.get('max_err_over_sum_abs'))
"
tail -3 $O/gemm_enc.err
