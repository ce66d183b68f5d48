#!/bin/bash
# round 5, call 12: what could pre-split A operands buy the stationary-weights GEMM at most?  The product library against two
# scratch builds of gemm_x3s.hip whose K loop does NO split arithmetic (the A terms are the raw bits of the loaded floats: wrong
# numbers, right instruction mix minus the split) -- `nosplit` with today's two 16-byte loads per lane per K-step, `nosplit3` with
# a third one (the 48 bytes per 8 elements three bf16 term planes would cost) -- interleaved on one box, 3 repetitions
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c12; mkdir -p $O
SH=collab_step_fwd,collab_step_dgrad,ddi_pred_fwd,ddi_pred_dgrad,cit_l2_fwd_k200,collab_fwd
for rep in 1 2 3; do
  for v in product nosplit nosplit3; do
    lib=$GRAFT_REPO_ROOT/plnlp_amd/libplnlp_hip.so; [ $v != product ] && lib=$GRAFT_REPO_ROOT/ab_x3s/libplnlp_hip_$v.so
    PLNLP_HIP_LIB=$lib timeout 600 python scripts/bench_gemm.py --shapes $SH --math bf16x3 2> $O/gemm_${v}_$rep.err | sed "s/^{/{\"variant\": \"$v\", \"rep\": $rep, /" >> $O/gemm_ab.jsonl
  done
done
python - <<PY
import json, collections
rows = [json.loads(l) for l in open("$O/gemm_ab.jsonl") if l.startswith("{")]
by = collections.defaultdict(list)
for r in rows: by[(r["shape"], r["variant"])].append(r["ms"])
for shape in dict.fromkeys(r["shape"] for r in rows):
    base = sorted(by[(shape, "product")])[1]
    print(shape, {v: (round(sorted(by[(shape, v)])[1], 4), round(sorted(by[(shape, v)])[1] / base, 3)) for v in ("product", "nosplit", "nosplit3")})
PY
