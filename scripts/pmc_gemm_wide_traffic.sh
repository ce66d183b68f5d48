#!/bin/bash
# HBM / fabric-side traffic of the whole-block weight-gradient kernel (separate --pmc passes, MI355X_MICROARCH.md HBM section)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SH=${SH:-cit_l2_wgrad,ddi_pred_wgrad,collab_wgrad_256x512}
OUT=${OUT:-gpurun_out/r05q/gemm_wide_traffic.json}
mkdir -p $(dirname $OUT)
rm -rf gpurun_out/pmc_gt
for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  d=gpurun_out/pmc_gt/$(echo $pass | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $pass -f csv -d $d -o a -- python3 scripts/bench_gemm.py --math wide --shapes $SH --iters 3 > /dev/null 2>&1
done
python3 scripts/pmc_collect.py _kernel $OUT "gpurun_out/pmc_gt/**/*counter_collection.csv" > /dev/null
rm -rf gpurun_out/pmc_gt
python3 - <<PY
import json
d=json.load(open("$OUT"))
for k,v in d.items():
    print(k[:100]); print("   ", {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
PY
