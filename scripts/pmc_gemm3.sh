#!/bin/bash
# split-bf16 GEMM: where the wave cycles go (separate --pmc passes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SH=${SH:-collab_fwd_plain,ddi_pred_fwd,collab_wgrad_T,cit_l2_fwd_k200}
OUT=${OUT:-gpurun_out/r02/pmc_gemm_x3.json}
mkdir -p $(dirname $OUT)
rm -rf gpurun_out/pmc_g3
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -f csv -d gpurun_out/pmc_g3/a -o g -- python3 scripts/bench_gemm.py --math bf16x3 --shapes $SH --iters 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES -f csv -d gpurun_out/pmc_g3/b -o g -- python3 scripts/bench_gemm.py --math bf16x3 --shapes $SH --iters 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY -f csv -d gpurun_out/pmc_g3/c -o g -- python3 scripts/bench_gemm.py --math bf16x3 --shapes $SH --iters 3 > gpurun_out/pmc_g3_c.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA -f csv -d gpurun_out/pmc_g3/d -o g -- python3 scripts/bench_gemm.py --math bf16x3 --shapes $SH --iters 3 > gpurun_out/pmc_g3_d.log 2>&1
python3 scripts/pmc_collect.py ${FILTER:-gemm_f32_kernel} $OUT "gpurun_out/pmc_g3/**/*counter_collection.csv" > /dev/null
tail -n 3 gpurun_out/pmc_g3_c.log gpurun_out/pmc_g3_d.log
rm -rf gpurun_out/pmc_g3
python3 - <<PY
import json
d=json.load(open("$OUT"))
for k,v in d.items():
    print(k[:90]); print("   ", {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
PY
