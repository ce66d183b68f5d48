#!/bin/bash
# clock + occupancy + MFMA counters of the GEMM kernels (separate --pmc passes: GRBM, SQ)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SH=collab_fwd,collab_fwd_plain,collab_fwd_plain_7rounds,ddi_pred_fwd,square4k
rm -rf gpurun_out/pmc_g2
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -f csv -d gpurun_out/pmc_g2/a -o g -- python3 scripts/bench_gemm.py --shapes $SH --iters 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES -f csv -d gpurun_out/pmc_g2/b -o g -- python3 scripts/bench_gemm.py --shapes $SH --iters 3 > /dev/null 2>&1
python3 scripts/pmc_collect.py gemm_f32_kernel gpurun_out/r02/pmc_gemm2.json "gpurun_out/pmc_g2/**/*counter_collection.csv"
rm -rf gpurun_out/pmc_g2
