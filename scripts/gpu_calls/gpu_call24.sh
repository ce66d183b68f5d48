#!/bin/bash
# clock and MFMA-busy of the split-bf16 GEMM and its ablations (is the kernel power-limited?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c24
SH=ddi_pred_fwd,collab_wgrad_T
for v in BASE NOSPLIT NOSTORE NOGLOAD; do
  if [ $v = BASE ]; then unset PLNLP_HIP_LIB; else export PLNLP_HIP_LIB=$PWD/plnlp_amd/build/abl/libplnlp_hip_x3_$v.so; fi
  rm -rf gpurun_out/pmc_t
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -f csv -d gpurun_out/pmc_t/a -o g -- python3 scripts/bench_gemm.py --math bf16x3 --shapes $SH --iters 10 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -f csv -d gpurun_out/pmc_t/b -o g -- python3 scripts/bench_gemm.py --math bf16x3 --shapes $SH --iters 10 > /dev/null 2>&1
  python3 scripts/pmc_collect.py gemm_f32_kernel gpurun_out/c24/pmc_$v.json "gpurun_out/pmc_t/**/*counter_collection.csv" > /dev/null
  python3 - <<PY
import json
d=json.load(open("gpurun_out/c24/pmc_$v.json"))
for k,x in d.items():
    us=x["kernel_us_under_pmc"]; clk=x["GRBM_GUI_ACTIVE"]/8/(us*1e3)
    print("$v", k.split("|")[0][-40:], k.split("=")[-1], "us", round(us,1), "clock GHz", round(clk,3), "MFMA busy", round(x["SQ_VALU_MFMA_BUSY_CYCLES"]/(1024*us*1e3*clk),3),
          "wait_any", round(x["SQ_WAIT_ANY"]/x["SQ_WAVE_CYCLES"],3), "wait_inst", round(x["SQ_WAIT_INST_ANY"]/x["SQ_WAVE_CYCLES"],3), "active", round(x["SQ_ACTIVE_INST_ANY"]/x["SQ_WAVE_CYCLES"],3))
PY
done
rm -rf gpurun_out/pmc_t
