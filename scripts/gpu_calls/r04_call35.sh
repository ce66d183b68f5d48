#!/bin/bash
# round 4, call 35: counters of the split-bf16 WEIGHT-GRADIENT GEMM (tile kernel, A transposed, split-K) on the path's shapes
# (separate --pmc passes, as scripts/refresh_profiles_r04.sh does for the forward / data-gradient shapes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r04c35; mkdir -p $R
SH=collab_wgrad_T,ddi_pred_wgrad,collab_wgrad
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -f csv -d $R/pmc_g/a -o g -- python3 scripts/bench_gemm.py --math bf16x3 --shapes $SH --iters 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES -f csv -d $R/pmc_g/b -o g -- python3 scripts/bench_gemm.py --math bf16x3 --shapes $SH --iters 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY -f csv -d $R/pmc_g/c -o g -- python3 scripts/bench_gemm.py --math bf16x3 --shapes $SH --iters 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA -f csv -d $R/pmc_g/d -o g -- python3 scripts/bench_gemm.py --math bf16x3 --shapes $SH --iters 3 > /dev/null 2>&1
python3 scripts/pmc_collect.py gemm $R/gemm_pmc_wgrad_raw.json "$R/pmc_g/**/*counter_collection.csv" > /dev/null
rm -rf $R/pmc_g
python3 - <<PY
import json
d = json.load(open("$R/gemm_pmc_wgrad_raw.json"))
out = {}
for k, v in d.items():
    if v.get("launches", 0) < 3 or "SQ_WAVE_CYCLES" not in v:
        continue
    wc = v["SQ_WAVE_CYCLES"]
    v["derived"] = {"clock_GHz_per_xcd": v.get("GRBM_GUI_ACTIVE", 0) / 8 / (v["kernel_us_under_pmc"] * 1e3),
                    "wave_wait_any": v["SQ_WAIT_ANY"] / wc, "wave_issue_stall": v["SQ_WAIT_INST_ANY"] / wc,
                    "wave_lds_issue_stall": v.get("SQ_WAIT_INST_LDS", 0) / wc,
                    "valu_per_mfma": v.get("SQ_INSTS_VALU", 0) / max(v.get("SQ_INSTS_MFMA", 1), 1),
                    "lds_per_mfma": v.get("SQ_INSTS_LDS", 0) / max(v.get("SQ_INSTS_MFMA", 1), 1),
                    "lds_bank_conflict_per_lds_active": v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1),
                    "mfma_busy_per_gui_cycle": v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(v.get("GRBM_GUI_ACTIVE", 1), 1)}
    out[k] = v
json.dump(out, open("$R/gemm_pmc_wgrad.json", "w"), indent=1)
for k, v in out.items():
    print(k[:110], {a: round(b, 3) for a, b in v["derived"].items()}, round(v["kernel_us_under_pmc"], 1), "us")
PY
