#!/bin/bash
# clock + occupancy + MFMA counters of the GEMM kernels (separate --pmc passes: GRBM, SQ)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SH=collab_fwd,collab_fwd_plain,collab_fwd_plain_7rounds,ddi_pred_fwd,square4k
rm -rf gpurun_out/pmc_g2
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -f csv -d gpurun_out/pmc_g2/a -o g -- python3 scripts/bench_gemm.py --shapes $SH --iters 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_BUSY_CU_CYCLES -f csv -d gpurun_out/pmc_g2/b -o g -- python3 scripts/bench_gemm.py --shapes $SH --iters 3 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, json, collections, re
per = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_g2/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_f32_kernel" not in r["Kernel_Name"]: continue
        key = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void plnlp::", "") + "|grid=" + r["Grid_Size"]
        per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/pmc_g2/*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_f32_kernel" not in r["Kernel_Name"]: continue
        key = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void plnlp::", "") + "|grid=" + r["Grid_Size"]
        dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for k, c in per.items():
    d = {n: sorted(v)[len(v) // 2] for n, v in c.items()}
    if dur[k]:
        d["kernel_us_under_pmc"] = sorted(dur[k])[len(dur[k]) // 2] / 1e3
        if "GRBM_GUI_ACTIVE" in d:
            d["clock_GHz"] = d["GRBM_GUI_ACTIVE"] / (d["kernel_us_under_pmc"] * 1e3)
    out[k] = d
json.dump(out, open("gpurun_out/r02/pmc_gemm2.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:4000])
PY
rm -rf gpurun_out/pmc_g2
