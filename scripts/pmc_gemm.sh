#!/bin/bash
# MFMA / wait / LDS-conflict counters of the GEMM kernels (one --pmc pass, kernel trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_gemm
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -f csv -d gpurun_out/pmc_gemm -o g -- python3 scripts/bench_gemm.py --shapes collab_fwd,collab_dgrad,collab_wgrad,square4k --iters 3 > gpurun_out/pmc_gemm_bench.log 2>&1
f=$(find gpurun_out/pmc_gemm -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, json, collections, re
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_f32_kernel" not in r["Kernel_Name"]: continue
    key = re.sub(r"\(.*", "", r["Kernel_Name"]) + "|grid=" + r["Grid_Size"]
    per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in per.items():
    med = {n: sorted(v)[len(v) // 2] for n, v in c.items()}
    d = dict(med)
    if med.get("SQ_BUSY_CYCLES"):
        # SQ_BUSY_CYCLES is summed over the SQs (one per CU-pair-group): MFMA busy per SIMD-cycle

        d["wait_any_frac_of_wave_cycles"] = med.get("SQ_WAIT_ANY", 0) / max(med.get("SQ_WAVE_CYCLES", 1), 1)
        d["wait_inst_frac_of_wave_cycles"] = med.get("SQ_WAIT_INST_ANY", 0) / max(med.get("SQ_WAVE_CYCLES", 1), 1)
        d["active_inst_frac_of_wave_cycles"] = med.get("SQ_ACTIVE_INST_ANY", 0) / max(med.get("SQ_WAVE_CYCLES", 1), 1)
    out[k] = d
json.dump(out, open("gpurun_out/pmc_gemm.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
rm -rf gpurun_out/pmc_gemm
