#!/bin/bash
# wide weight gradient: kernel durations from a trace (wide kernel / its reduce / tile kernel / generic reduce)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05w
rm -rf gpurun_out/r05w/trace
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r05w/trace -o t -- python3 scripts/bench_gemm.py --shapes cit_l2_wgrad,cit_in_wgrad --math wide --iters 5 > gpurun_out/r05w/trace_bench.txt 2>&1
grep shape gpurun_out/r05w/trace_bench.txt | cut -c1-130
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r05w/trace/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "gemm" in n or "wgrad" in n or "reduce" in n:
        d[n[:70] + "|grid=" + str(r.get("Grid_Size", r.get("Grid_Size_X", "?")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v.sort(); print(f"{len(v):4d}  median {v[len(v)//2]:9.1f} us  min {v[0]:9.1f}  {k}")
PY
rm -rf gpurun_out/r05w/trace
