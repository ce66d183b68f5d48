#!/bin/bash
# the default bench command (with the HBM-roofline stress graph) under rocprofv3: kernel stats for the `roofline` kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/c68; mkdir -p $R
rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o full -- python3 bench.py --no-cpu-baseline --no-parity > $R/bench_collab_full_under_rocprof.json 2>/dev/null
f=$(find $R/prof -name "*kernel_stats.csv" | head -1); cp $f $R/kernel_stats_full.csv
rm -rf $R/prof
python3 - <<'PY'
import json, csv
d=json.loads(open("gpurun_out/c68/bench_collab_full_under_rocprof.json").read().strip().splitlines()[-1])
r=d["roofline"]; print("bench roofline:", r["kernel"], r.get("kernel_form"), "kernel_ms", round(r["kernel_ms"],3), "frac", round(r["frac"],3))
for row in csv.DictReader(open("gpurun_out/c68/kernel_stats_full.csv")):
    if "csr_agg" in row["Name"]:
        print(row["Name"][:70], "calls", row["Calls"], "avg us", round(float(row["AverageNs"])/1e3,1), "max us", round(float(row["MaxNs"])/1e3,1))
PY
