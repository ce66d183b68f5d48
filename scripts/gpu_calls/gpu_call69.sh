#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/c69; mkdir -p $R
rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o full -- python3 bench.py --no-cpu-baseline --no-parity > $R/bench_collab_full_under_rocprof.json 2>/dev/null
f=$(find $R/prof -name "*kernel_stats.csv" | head -1); cp $f $R/kernel_stats_full.csv
f=$(find $R/prof -name "*kernel_trace.csv" | head -1)
python3 - <<'PY' > gpurun_out/c69/roofline_kernel_calls.txt
import json
d=json.loads(open("gpurun_out/c69/bench_collab_full_under_rocprof.json").read().strip().splitlines()[-1])
r=d["roofline"]; print("# bench.py's `roofline` object in this (profiled) run:", r["kernel"], "|", r.get("kernel_form"), "| kernel_ms", round(r["kernel_ms"],3), "achieved GB/s", round(r["achieved"],1), "frac", round(r["frac"],4))
print("# (bench.py times 3 warm-up + 20 launches with device events on the launch stream; the 3 + 3 + 3 launches before them are the autotuner's)")
PY
python3 scripts/kernel_calls.py $f "csr_agg_vec_kernel<1, 32, false" 20 >> $R/roofline_kernel_calls.txt
python3 scripts/kernel_calls.py $f "csr_agg_vec_kernel<2, 64, false" >> $R/roofline_kernel_calls.txt
python3 scripts/kernel_calls.py $f "csr_agg_vec_kernel<1, 64, false" 20 >> $R/roofline_kernel_calls.txt
rm -rf $R/prof
cat $R/roofline_kernel_calls.txt | cut -c1-400
