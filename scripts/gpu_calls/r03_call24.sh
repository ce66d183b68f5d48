#!/bin/bash
# round 3, call 24: hub chunk pass in XCD-pinned slabs: range / length / threshold sweep + per-kernel times
O=gpurun_out/r03c24; mkdir -p $O
python scripts/bench_agg.py --cases collab --feat 256 --tune 0,128 --hub-order none,32768:128,32768:256,65536:128,65536:256,131072:128,24576:128,32768:64 > $O/agg_hub_sweep.jsonl 2> $O/agg_hub_sweep.err
python scripts/bench_agg.py --cases collab --feat 256 --tune 128 --threshold 64,256,512 --hub-order none,32768:128 >> $O/agg_hub_sweep.jsonl 2>> $O/agg_hub_sweep.err
python - <<'PY'
import json
for l in open("gpurun_out/r03c24/agg_hub_sweep.jsonl"):
    r = json.loads(l)
    print(r["case"], r["feat"], "tune", r["tune"], "hub", r["hub_order"], "thr", r["threshold"], "chunks", r["chunks"], "ms", r["ms"])
PY
rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o agg -- python3 scripts/bench_agg.py --cases collab --feat 256 --tune 0,128 --hub-order none,32768:128 > /dev/null 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python scripts/kernel_groups.py $f csr_agg > $O/kernel_groups.txt; cat $O/kernel_groups.txt | cut -c1-60,90-200
rm -rf $O/prof
