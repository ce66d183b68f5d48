#!/bin/bash
# round 3, call 23: hub chunk pass in XCD-pinned slabs (main pass full width), per-kernel times
O=gpurun_out/r03c23; mkdir -p $O
python scripts/bench_agg.py --cases collab --feat 256 --tune 0,128 --hub-order none,8192:128,16384:128,32768:128,8192:512,4096:512 > $O/agg_hub_sweep.jsonl 2> $O/agg_hub_sweep.err
python scripts/bench_agg.py --cases collab --feat 512 --tune 0,128 --hub-order none,8192:128 >> $O/agg_hub_sweep.jsonl 2>> $O/agg_hub_sweep.err
python scripts/bench_agg.py --cases ddi --feat 256,512 --tune 0,64,128,16,32 >> $O/agg_hub_sweep.jsonl 2>> $O/agg_hub_sweep.err
python scripts/bench_agg.py --cases citation2 --feat 256 --tune 0,128 --hub-order none,65536:1024 >> $O/agg_hub_sweep.jsonl 2>> $O/agg_hub_sweep.err
cut -c1-60,95-260 $O/agg_hub_sweep.jsonl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o agg -- python3 $GRAFT_REPO_ROOT/scripts/bench_agg.py --cases collab --feat 256 --tune 0,128 --hub-order none,8192:128 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 - {} <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# group by kernel name + grid size, in order of first appearance
agg = collections.OrderedDict()
for r in rows:
    k = (r["Kernel_Name"][:70], r.get("Grid_Size") or r.get("Grid_Size_X"))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(k, []).append(d)
for k, v in agg.items():
    if "csr_agg" in k[0]:
        v2 = sorted(v)
        print(k, "n=%d median %.1f us" % (len(v), v2[len(v2) // 2]))
PY
rm -rf $O/prof
