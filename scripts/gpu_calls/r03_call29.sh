#!/bin/bash
# round 3, call 29: index prefetch in the aggregation's batch loop: aggregation tests, micro-benchmarks, step A/B
O=gpurun_out/r03c29; mkdir -p $O
python -m pytest tests -x -q -m gpu -k "aggregate or agg or incidence or spmm or full_size" > $O/agg_tests.log 2>&1; echo "rc=$?" >> $O/agg_tests.log; tail -n 4 $O/agg_tests.log
python scripts/bench_agg.py --cases collab --feat 256 --tune 0,128 --hub-order none,65536:256,32768:256,32768:512,65536:512 > $O/agg.jsonl 2>/dev/null
python scripts/bench_agg.py --cases ddi --feat 512 --tune 0,16 >> $O/agg.jsonl 2>/dev/null
python scripts/bench_agg.py --cases citation2,uniform_big --feat 256 --tune 0 >> $O/agg.jsonl 2>/dev/null
python scripts/bench_agg.py --cases collab --feat 256 --weighted --tune 0,128 --hub-order none,65536:256 >> $O/agg.jsonl 2>/dev/null
python - <<'PY'
import json
for l in open("gpurun_out/r03c29/agg.jsonl"):
    r = json.loads(l)
    print(r["case"], r["feat"], "tune", r["tune"], "hub", r["hub_order"], "chunks", r["chunks"], "ms", r["ms"])
PY
for i in 1 2; do
PLNLP_AGG_AUTOTUNE=0 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_collab_notune$i.json 2>/dev/null
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_collab_tuned$i.json 2>/dev/null
done
for f in notune1 tuned1 notune2 tuned2; do python -c "
import json,sys; r=json.loads(open('$O/bench_collab_$f.json').read().strip().splitlines()[-1]); print('$f', r['ms_per_step'], r['value'], r.get('ms_per_step_full_forward'))"; done
for w in ddi citation2; do python bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', r['ms_per_step'], r['value'])"; done
rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o agg -- python3 scripts/bench_agg.py --cases collab --feat 256 --tune 0,128 --hub-order none,65536:256,32768:256 > /dev/null 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python scripts/kernel_groups.py $f csr_agg > $O/kernel_groups.txt; cat $O/kernel_groups.txt | cut -c1-60,90-200
rm -rf $O/prof
