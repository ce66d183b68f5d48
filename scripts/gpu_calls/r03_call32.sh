#!/bin/bash
# round 3, call 32: chunk pass fused into the main pass's launch: tests, micro-benchmark A/B, step A/B
O=gpurun_out/r03c32; mkdir -p $O
python -m pytest tests -x -q -m gpu -k "aggregate or agg or incidence or full_size or fused or captured" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 4 $O/tests.log
for f in 0 1; do
PLNLP_AGG_FUSED=$f python scripts/bench_agg.py --cases collab --feat 256 --tune 0,128 --hub-order none,65536:256 > $O/agg_fused$f.jsonl 2>/dev/null
PLNLP_AGG_FUSED=$f python scripts/bench_agg.py --cases collab --feat 256 --weighted --tune 0 >> $O/agg_fused$f.jsonl 2>/dev/null
PLNLP_AGG_FUSED=$f python scripts/bench_agg.py --cases ddi --feat 512 --tune 0,16 >> $O/agg_fused$f.jsonl 2>/dev/null
PLNLP_AGG_FUSED=$f python scripts/bench_agg.py --cases citation2 --feat 256 --tune 0 >> $O/agg_fused$f.jsonl 2>/dev/null
done
python - <<'PY'
import json
for f in (0, 1):
    for l in open("gpurun_out/r03c32/agg_fused%d.jsonl" % f):
        r = json.loads(l)
        print("fused", f, r["case"], r["feat"], "tune", r["tune"], "hub", r["hub_order"], "ms", r["ms"])
PY
for i in 1 2 3; do
for f in 0 1; do
PLNLP_AGG_FUSED=$f python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_collab_f${f}_$i.json 2>/dev/null
python -c "
import json,sys; r=json.loads(open('$O/bench_collab_f${f}_$i.json').read().strip().splitlines()[-1]); print('collab fused=$f', r['ms_per_step'], r['value'], r.get('ms_per_step_full_forward'))"
done; done
for f in 0 1; do for w in ddi citation2; do PLNLP_AGG_FUSED=$f python bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w fused=$f', r['ms_per_step'], r['value'])"; done; done
