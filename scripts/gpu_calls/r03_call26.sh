#!/bin/bash
# round 3, call 26: hub form on the unmapped launches only: step A/B, whole suite
O=gpurun_out/r03c26; mkdir -p $O
for i in 1 2 3; do
PLNLP_AGG_AUTOTUNE=0 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_collab_notune$i.json 2>/dev/null
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_collab_tuned$i.json 2>/dev/null
done
for f in notune1 tuned1 notune2 tuned2 notune3 tuned3; do python -c "
import json,sys; r=json.loads(open('$O/bench_collab_$f.json').read().strip().splitlines()[-1]); print('$f', r['ms_per_step'], r['value'], r.get('ms_per_step_full_forward'))"; done
python scripts/bench_step_launches.py > $O/step_launches.json 2>/dev/null; cat $O/step_launches.json | cut -c1-1500
python -m pytest tests -x -q -m gpu --durations=6 > $O/suite.log 2>&1; echo "rc=$?" >> $O/suite.log
tail -n 12 $O/suite.log
cp gpurun_out/trained_parity_table.txt $O/ 2>/dev/null
