#!/bin/bash
# round 3, call 40: the aggregation form tuned in one-process training too: step A/B, tests
O=gpurun_out/r03c40; mkdir -p $O
for i in 1 2 3; do for t in 0 1; do
PLNLP_AGG_AUTOTUNE=$t python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_collab_t${t}_$i.json 2>/dev/null
python -c "
import json; r=json.loads(open('$O/bench_collab_t${t}_$i.json').read().strip().splitlines()[-1]); print('collab tune=$t', r['ms_per_step'], r['value'], r['train_epoch']['value'])"
done; done
for t in 0 1; do for w in ddi citation2; do PLNLP_AGG_AUTOTUNE=$t python bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w tune=$t', r['ms_per_step'], r['value'])"; done; done
python -m pytest tests -x -q -m gpu --deselect tests/test_hip_round3.py::test_trained_regime_hits_parity_over_seeds > $O/suite.log 2>&1; echo "rc=$?" >> $O/suite.log; grep -n "passed\|failed\|FAILED" $O/suite.log | tail -4
