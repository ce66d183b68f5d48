#!/bin/bash
# round 4, call 11: mutation ladder of the trained-regime harness; the non-trained round-4 tests on the final kernel; benches
O=$GRAFT_REPO_ROOT/gpurun_out/r04c11; mkdir -p $O
timeout 900 python scripts/parity_sensitivity.py 32 > $O/parity_sensitivity.txt 2> $O/parity_sensitivity.err; cat $O/parity_sensitivity.txt; tail -2 $O/parity_sensitivity.err
timeout 1800 python -m pytest tests/test_hip_round4.py -q -k "not trained" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -5 $O/tests.log
for w in collab ddi citation2; do
  timeout 900 python bench.py --workload $w --steps 20 --warmup 6 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_$w.json 2> $O/bench_$w.err
  python -c "
import json; r = json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', r['ms_per_step'], 'ms', r['value'] / 1e6, 'M edges/s', 'f32_mfma', r.get('ms_per_step_f32_mfma'), r.get('value_f32_mfma'))
"
done
