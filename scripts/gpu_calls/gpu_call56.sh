#!/bin/bash
mkdir -p gpurun_out/c56
for w in collab ddi citation2; do
python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(d['ms_per_step'],4))"
done
timeout 1800 python -m pytest tests/test_hip_parity.py tests/test_hip_round2.py -q -m gpu --deselect tests/test_hip_parity.py::test_hits20_training_parity_ddi_recipe --deselect tests/test_hip_round2.py::test_hits20_ddi_recipe_parity_over_seeds > gpurun_out/c56/all.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/c56/all.log | tail -8
