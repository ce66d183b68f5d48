#!/bin/bash
mkdir -p gpurun_out/c40
for w in collab ddi citation2; do
python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>gpurun_out/c40/err_$w.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['ms_per_step'], 'host enqueue ms/step', d['host_enqueue_ms_per_step'])"
tail -2 gpurun_out/c40/err_$w.txt
done
timeout 1800 python -m pytest tests/test_hip_parity.py tests/test_hip_round2.py -q -m gpu --deselect tests/test_hip_parity.py::test_hits20_training_parity_ddi_recipe --deselect tests/test_hip_round2.py::test_hits20_ddi_recipe_parity_over_seeds 2>&1 | tail -6
