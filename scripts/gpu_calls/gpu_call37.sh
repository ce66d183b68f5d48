#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/c37; mkdir -p $R
rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o p -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-stress --no-roofline > $R/bench_collab_under_rocprof.json 2>/dev/null
f=$(find $R/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 10 45 sequence > $R/step_breakdown_collab.txt
rm -rf $R/prof
head -24 $R/step_breakdown_collab.txt | cut -c1-150
timeout 1800 python -m pytest tests/test_hip_parity.py tests/test_hip_round2.py -q -m gpu --deselect tests/test_hip_parity.py::test_hits20_training_parity_ddi_recipe --deselect tests/test_hip_round2.py::test_hits20_ddi_recipe_parity_over_seeds 2>&1 | tail -6 > $R/all.log
cat $R/all.log
python __graft_entry__.py smoke 2>&1 | tail -2
