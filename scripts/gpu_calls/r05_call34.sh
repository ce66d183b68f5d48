#!/bin/bash
# citation2's committed line + step breakdown on the final tree (table padded to 64 columns)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r05q; mkdir -p $R
python bench.py --workload citation2 --steps 10 --warmup 5 --no-parity --no-stress --cpu-steps 1 > $R/bench_citation2.json 2>/dev/null
rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o step -- python3 bench.py --workload citation2 --steps 12 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
f=$(find $R/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 6 45 sequence > $R/step_breakdown_citation2.txt
rm -rf $R/prof
head -14 $R/step_breakdown_citation2.txt
python -c "
import json; d=json.loads(open('$R/bench_citation2.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('ms_per_step_f32_mfma'))"
