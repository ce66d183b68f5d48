#!/bin/bash
R=gpurun_out/r02i; mkdir -p $R
for i in 1 2; do
python bench.py > $R/bench_collab_$i.json 2> $R/bench_collab.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r02i/bench_collab_$i.json").read().strip().splitlines()[-1])
print("full run $i:", round(d["ms_per_step"],3), round(d["value"]/1e6,2), round(d["roofline"]["frac"],3), round(d["roofline_mfma"]["frac"],3), d["host_enqueue_ms_per_step"])
PY
python bench.py --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('short run:', round(d['ms_per_step'],3), d['host_enqueue_ms_per_step'])"
done
rocm-smi --showclocks --showtemp --showpower 2>/dev/null | head -30
