#!/bin/bash
# exactly what the driver runs at round end
s=$(date +%s); python -m pytest tests/ -x -q -m gpu 2>&1 | tail -3; e=$(date +%s); echo "gpu suite seconds: $((e-s))"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
s=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/bench_final.json 2>/dev/null; e=$(date +%s); echo "bench seconds: $((e-s))"; python -c "
import json; d=json.loads(open('gpurun_out/bench_final.json').read().strip().splitlines()[-1]); print(d['metric'], d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
