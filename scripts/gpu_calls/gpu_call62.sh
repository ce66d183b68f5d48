#!/bin/bash
R=gpurun_out/r02i; mkdir -p $R
s=$(date +%s.%N); python bench.py > $R/bench_collab.json 2> $R/bench_collab.err; e=$(date +%s.%N)
echo "default bench.py wall seconds: $(python -c "print(round($e-$s,1))")" | tee $R/bench_collab_wall.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02i/bench_collab.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],3), round(d["value"]/1e6,2), d["roofline"]["frac"], d["roofline_mfma"]["frac"], d["cpu_baseline"]["value"], d["hits50_parity"])
PY
