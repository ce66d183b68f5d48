#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r02i; mkdir -p $R
/usr/bin/time -v python bench.py > $R/bench_collab.json 2> $R/bench_collab.err; grep -E "Elapsed|Maximum resident" $R/bench_collab.err
rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o collab -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-stress > $R/bench_collab_under_rocprof.json 2>/dev/null
f=$(find $R/prof -name "*kernel_stats.csv" | head -1); cp $f $R/kernel_stats_collab.csv
f=$(find $R/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 10 45 sequence > $R/step_breakdown_collab.txt
rm -rf $R/prof
for w in ddi citation2; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-parity --no-stress --cpu-steps 1 > $R/bench_$w.json 2>/dev/null
  rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o p -- python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
  f=$(find $R/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 5 40 > $R/step_breakdown_$w.txt
  rm -rf $R/prof
done
python bench.py --force-dist --no-cpu-baseline --no-parity --no-stress --no-roofline > $R/bench_collab_shard_1rank.json 2>/dev/null
python bench.py --workload rmat --scale 0.25 --steps 3 --warmup 1 > $R/bench_rmat_s025.json 2>/dev/null
python scripts/bench_agg.py --cases collab,uniform,uniform_big,ddi --feat 256,512 --tune 0,16,32 > $R/agg_microbench.jsonl 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02i/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], round(d["ms_per_step"],3), round(d["value"]/1e6,2), {k:round(v,3) for k,v in d.get("roofline",{}).items() if k in ("achieved","frac")})
PY
