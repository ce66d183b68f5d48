#!/bin/bash
mkdir -p gpurun_out/c32
{
echo "== default stores"; python scripts/bench_agg.py --cases collab,uniform_big --feat 256,512 --tune 0 2>/dev/null
echo "== non-temporal stores"; PLNLP_HIP_LIB=$PWD/plnlp_amd/build/abl/libplnlp_hip_aggnt.so python scripts/bench_agg.py --cases collab,uniform_big --feat 256,512 --tune 0 2>/dev/null
echo "== bench collab default"; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
echo "== bench collab NT stores"; PLNLP_HIP_LIB=$PWD/plnlp_amd/build/abl/libplnlp_hip_aggnt.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
} > gpurun_out/c32/agg_nt.txt 2>&1
cut -c1-260 gpurun_out/c32/agg_nt.txt
