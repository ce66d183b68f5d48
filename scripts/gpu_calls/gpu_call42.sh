#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/c42; mkdir -p $R
export PLNLP_PROLOGUE_OVERLAP=0
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no side stream:', d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'])"
rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o p -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
f=$(find $R/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 10 45 sequence > $R/step_breakdown_collab_nooverlap.txt
rm -rf $R/prof
grep -A300 "launch sequence" $R/step_breakdown_collab_nooverlap.txt | awk '$1+0>=900' | cut -c1-120 | head -70
