#!/bin/bash
# round 4, call 15: what the side stream's work costs the main stream (per-kernel durations with / without overlap)
O=$GRAFT_REPO_ROOT/gpurun_out/r04c15; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in collab citation2; do
rocprofv3 --kernel-trace -f csv -d $O/prof -o step -- python3 bench.py --workload $w --steps 40 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python scripts/overlap_profile.py $f 30 > $O/overlap_$w.txt
head -3 $f > $O/trace_head_$w.csv
rm -rf $O/prof
cat $O/overlap_$w.txt | cut -c1-150
done
