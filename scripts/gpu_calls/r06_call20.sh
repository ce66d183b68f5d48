#!/bin/bash
# round 6, call 20: gate / accumulate epilogues back on the panel kernel, ROWDOT head at 224-column tiles, 1-column colsum with 32
# loads in flight, indexed weight copies in the GCN input conv: tests, citation2 / ddi steps x 2, citation2's trace, and on which
# STREAM the step's fillBufferAligned launches run (collab, ddi)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_round6.py -q -m gpu -x -k "block_kernel or one_column or bias" 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_hip_round4.py tests/test_hip_round5.py tests/test_hip_parity.py -q -m gpu -x -k "citation or gcn or head or rowdot or colsum" 2>&1 | tail -4
for rep in 1 2; do
  for w in citation2 ddi; do
  python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$w rep$rep', round(r['ms_per_step'], 4), 'host_busy', round(r['host_busy_ms_per_step'], 3))"
  done
done | tee $O/call20_steps.txt
for w in citation2 ddi collab; do
  rocprofv3 --kernel-trace --stats -f csv -d $O/prof20 -o step -- python3 bench.py --workload $w --steps 12 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
  f=$(find $O/prof20 -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 6 45 sequence > $O/call20_step_breakdown_$w.txt
  python - $f > $O/call20_fills_$w.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
loss = [i for i, r in enumerate(rows) if 'pairwise_loss_kernel' in r['Kernel_Name']]
a, b = loss[-7], loss[-1]
per = collections.defaultdict(lambda: [0, 0.0])
main = collections.Counter(r.get('Stream_Id', r.get('Queue_Id')) for r in rows[a:b]).most_common(1)[0][0]
for r in rows[a:b]:
    if 'fillBuffer' in r['Kernel_Name'] or 'copyBuffer' in r['Kernel_Name']:
        k = (r['Kernel_Name'][:40], r.get('Stream_Id', r.get('Queue_Id')), r.get('Grid_Size', r.get('Grid_Size_X', '?')))
        per[k][0] += 1; per[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print("main stream id:", main, " (6 steady-state steps)")
for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:42s} stream {k[1]:>3s} grid {k[2]:>10s}  {n / 6:5.2f} calls/step {t / 6:8.1f} us/step")
PY
  rm -rf $O/prof20
done
cat $O/call20_fills_*.txt
head -45 $O/call20_step_breakdown_citation2.txt | cut -c1-130
