#!/bin/bash
# collab step: what the main stream waits for between the forward's last kernel and the loss (all streams, two steps)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r05q; mkdir -p $R
rocprofv3 --kernel-trace -f csv -d $R/prof -o step -- python3 bench.py --workload collab --steps 12 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > $R/gap_bench.json 2>/dev/null
f=$(find $R/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
loss = [i for i, r in enumerate(rows) if 'pairwise_loss_kernel' in r['Kernel_Name']]
a, b = loss[-4], loss[-2]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b + 3]:
    q = r.get('Stream_Id', r.get('Queue_Id', '?'))
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f}  s{q:>3s}  {r['Kernel_Name'][:90]}")
PY
rm -rf $R/prof
python3 -c "
import json; d=json.loads(open('$R/gap_bench.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('ms_per_step','host_enqueue_ms_per_step','host_busy_ms_per_step')})"
