"""Summarise a rocprofv3 kernel-trace csv of bench.py: steady-state wall vs busy time and the
per-step cost of every kernel (steps are delimited by the pairwise-loss kernel)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
loss_idx = [i for i, r in enumerate(rows) if 'pairwise_loss_kernel' in r['Kernel_Name']]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
a, b = loss_idx[-nsteps - 1], loss_idx[-1]
win = rows[a:b]
wall = int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in win)
print(f"{nsteps} steady-state steps: wall {wall/1e6/nsteps:.3f} ms/step, kernel-busy {busy/1e6/nsteps:.3f} ms/step, "
      f"{len(win)/nsteps:.0f} launches/step")
agg = collections.Counter(); cnt = collections.Counter()
for r in win:
    n = r['Kernel_Name'][:96]
    agg[n] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); cnt[n] += 1
print(f"{'us/step':>9s} {'calls/step':>10s}  kernel")
for n, t in agg.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
    print(f"{t/1e3/nsteps:9.1f} {cnt[n]/nsteps:10.1f}  {n}")

if len(sys.argv) > 4 and sys.argv[4] == "sequence":      # every launch of the last full step, in order
    a, b = loss_idx[-2], loss_idx[-1]
    t0 = int(rows[a]['Start_Timestamp'])
    print("\nlaunch sequence of one step (start us, duration us, kernel):")
    for r in rows[a:b]:
        q = r.get('Stream_Id', r.get('Queue_Id', '?'))
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f}  s{q:>3s}  {r['Kernel_Name'][:86]}")
