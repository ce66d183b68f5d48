"""From a rocprofv3 kernel-trace csv of bench.py: how long each main-stream kernel of the training step runs when a
side-stream kernel (the next batch's index preparation) executes next to it, and when none does.
usage: overlap_profile.py <kernel_trace.csv> [n_steps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
loss_idx = [i for i, r in enumerate(rows) if "pairwise_loss_kernel" in r["Kernel_Name"]]
a, b = loss_idx[-nsteps - 1], loss_idx[-1]
win = rows[a:b]
key = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
by_stream = collections.Counter(r[key] for r in win)
main = by_stream.most_common(1)[0][0]            # the stream with the most launches in the window is not the side stream ...
busy = collections.Counter()
for r in win:
    busy[r[key]] += r["e"] - r["s"]
main = busy.most_common(1)[0][0]                 # ... the one with the most kernel time is the main one
side = [r for r in win if r[key] != main]
print(f"{nsteps} steps; streams by kernel time (us/step): " + ", ".join(f"{k}: {v / 1e3 / nsteps:.0f}" for k, v in busy.most_common()))
stat = collections.defaultdict(lambda: [0, 0, 0, 0, 0])       # n_alone, t_alone, n_overlapped, t_overlapped, t_overlap
for r in win:
    if r[key] != main:
        continue
    ov = sum(max(0, min(r["e"], q["e"]) - max(r["s"], q["s"])) for q in side)
    st = stat[r["Kernel_Name"][:70]]
    if ov > 0:
        st[2] += 1; st[3] += r["e"] - r["s"]; st[4] += ov
    else:
        st[0] += 1; st[1] += r["e"] - r["s"]
print(f"{'alone: n':>9s} {'us':>8s} {'with side work: n':>18s} {'us':>8s} {'overlap us':>10s}  kernel")
tot_excess = 0.0
for n, (n0, t0, n1, t1, ov) in sorted(stat.items(), key=lambda kv: -(kv[1][1] + kv[1][3])):
    m0 = t0 / n0 / 1e3 if n0 else float("nan")
    m1 = t1 / n1 / 1e3 if n1 else float("nan")
    if n0 and n1:
        tot_excess += (m1 - m0) * n1
    print(f"{n0:9d} {m0:8.1f} {n1:18d} {m1:8.1f} {ov / max(n1, 1) / 1e3:10.1f}  {n}")
print(f"main-stream time attributable to running next to side work: {tot_excess / nsteps:.1f} us/step")
# idle gaps on the main stream
mrows = [r for r in win if r[key] == main]
gaps = sum(max(0, q["s"] - p["e"]) for p, q in zip(mrows, mrows[1:]))
print(f"main-stream gaps between kernels: {gaps / 1e3 / nsteps:.1f} us/step")
