"""For each main-stream kernel of the steady-state steps in a rocprofv3 kernel trace: its duration and which
side-stream kernels ran concurrently.  python scripts/overlap_profile.py <kernel_trace.csv> [n_steps]"""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
qcol = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
queues = collections.Counter(r[qcol] for r in rows)
main_q = max(queues, key=lambda q: sum(r["e"] - r["s"] for r in rows if r[qcol] == q))
marks = [i for i, r in enumerate(rows) if "pairwise_loss_kernel" in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lo, hi = marks[-n - 1], marks[-1]
t0 = rows[lo]["s"]
side = [r for r in rows[lo:hi] if r[qcol] != main_q]
for r in rows[lo:hi]:
    if r[qcol] != main_q:
        continue
    ov = [(x["Kernel_Name"].split("(")[0][-40:], min(x["e"], r["e"]) - max(x["s"], r["s"])) for x in side
          if x["s"] < r["e"] and x["e"] > r["s"]]
    ovs = " | ".join("%s %.0fus" % (k, d / 1e3) for k, d in ov)
    print("%9.1f %7.1f  %-50s %s" % ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, r["Kernel_Name"].split("(")[0][-50:], ovs))
