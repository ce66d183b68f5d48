"""Per-call durations of one kernel in a rocprofv3 kernel-trace csv, in launch order.
usage: kernel_calls.py <kernel_trace.csv> <kernel name substring> [<last N calls to average>]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
n = int(sys.argv[3]) if len(sys.argv) > 3 else len(d)
print(f"{len(d)} calls of *{sys.argv[2]}*; ms per call in launch order:")
print(" ".join(f"{x:.2f}" for x in d))
print(f"average of all {len(d)}: {sum(d) / max(len(d), 1):.3f} ms; average of the last {n}: {sum(d[-n:]) / max(len(d[-n:]), 1):.3f} ms")
