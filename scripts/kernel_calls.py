"""Per-call durations of one kernel in a rocprofv3 kernel-trace csv, grouped by launch grid (the same template can serve
several problems in one run: bench.py's roofline graph and its small parity-run graph)."""
import collections, csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
groups = collections.OrderedDict()
for r in rows:
    key = r.get('Grid_Size', r.get('Grid_Size_X', '?'))
    groups.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
print(f"{len(rows)} calls of *{sys.argv[2]}* in {len(groups)} launch grid(s); ms per call in launch order, per grid:")
for key, d in groups.items():
    d20 = d[-20:]
    print(f"grid {key}: {len(d)} calls: " + " ".join(f"{x:.2f}" for x in d))
    print(f"   average of all {len(d)}: {sum(d) / len(d):.3f} ms; average of the last {len(d20)}: {sum(d20) / len(d20):.3f} ms")
