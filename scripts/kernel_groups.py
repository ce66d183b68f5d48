"""Median duration of every (kernel, grid) group of a rocprofv3 kernel trace, in order of first appearance.
python scripts/kernel_groups.py <kernel_trace.csv> [name substring]"""
import collections, csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2] if len(sys.argv) > 2 else ""
groups = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"]
    if want not in name:
        continue
    grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
    groups.setdefault((name[:90], grid), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), v in groups.items():
    v = sorted(v)
    print("%-92s grid %-9s n=%-4d median %8.1f us  min %8.1f us" % (name, grid, len(v), v[len(v) // 2], v[0]))
