"""Post-process rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv output of scripts/bench_agg.py into
per-launch HBM traffic of the aggregation kernels (MI355X_MICROARCH.md, HBM section: counters are in
KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads -> doubled)."""
import csv, json, sys, collections, re

def load(path, counter):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            per[(re.sub(r"\(.*", "", r["Kernel_Name"]), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return per

fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
out = {}
for key in fetch:
    name, grid = key
    if "csr_agg" not in name:
        continue
    f = sorted(fetch[key])[len(fetch[key]) // 2]
    w = sorted(write.get(key, [0.0]))[len(write.get(key, [0.0])) // 2]
    out[f"{name}|grid={grid}"] = {"fetch_KiB_raw": f, "write_KiB": w,
                                  "hbm_bytes_corrected": (2 * f + w) * 1024, "launches": len(fetch[key])}
json.dump(out, sys.stdout, indent=1)
