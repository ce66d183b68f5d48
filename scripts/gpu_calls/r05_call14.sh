#!/bin/bash
# round 5, call 14: the round's measurement set on the final tree
bash scripts/refresh_profiles_r05.sh > gpurun_out/r05p_log.txt 2>&1
tail -5 gpurun_out/r05p_log.txt
python - <<PY
import json
for w in ("collab", "ddi", "citation2"):
    r = json.loads(open("gpurun_out/r05p/bench_%s.json" % w).read().strip().splitlines()[-1])
    print(w, round(r["ms_per_step"], 4), "ms", round(r["value"] / 1e6, 2), "M edges/s; f32", r.get("ms_per_step_f32_mfma"), "| roofline", r["roofline"].get("kernel_ms"), r["roofline"].get("frac"), "| cpu", (r.get("cpu_baseline") or {}).get("value"))
PY
