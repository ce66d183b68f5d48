#!/bin/bash
# round 6, call 11: the bench lines after the record changes (default = collab; ddi; citation2), wall time of the default run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
/usr/bin/time -v python bench.py > $O/call11_bench_collab.json 2> $O/call11_bench_collab.err; grep -E "Elapsed|Maximum resident" $O/call11_bench_collab.err; tail -c 600 $O/call11_bench_collab.json; echo
python bench.py --workload ddi --steps 10 --warmup 5 --no-parity --no-stress --cpu-steps 1 > $O/call11_bench_ddi.json 2> $O/call11_bench_ddi.err; tail -3 $O/call11_bench_ddi.err; tail -c 300 $O/call11_bench_ddi.json; echo
python bench.py --workload citation2 --steps 10 --warmup 5 --no-parity --no-stress --cpu-steps 1 > $O/call11_bench_citation2.json 2> $O/call11_bench_citation2.err; tail -3 $O/call11_bench_citation2.err; tail -c 300 $O/call11_bench_citation2.json; echo
python - <<PY
import json
for w in ("collab", "ddi", "citation2"):
    try:
        r = json.loads([l for l in open("$O/call11_bench_%s.json" % w) if l.startswith("{")][-1])
    except Exception as e:
        print(w, "no line", e); continue
    print(w, "ms/step", round(r["ms_per_step"], 4), "value", round(r["value"] / 1e6, 2), "M edges/s")
    for k in ("roofline", "roofline_workload_agg", "roofline_mfma", "roofline_agg_adam"):
        if k in r:
            o = r[k]
            print("   ", k, "ms", round(o.get("kernel_ms", 0), 4), "frac", o.get("frac"), "|", str(o.get("subject", ""))[:90])
    if "hits50_parity" in r:
        h = r["hits50_parity"]
        print("    hits50_parity", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in h.items() if k in ("gpu_valid", "gpu_test", "cpu_valid", "cpu_test", "cpu64_valid", "cpu64_test", "max_abs_diff_points", "cpu32_vs_f64_points", "seconds")})
    if "cpu_baseline" in r: print("    cpu_baseline", r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"])
PY
