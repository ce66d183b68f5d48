#!/bin/bash
# round 6, call 10: bias gradient out of the wide weight-gradient kernel, second version (accumulator outside Split): same-box A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_round6.py -q -m gpu -x -k "bias_gradient" 2>&1 | tail -3
for rep in 1 2 3; do
for w in collab citation2; do
for mode in 1 0; do
  PLNLP_PROBE_COLSUM_IN_WGRAD=$mode timeout 300 python - <<PY 2>/dev/null
import os, sys, json, io, contextlib
sys.argv = ["bench.py", "--workload", "$w", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-parity", "--no-stress", "--no-roofline"]
sys.path.insert(0, os.getcwd())
import plnlp_amd as P
P.ops.COLSUM_IN_WGRAD["enabled"] = os.environ["PLNLP_PROBE_COLSUM_IN_WGRAD"] == "1"
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
for l in buf.getvalue().splitlines():
    if l.startswith("{"):
        r = json.loads(l)
        print("$w colsum_in_wgrad=$mode rep$rep", round(r["ms_per_step"], 4))
PY
done; done; done > $O/call10_colsum_ab.txt 2>&1
cat $O/call10_colsum_ab.txt
