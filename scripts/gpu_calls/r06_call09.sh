#!/bin/bash
# round 6, call 9: bias gradient out of the wide weight-gradient kernel -- tests, then same-box A/B of the three steps (on / off)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round6.py tests/test_hip_round5.py tests/test_hip_parity.py -q -m gpu -x -k "bias_gradient or wide_weight or single_step or trajectory or hits50_training or hits20_training" 2>&1 | tail -6 > $O/call09_tests.txt; cat $O/call09_tests.txt
cat > /tmp/ab_colsum.py <<'PY'
import os, sys, json, subprocess
PY
for rep in 1 2 3; do
for w in citation2 collab ddi; do
for mode in 1 0; do
  PLNLP_PROBE_COLSUM_IN_WGRAD=$mode timeout 300 python - <<PY
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
        print("$w colsum_in_wgrad=$mode rep$rep", round(r["ms_per_step"], 4), {k: v for k, v in r.get("kernel_families_per_step", {}).items() if "gemm" in k})
PY
done; done; done > $O/call09_colsum_ab.txt 2>&1
cat $O/call09_colsum_ab.txt
