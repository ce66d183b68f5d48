#!/bin/bash
# round 6, call 38: signed error of the dense aggregation against float64 per forced slice count
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for s in 4 3 5 7 6 8 2; do
python - <<PY 2>/dev/null | grep '"dense": true' | sed "s/^/slices=$s /" | cut -c1-200
import runpy
from plnlp_amd import _lib
_lib.load().plnlp_dense_aggregate_tuning($s)
runpy.run_path("scripts/probe_dense_agg_error.py", run_name="__main__")
PY
done | tee gpurun_out/r06/call38_signed_error.txt
