#!/bin/bash
# round 6, call 37: which change moved the full-size ddi step's scorer-bias gradient: the dense aggregation's slice count (forced 4 / 3 / 5 / 7), or not the dense form at all
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for s in 4 3 5 7; do
python - <<PY 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | grep -E "passed|failed|AssertionError" | sed "s/^/slices=$s: /"
import sys, pytest
from plnlp_amd import _lib
_lib.load().plnlp_dense_aggregate_tuning($s)
sys.exit(pytest.main(["tests/test_hip_round4.py::test_full_size_ddi_step_matches_the_oracle", "-q", "-m", "gpu", "-x", "-k", "bf16x3"]))
PY
done
PLNLP_DENSE_AGG=0 python -m pytest "tests/test_hip_round4.py::test_full_size_ddi_step_matches_the_oracle" -q -m gpu -x -k bf16x3 2>&1 | grep -E "passed|failed|AssertionError" | sed "s/^/csr: /"
