#!/bin/bash
# round 6, call 40: the full-size ddi step's gradient errors (every parameter) per forced slice count of the dense aggregation, and on the CSR kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
sed 's/^        assert err <= max(rel \* scale, 4 \* yard) + floor, (key, err, yard, scale)/        print("GRADERR", key, "err", err, "bound", max(rel * scale, 4 * yard) + floor, "yard", yard, "scale", scale)/' tests/test_hip_round4.py > tests/_tmp_round4_print.py
for s in 4 3 5 7 6 8; do
python - <<PY 2>&1 | grep "GRADERR" | sed "s/^/slices=$s /"
import sys, pytest
from plnlp_amd import _lib
_lib.load().plnlp_dense_aggregate_tuning($s)
sys.exit(pytest.main(["tests/_tmp_round4_print.py::test_full_size_ddi_step_matches_the_oracle", "-q", "-s", "-m", "gpu", "-k", "bf16x3"]))
PY
done | tee gpurun_out/r06/call40_graderr.txt
PLNLP_DENSE_AGG=0 python -m pytest "tests/_tmp_round4_print.py::test_full_size_ddi_step_matches_the_oracle" -q -s -m gpu -k bf16x3 2>&1 | grep GRADERR | sed "s/^/csr /" | tee -a gpurun_out/r06/call40_graderr.txt
rm -f tests/_tmp_round4_print.py
