#!/bin/bash
# round 6, call 36: the full-size ddi step test alone, with its message; again with four K slices forced in the dense aggregation
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest "tests/test_hip_round4.py::test_full_size_ddi_step_matches_the_oracle" -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -40
