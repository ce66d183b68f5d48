#!/bin/bash
# round 5, call 4: tests/test_hip_multirank.py with per-parameter deviation tables (the MLP cases' weights differ from the
# one-process run by more than reassociation after 6 steps: why?) and the padded empty block
O=$GRAFT_REPO_ROOT/gpurun_out/r05c04; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_multirank.py -q -m gpu > $O/multirank.txt 2>&1; grep -v "^E   *File\|^E     " $O/multirank.txt | grep "^E\|FAILED\|passed\|failed" | cut -c1-260 | head -150
