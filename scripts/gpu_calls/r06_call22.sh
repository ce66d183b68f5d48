#!/bin/bash
# round 6, call 22: the test groups around the GCN input conv / fused Adam / capture, with their full tails
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_round2.py tests/test_hip_round4.py tests/test_hip_round5.py tests/test_hip_parity.py -q -m gpu -x -k "citation or gcn or adam or capture or graph" 2>&1 | tail -30
