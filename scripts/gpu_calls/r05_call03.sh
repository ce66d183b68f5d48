#!/bin/bash
# round 5, call 3: the world-2 / world-4 test after the empty-result fix; GPU pilot of the wide trained-parity legs
O=$GRAFT_REPO_ROOT/gpurun_out/r05c03; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_multirank.py -q -m gpu > $O/multirank.txt 2>&1; tail -40 $O/multirank.txt | cut -c1-300
timeout 1200 python scripts/pilot_wide_legs.py > $O/pilot.txt 2>&1; grep -v amdgpu.ids $O/pilot.txt | tail -40 | cut -c1-400
