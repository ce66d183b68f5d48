#!/bin/bash
# round 6, call 46: the first part of the GPU suite (where call 43 died at 98 s) eight times, keeping the head of any fatal error
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do
  SECONDS=0
  timeout 1500 python -X faulthandler -m pytest tests/test_hip_multirank.py tests/test_hip_parity.py tests/test_hip_round2.py -v -m gpu 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" > $O/call46_run$i.txt
  echo "run $i: ${SECONDS}s: $(grep -c PASSED $O/call46_run$i.txt) passed; $(tail -1 $O/call46_run$i.txt | cut -c1-100)"
  if grep -q "Fatal Python error\|dumped core\|Memory access fault\|Segmentation\|Aborted" $O/call46_run$i.txt; then
    grep -n "PASSED\|FAILED" $O/call46_run$i.txt | tail -2
    grep -n -B5 -A40 "Fatal Python error\|Memory access fault" $O/call46_run$i.txt | grep -v "dist-packages/_pytest\|dist-packages/pluggy" | head -70
  fi
done
