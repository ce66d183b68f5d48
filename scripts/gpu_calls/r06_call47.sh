#!/bin/bash
# round 6, call 47: the next part of the GPU suite (rounds 3 and 4: where call 43 must have been at 98 s) five times, keeping the head of any fatal error
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for i in 1 2 3 4 5; do
  SECONDS=0
  timeout 1500 python -X faulthandler -m pytest tests/test_hip_round3.py tests/test_hip_round4.py -v -m gpu 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" > $O/call47_run$i.txt
  echo "run $i: ${SECONDS}s: $(grep -c PASSED $O/call47_run$i.txt) passed; $(tail -1 $O/call47_run$i.txt | cut -c1-100)"
  if grep -q "Fatal Python error\|dumped core\|Memory access fault\|Segmentation\|Aborted" $O/call47_run$i.txt; then
    grep -n "PASSED\|FAILED" $O/call47_run$i.txt | tail -2
    grep -n -B5 -A40 "Fatal Python error\|Memory access fault" $O/call47_run$i.txt | grep -v "dist-packages/_pytest\|dist-packages/pluggy" | head -70
  fi
done
