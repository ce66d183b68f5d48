#!/bin/bash
# round 6, call 45: the GPU suite three times in a row, to catch an intermittent fatal error seen once (call 43) with its head
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for i in 1 2 3; do
  SECONDS=0
  timeout 3000 python -X faulthandler -m pytest tests -v -m gpu 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" > $O/call45_run$i.txt
  echo "run $i: ${SECONDS}s: $(grep -c PASSED $O/call45_run$i.txt) passed; $(tail -1 $O/call45_run$i.txt | cut -c1-120)"
  if grep -q "Fatal Python error\|dumped core\|Memory access fault\|Segmentation" $O/call45_run$i.txt; then
    grep -n "PASSED\|FAILED" $O/call45_run$i.txt | tail -2
    grep -n -A45 "Fatal Python error\|Memory access fault" $O/call45_run$i.txt | grep -v "dist-packages/_pytest\|dist-packages/pluggy" | head -80
  fi
done
