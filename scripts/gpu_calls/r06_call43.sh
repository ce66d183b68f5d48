#!/bin/bash
# round 6, call 43: the whole GPU suite on the current tree (no -x: every failure listed) + smoke
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
SECONDS=0
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -25 > $O/call43_suite.txt
echo "suite: ${SECONDS}s" >> $O/call43_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -2 >> $O/call43_suite.txt
cat $O/call43_suite.txt
