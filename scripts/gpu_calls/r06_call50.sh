#!/bin/bash
# round 6, call 50: the round's last tree: the measurement set (scripts/refresh_profiles_r06.sh), then the whole GPU suite + smoke
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/refresh_profiles_r06.sh > gpurun_out/refresh_r06_final.log 2>&1
O=gpurun_out/r06; mkdir -p $O
SECONDS=0
timeout 3000 python -X faulthandler -m pytest tests -q -m gpu 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" > $O/call50_suite_full.txt
tail -3 $O/call50_suite_full.txt > $O/call50_suite.txt
echo "suite: ${SECONDS}s" >> $O/call50_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -2 >> $O/call50_suite.txt
cat $O/call50_suite.txt
