#!/bin/bash
# round 6, call 44: where the GPU suite dies (verbose names; the head of the fatal error)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 3000 python -m pytest tests -v -m gpu -x 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" > $O/call44_full.txt
grep -n "Fatal\|fault\|Segmentation\|Aborted\|HSA\|error" $O/call44_full.txt | head -20
grep -n "PASSED\|FAILED" $O/call44_full.txt | tail -4
grep -n -B3 -A30 "Fatal Python error" $O/call44_full.txt | grep -v "site-packages\|dist-packages" | head -70
