#!/bin/bash
# round 6, call 42: scorer hidden sign pattern and bias gradient of the full-size ddi step per forced slice count
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python scripts/probe_dense_step_bias.py 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | grep "^{" | tee gpurun_out/r06/call42_bias.txt | tail -3
