#!/bin/bash
# round 6, call 41: the ddi encoder's output per forced slice count against the CSR kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python scripts/probe_dense_step_forward.py 2>/dev/null | tee gpurun_out/r06/call41_forward.txt
