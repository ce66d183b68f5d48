#!/bin/bash
# round 6, call 30: experiment -- the collab step's aggregation launches in eight XCD-pinned column slabs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python scripts/probe_xcd_slabs_step.py 2>&1 | tail -6 | tee gpurun_out/r06/call30_xcd_slabs.txt
