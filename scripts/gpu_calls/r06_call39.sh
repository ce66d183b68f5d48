#!/bin/bash
# round 6, call 39: column-coherent error of the dense aggregation per forced slice count
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python scripts/probe_dense_agg_columns.py 2>/dev/null | tee gpurun_out/r06/call39_columns.txt
