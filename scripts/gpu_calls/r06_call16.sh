#!/bin/bash
mkdir -p gpurun_out/r06
timeout 600 python scripts/probe_dense_agg_error.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/call16_dense_err.txt; cat gpurun_out/r06/call16_dense_err.txt
