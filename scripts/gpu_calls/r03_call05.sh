#!/bin/bash
mkdir -p gpurun_out/r03c05
python scripts/probe_gemm_shapes.py > gpurun_out/r03c05/gemm_shapes.txt 2>&1
tail -n 40 gpurun_out/r03c05/gemm_shapes.txt
