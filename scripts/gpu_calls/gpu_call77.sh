#!/bin/bash
s=$(date +%s); python -m pytest tests/ -x -q -m gpu > gpurun_out/gpu_suite.log 2>&1; echo "pytest rc=$?"; e=$(date +%s); grep -E "passed|failed|error" gpurun_out/gpu_suite.log | tail -3; echo "seconds: $((e-s))"
