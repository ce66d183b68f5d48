#!/bin/bash
python -m pytest tests/ -x -q -m gpu > gpurun_out/gpu_suite.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/gpu_suite.log | tail -3
