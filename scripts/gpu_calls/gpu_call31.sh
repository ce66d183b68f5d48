#!/bin/bash
mkdir -p gpurun_out/c31
timeout 900 python -m pytest tests/test_hip_round2.py -q -m gpu -k "pair_gathered or fused_edge_mlp" 2>&1 | tail -12 > gpurun_out/c31/tests.log
cat gpurun_out/c31/tests.log
