#!/bin/bash
mkdir -p gpurun_out/r06
timeout 600 python scripts/probe_dense_agg_error.py 2>&1 | grep -v amdgpu.ids | grep '"dense": true' > gpurun_out/r06/call17_dense_err.txt; cat gpurun_out/r06/call17_dense_err.txt
timeout 900 python -m pytest tests/test_hip_round4.py tests/test_hip_round6.py -q -m gpu -k "full_size_ddi or dense" 2>&1 | grep -v amdgpu.ids | grep -E "^E  .*Assertion|passed|failed|FAILED" | cut -c1-300
timeout 300 python - 2>&1 <<'PY' | grep -v amdgpu
import os, sys, json
sys.path.insert(0, os.getcwd())
import torch
import plnlp_amd as P
from plnlp_amd import synthetic
from bench import time_kernel
dev = torch.device("cuda")
g = synthetic.make_graph("ddi", seed=2, device=dev, weighted=False)
adj, n = g["adj_t"], g["num_nodes"]
x = torch.randn(n, 512, device=dev)
for rep in range(3):
    print("dense forward ms", round(time_kernel(lambda: P.ops.csr_aggregate(adj, x, "mean", False), iters=30) * 1e3, 4))
PY
