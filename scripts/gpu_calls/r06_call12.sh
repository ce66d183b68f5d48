#!/bin/bash
# round 6, call 12: the dense aggregation -- its tests, the ddi parity tests on it, microbench per launch, same-box A/B of the ddi step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round6.py -q -m gpu -x -k "dense" 2>&1 | tail -12 > $O/call12_tests.txt; cat $O/call12_tests.txt
timeout 600 python - > $O/call12_dense_micro.txt 2>&1 <<'PY'
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
P.ops.tune_aggregation(adj, [512])
for on in (False, True, False, True):
    P.ops.DENSE_AGG["enabled"] = on
    t1 = time_kernel(lambda: P.ops.csr_aggregate(adj, x, "mean", False), iters=30)
    t2 = time_kernel(lambda: P.ops.csr_aggregate(adj.t_mean(), x, "sum", True), iters=30)
    print(json.dumps({"dense": on, "forward_mean_ms": round(t1 * 1e3, 4), "transposed_ms": round(t2 * 1e3, 4)}), flush=True)
PY
grep -v amdgpu.ids $O/call12_dense_micro.txt
for rep in 1 2 3; do for mode in 1 0; do
  PLNLP_DENSE_AGG=$mode timeout 300 python bench.py --workload ddi --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('ddi dense_agg=$mode rep$rep', round(r['ms_per_step'], 4), {k: v for k, v in r['kernel_families_per_step'].items() if k.startswith('agg')})"
done; done > $O/call12_ddi_ab.txt 2>&1; cat $O/call12_ddi_ab.txt
timeout 1500 python -m pytest tests/test_hip_round5.py tests/test_hip_round6.py tests/test_hip_round4.py -q -m gpu -k "ddi or full_size" 2>&1 | tail -8 > $O/call12_ddi_tests.txt; cat $O/call12_ddi_tests.txt
