#!/bin/bash
# round 6, call 15: the dense aggregation with a four-term split: tests, per-launch times, ddi step A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_round4.py tests/test_hip_round6.py tests/test_hip_round5.py -q -m gpu -s -k "dense or full_size_ddi or (teacher_forced and ddi_wide) or (trained_regime and ddi_wide) or (reverse and ddi_wide) or (mutation and ddi_wide)" 2>&1 | grep -v amdgpu.ids | grep -E "^E  |passed|failed|FAILED|teacher-forced|ddi_wide" | cut -c1-420 > $O/call15_tests.txt; cat $O/call15_tests.txt
timeout 600 python - > $O/call15_dense_micro.txt 2>&1 <<'PY'
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
grep -v amdgpu.ids $O/call15_dense_micro.txt
for rep in 1 2 3; do for mode in 1 0; do
  PLNLP_DENSE_AGG=$mode timeout 300 python bench.py --workload ddi --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('ddi dense_agg=$mode rep$rep', round(r['ms_per_step'], 4), {k: v for k, v in r['kernel_families_per_step'].items() if k.startswith('agg')})"
done; done > $O/call15_ddi_ab.txt 2>&1; cat $O/call15_ddi_ab.txt
rocprofv3 --kernel-trace --stats -f csv -d $O/prof15 -o step -- python3 bench.py --workload ddi --steps 12 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
f=$(find $O/prof15 -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 6 45 sequence > $O/call15_step_breakdown_ddi.txt; rm -rf $O/prof15; head -30 $O/call15_step_breakdown_ddi.txt
