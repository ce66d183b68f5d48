#!/bin/bash
# GCN block conv + sharded GCN step on the GPU; citation2 through the row-sharded path on one rank
mkdir -p gpurun_out/c14
timeout 600 python -m pytest tests/test_hip_round2.py -q -m gpu -k "block_conv or rccl" -x 2>&1 | tail -15 > gpurun_out/c14/tests.log
timeout 600 python bench.py --workload citation2 --gpus 1 --steps 10 --warmup 3 --dp-exchange shard --no-parity > gpurun_out/c14/bench_citation2_shard1.json 2> gpurun_out/c14/bench_citation2_shard1.err
tail -5 gpurun_out/c14/bench_citation2_shard1.err
cat gpurun_out/c14/tests.log
