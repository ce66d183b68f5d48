set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "gemm or wgrad or encoder or mlp or trajectory or single_step or row_sparse" > gpurun_out/r02/pytest_gemm.log 2>&1; echo "gemm tests rc=$?"
tail -5 gpurun_out/r02/pytest_gemm.log
timeout 600 python scripts/bench_gemm.py > gpurun_out/r02/gemm_microbench_v5.jsonl 2>gpurun_out/r02/gemm_microbench_v5.err; cat gpurun_out/r02/gemm_microbench_v5.jsonl
timeout 300 python scripts/debug_round2.py loop > gpurun_out/r02/debug_loop.log 2>&1; tail -20 gpurun_out/r02/debug_loop.log
timeout 900 python scripts/debug_round2.py hits 12 > gpurun_out/r02/debug_hits.log 2>&1; tail -16 gpurun_out/r02/debug_hits.log
timeout 600 python -m pytest tests/test_hip_round2.py -q -m gpu -k "mrr" > gpurun_out/r02/pytest_mrr.log 2>&1; tail -3 gpurun_out/r02/pytest_mrr.log
