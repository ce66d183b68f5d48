set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests/test_hip_parity.py -q -m gpu > gpurun_out/r02/pytest_parity5.log 2>&1; tail -4 gpurun_out/r02/pytest_parity5.log
timeout 1500 python -m pytest tests/test_hip_round2.py -q -m gpu -k "not hits20" > gpurun_out/r02/pytest_round2d.log 2>&1; tail -4 gpurun_out/r02/pytest_round2d.log
python scripts/bench_gemm.py > gpurun_out/r02/gemm_microbench_v8.jsonl 2>/dev/null; cut -c1-140 gpurun_out/r02/gemm_microbench_v8.jsonl
for w in collab ddi citation2; do timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-parity --no-stress --no-cpu-baseline --no-roofline > gpurun_out/r02/bench_${w}_q5.json 2>/dev/null; head -c 330 gpurun_out/r02/bench_${w}_q5.json; echo; done
