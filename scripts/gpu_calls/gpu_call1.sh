set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests/test_hip_round2.py -x -q -m gpu -s > gpurun_out/r02/pytest_round2.log 2>&1; echo "round2 rc=$?" 
tail -30 gpurun_out/r02/pytest_round2.log
timeout 1500 python -m pytest tests/test_hip_parity.py -q -m gpu > gpurun_out/r02/pytest_parity.log 2>&1; echo "parity rc=$?"
tail -5 gpurun_out/r02/pytest_parity.log
timeout 600 python bench.py > gpurun_out/r02/bench_collab.json 2> gpurun_out/r02/bench_collab.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r02/bench_collab.json
