set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 1800 python -m pytest tests/test_hip_round2.py -q -m gpu -s > gpurun_out/r02/pytest_round2.log 2>&1; echo "round2 rc=$?"
tail -15 gpurun_out/r02/pytest_round2.log
timeout 1500 python -m pytest tests/test_hip_parity.py -q -m gpu > gpurun_out/r02/pytest_parity.log 2>&1; echo "parity rc=$?"
tail -5 gpurun_out/r02/pytest_parity.log
timeout 600 python bench.py --no-cpu-baseline --no-parity --no-stress > gpurun_out/r02/bench_collab_quick.json 2> gpurun_out/r02/bench_collab_quick.err; echo "bench rc=$?"
head -c 400 gpurun_out/r02/bench_collab_quick.json; echo
timeout 600 python bench.py --force-dist --no-cpu-baseline --no-parity --no-stress --no-roofline > gpurun_out/r02/bench_collab_shard1.json 2> gpurun_out/r02/bench_collab_shard1.err; echo "shard1 rc=$?"
head -c 1200 gpurun_out/r02/bench_collab_shard1.json; echo; tail -5 gpurun_out/r02/bench_collab_shard1.err
timeout 900 python bench.py --workload rmat --scale 0.1 --steps 3 --warmup 1 > gpurun_out/r02/bench_rmat_s01.json 2> gpurun_out/r02/bench_rmat_s01.err; echo "rmat rc=$?"
cat gpurun_out/r02/bench_rmat_s01.json; tail -5 gpurun_out/r02/bench_rmat_s01.err
