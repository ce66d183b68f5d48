set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_hip_round2.py -q -m gpu -x -k "row_restricted or slab or block or sharded_step" > gpurun_out/r02/pytest_sparsefwd.log 2>&1; tail -30 gpurun_out/r02/pytest_sparsefwd.log
timeout 1200 python -m pytest tests/test_hip_parity.py -q -m gpu > gpurun_out/r02/pytest_parity3.log 2>&1; tail -12 gpurun_out/r02/pytest_parity3.log
timeout 300 python bench.py --no-cpu-baseline --no-parity > gpurun_out/r02/bench_collab_sf.json 2>gpurun_out/r02/bench_collab_sf.err; head -c 330 gpurun_out/r02/bench_collab_sf.json; echo; tail -3 gpurun_out/r02/bench_collab_sf.err
PLNLP_SPARSE_FORWARD=0 timeout 300 python bench.py --no-cpu-baseline --no-parity --no-stress --no-roofline > gpurun_out/r02/bench_collab_nosf.json 2>/dev/null; head -c 330 gpurun_out/r02/bench_collab_nosf.json; echo
for w in ddi citation2; do timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --no-parity --no-stress --no-cpu-baseline > gpurun_out/r02/bench_$w.json 2>gpurun_out/r02/bench_$w.err; head -c 330 gpurun_out/r02/bench_$w.json; echo; tail -3 gpurun_out/r02/bench_$w.err; done
PLNLP_SPARSE_FORWARD=0 timeout 600 python bench.py --workload citation2 --steps 10 --warmup 2 --no-parity --no-stress --no-cpu-baseline --no-roofline > gpurun_out/r02/bench_citation2_nosf.json 2>/dev/null; head -c 330 gpurun_out/r02/bench_citation2_nosf.json; echo
