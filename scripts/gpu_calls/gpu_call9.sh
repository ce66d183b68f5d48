set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_hip_round2.py -q -m gpu -x -k "row_restricted or block or sharded_step" > gpurun_out/r02/pytest_sf2.log 2>&1; tail -5 gpurun_out/r02/pytest_sf2.log
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "gemm or wgrad or row_sparse or trajectory" > gpurun_out/r02/pytest_g2.log 2>&1; tail -3 gpurun_out/r02/pytest_g2.log
timeout 300 python bench.py --no-cpu-baseline --no-parity --no-stress --no-roofline > gpurun_out/r02/bench_collab_q3.json 2>/dev/null; head -c 330 gpurun_out/r02/bench_collab_q3.json; echo
for w in collab ddi citation2; do
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r02/prof_$w -o p -- python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-stress --no-roofline > gpurun_out/r02/bench_${w}_under_rocprof.json 2>/dev/null
f=$(find gpurun_out/r02/prof_$w -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 5 40 > gpurun_out/r02/step_breakdown_$w.txt; head -45 gpurun_out/r02/step_breakdown_$w.txt
f=$(find gpurun_out/r02/prof_$w -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r02/kernel_stats_$w.csv
rm -rf gpurun_out/r02/prof_$w
done
