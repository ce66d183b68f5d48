set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 1200 python -m pytest tests/test_hip_parity.py -q -m gpu > gpurun_out/r02/pytest_parity4.log 2>&1; tail -8 gpurun_out/r02/pytest_parity4.log
timeout 900 python -m pytest tests/test_hip_round2.py -q -m gpu -k "not hits20" > gpurun_out/r02/pytest_round2c.log 2>&1; tail -8 gpurun_out/r02/pytest_round2c.log
python scripts/bench_gemm.py --shapes collab_fwd,collab_fwd_plain,collab_fwd_7rounds,collab_dgrad,ddi_pred_fwd,cit_in_fwd_k180,cit_l2_fwd_k200,collab_dgrad_T > gpurun_out/r02/gemm_microbench_v7.jsonl 2>/dev/null; cut -c1-140 gpurun_out/r02/gemm_microbench_v7.jsonl
timeout 300 python bench.py --no-cpu-baseline --no-parity > gpurun_out/r02/bench_collab_sf.json 2>gpurun_out/r02/bench_collab_sf.err; head -c 330 gpurun_out/r02/bench_collab_sf.json; echo; tail -3 gpurun_out/r02/bench_collab_sf.err
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r02/prof -o collab -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-stress --no-roofline > gpurun_out/r02/bench_collab_under_rocprof.json 2>/dev/null
f=$(find gpurun_out/r02/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 10 45 sequence > gpurun_out/r02/step_breakdown2.txt; head -75 gpurun_out/r02/step_breakdown2.txt
rm -rf gpurun_out/r02/prof
for w in ddi citation2; do timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --no-parity --no-stress --no-cpu-baseline > gpurun_out/r02/bench_$w.json 2>gpurun_out/r02/bench_$w.err; head -c 330 gpurun_out/r02/bench_$w.json; echo; done
