set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01
python bench.py > gpurun_out/r01/bench_collab.json 2> gpurun_out/r01/bench_collab.err
tail -c 600 gpurun_out/r01/bench_collab.json
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r01/prof -o collab -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/r01/bench_collab_under_rocprof.json 2>/dev/null
f=$(find gpurun_out/r01/prof -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r01/kernel_stats.csv
f=$(find gpurun_out/r01/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 10 40 > gpurun_out/r01/step_breakdown.txt
rm -rf gpurun_out/r01/prof
for w in ddi citation2; do python bench.py --workload $w --steps 10 --warmup 2 --no-parity --no-stress --cpu-steps 1 > gpurun_out/r01/bench_$w.json 2>/dev/null; done
python scripts/bench_gemm.py > gpurun_out/r01/gemm_microbench.jsonl 2>/dev/null
python scripts/bench_agg.py --cases collab,uniform,uniform_big,ddi > gpurun_out/r01/agg_microbench.jsonl 2>/dev/null
ls -la gpurun_out/r01
