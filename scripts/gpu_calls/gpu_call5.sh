set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_hip_round2.py -q -m gpu -k "slab" > gpurun_out/r02/pytest_slab.log 2>&1; tail -4 gpurun_out/r02/pytest_slab.log
python scripts/bench_agg.py --cases uniform_big,collab,rmat23 --feat 256,512 --tune 0,16,32 > gpurun_out/r02/agg_tune2.jsonl 2>/dev/null; cut -c1-200 gpurun_out/r02/agg_tune2.jsonl
python scripts/bench_agg.py --cases uniform_big --feat 200,256,512 --weighted --tune 0,16,32 > gpurun_out/r02/agg_tune2w.jsonl 2>/dev/null; cut -c1-200 gpurun_out/r02/agg_tune2w.jsonl
python scripts/bench_gemm.py > gpurun_out/r02/gemm_microbench_v6.jsonl 2>/dev/null; cut -c1-140 gpurun_out/r02/gemm_microbench_v6.jsonl
bash scripts/pmc_gemm2.sh > gpurun_out/r02/pmc_gemm2.log 2>&1; tail -70 gpurun_out/r02/pmc_gemm2.log
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r02/prof -o collab -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-stress --no-roofline > gpurun_out/r02/bench_collab_under_rocprof.json 2>/dev/null
f=$(find gpurun_out/r02/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 10 45 sequence > gpurun_out/r02/step_breakdown.txt; head -60 gpurun_out/r02/step_breakdown.txt
f=$(find gpurun_out/r02/prof -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r02/kernel_stats.csv
rm -rf gpurun_out/r02/prof
timeout 300 python bench.py --force-dist --no-cpu-baseline --no-parity --no-stress --no-roofline > gpurun_out/r02/bench_collab_shard1b.json 2>/dev/null; head -c 300 gpurun_out/r02/bench_collab_shard1b.json
