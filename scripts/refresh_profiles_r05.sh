# round-5 measurement set: everything DESIGN.md / profiles/ quote for the FINAL round-5 tree, in one pass on one MI355X
# (the same-box A/Bs of the round are their own calls: scripts/gpu_calls/r05_call*.sh)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r05p; mkdir -p $R
# 1. the default command (what the driver runs), and the same command under rocprofv3 --kernel-trace --stats
python bench.py --steps 20 --warmup 3 > $R/bench_collab.json 2> $R/bench_collab.err; tail -c 300 $R/bench_collab.json
rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o collab -- python3 bench.py --steps 20 --warmup 3 > $R/bench_collab_under_rocprof.json 2>/dev/null
f=$(find $R/prof -name "*kernel_stats.csv" | head -1); cp $f $R/bench_collab_rocprofv3_kernel_stats.csv
f=$(find $R/prof -name "*kernel_trace.csv" | head -1)
python scripts/kernel_calls.py $f "csr_agg_vec_kernel<1, 32, false" 20 > $R/roofline_kernel_calls.txt
rm -rf $R/prof
# 2. the step alone under the kernel trace: per-step breakdown and launch sequence (all three recipes)
for w in collab ddi citation2; do
  rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o step -- python3 bench.py --workload $w --steps 12 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
  f=$(find $R/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 6 45 sequence > $R/step_breakdown_$w.txt
  rm -rf $R/prof
done
# 3. the other workloads and forms
for w in ddi citation2; do
  python bench.py --workload $w --steps 10 --warmup 5 --no-parity --no-stress --cpu-steps 1 > $R/bench_$w.json 2>/dev/null
done
python bench.py --workload rmat --as-rank 0/8 --steps 5 --warmup 2 > $R/bench_rmat_rank0of8.json 2>/dev/null
for mode in shard grads scores; do
  python bench.py --force-dist --dp-exchange $mode --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $R/bench_collab_${mode}_1rank.json 2>/dev/null
done
python bench.py --gpus 2 --share-gpu --steps 10 --warmup 3 --no-strong > $R/bench_collab_2ranks_shared_gpu.json 2>/dev/null
python scripts/bench_agg.py --cases collab,uniform_big,ddi --feat 256,512 --tune 0,16,32 > $R/agg_microbench.jsonl 2>/dev/null
python scripts/bench_gemm.py --math st --error > $R/gemm_microbench.jsonl 2>/dev/null
ls -la $R
