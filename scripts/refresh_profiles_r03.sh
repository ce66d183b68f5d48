# round-3 measurement set: everything DESIGN.md / profiles/ quote, in one pass on one MI355X
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r03p; mkdir -p $R
# 1. the default command (what the driver runs), and the same command under rocprofv3 --kernel-trace --stats
python bench.py --steps 20 --warmup 5 > $R/bench_collab.json 2> $R/bench_collab.err; tail -c 300 $R/bench_collab.json
rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o collab -- python3 bench.py --steps 20 --warmup 5 > $R/bench_collab_under_rocprof.json 2>/dev/null
f=$(find $R/prof -name "*kernel_stats.csv" | head -1); cp $f $R/bench_collab_rocprofv3_kernel_stats.csv
f=$(find $R/prof -name "*kernel_trace.csv" | head -1)
python scripts/kernel_calls.py $f "csr_agg_vec_kernel<1, 32, false" 20 > $R/roofline_kernel_calls.txt
python scripts/kernel_calls.py $f "csr_agg_fused_kernel<1, 64, 1, 64, true" 20 >> $R/roofline_kernel_calls.txt
rm -rf $R/prof
# 2. the step alone under the kernel trace: per-step breakdown and launch sequence
rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o step -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
f=$(find $R/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 10 45 sequence > $R/step_breakdown_collab.txt
rm -rf $R/prof
# 3. HBM / fabric traffic (separate --pmc passes): the roofline kernel on the graph that does not fit, and the step's own launches
for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$R/pmc_a/$(echo $pass | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $pass -f csv -d $d -o a -- python3 scripts/bench_agg.py --cases uniform_big --feat 512 --tune 16 > /dev/null 2>&1
  d=$R/pmc_s/$(echo $pass | tr ' ' '_')
  PLNLP_AGG_FORM=65664 rocprofv3 --kernel-trace --pmc $pass -f csv -d $d -o s -- python3 scripts/bench_step_launches.py > /dev/null 2>&1   # (the form the tuner picks on this graph, pinned: counter runs serialise the kernels it would time)
done
python3 scripts/pmc_collect.py csr_agg $R/agg_pmc_uniform_big.json "$R/pmc_a/**/*counter_collection.csv" > /dev/null
python3 scripts/pmc_collect.py csr_agg $R/agg_pmc_step_launches.json "$R/pmc_s/**/*counter_collection.csv" > /dev/null
rm -rf $R/pmc_a $R/pmc_s
python3 scripts/bench_step_launches.py > $R/step_launches.json 2>/dev/null
# 4. the other workloads and forms
for w in ddi citation2; do
  python bench.py --workload $w --steps 10 --warmup 5 --no-parity --no-stress --cpu-steps 1 > $R/bench_$w.json 2>/dev/null
  rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o p -- python3 bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
  f=$(find $R/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 5 40 > $R/step_breakdown_$w.txt
  rm -rf $R/prof
done
python bench.py --workload rmat --scale 0.25 --steps 5 --warmup 2 > $R/bench_rmat_s025.json 2>/dev/null
for mode in shard grads scores; do
  python bench.py --force-dist --dp-exchange $mode --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $R/bench_collab_${mode}_1rank.json 2>/dev/null
done
python bench.py --workload citation2 --force-dist --dp-exchange grads --steps 10 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > $R/bench_citation2_grads_1rank.json 2>/dev/null
python scripts/bench_gemm.py --math ab --error > $R/gemm_microbench.jsonl 2>/dev/null
python scripts/bench_agg.py --cases collab,uniform_big,ddi --feat 256,512 --tune 0,16,32 > $R/agg_microbench.jsonl 2>/dev/null
python scripts/bench_agg.py --cases collab --feat 256 --tune 128 --hub-order 65536:256 >> $R/agg_microbench.jsonl 2>/dev/null
ls -la $R
