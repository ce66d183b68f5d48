# round 4, final tree: the bench lines, the rocprofv3 stats of the default command and the step breakdowns (the GEMM counters and
# micro-benchmarks of scripts/refresh_profiles_r04.sh are unchanged by the later commits and are not repeated)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r04f; mkdir -p $R
python bench.py --steps 20 --warmup 5 > $R/bench_collab.json 2> $R/bench_collab.err; tail -c 300 $R/bench_collab.json
rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o collab -- python3 bench.py --steps 20 --warmup 5 > $R/bench_collab_under_rocprof.json 2>/dev/null
f=$(find $R/prof -name "*kernel_stats.csv" | head -1); cp $f $R/bench_collab_rocprofv3_kernel_stats.csv
f=$(find $R/prof -name "*kernel_trace.csv" | head -1)
python scripts/kernel_calls.py $f "csr_agg_vec_kernel<1, 32, false" 20 > $R/roofline_kernel_calls.txt
rm -rf $R/prof
for w in collab ddi citation2; do
  rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o step -- python3 bench.py --workload $w --steps 12 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
  f=$(find $R/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 6 45 sequence > $R/step_breakdown_$w.txt
  rm -rf $R/prof
done
for w in ddi citation2; do
  python bench.py --workload $w --steps 10 --warmup 5 --no-parity --no-stress --cpu-steps 1 > $R/bench_$w.json 2>/dev/null
done
for mode in shard grads scores; do
  python bench.py --force-dist --dp-exchange $mode --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $R/bench_collab_${mode}_1rank.json 2>/dev/null
done
ls -la $R
