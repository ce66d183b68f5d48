# round-2 measurement set: everything DESIGN.md / profiles/ quote, in one pass on one MI355X
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r02g; mkdir -p $R
python bench.py > $R/bench_collab.json 2> $R/bench_collab.err; tail -c 400 $R/bench_collab.json
rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o collab -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-stress > $R/bench_collab_under_rocprof.json 2>/dev/null
f=$(find $R/prof -name "*kernel_stats.csv" | head -1); cp $f $R/kernel_stats_collab.csv
f=$(find $R/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 10 45 sequence > $R/step_breakdown_collab.txt
rm -rf $R/prof
for w in ddi citation2; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-parity --no-stress --cpu-steps 1 > $R/bench_$w.json 2>/dev/null
  rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o p -- python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
  f=$(find $R/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 5 40 > $R/step_breakdown_$w.txt
  rm -rf $R/prof
done
python bench.py --force-dist --no-cpu-baseline --no-parity --no-stress --no-roofline > $R/bench_collab_shard_1rank.json 2>/dev/null
python bench.py --force-dist --dp-exchange scores --no-cpu-baseline --no-parity --no-stress --no-roofline > $R/bench_collab_scores_1rank.json 2>/dev/null
python bench.py --workload rmat --scale 0.25 --steps 3 --warmup 1 > $R/bench_rmat_s025.json 2>/dev/null; cat $R/bench_rmat_s025.json
python scripts/bench_gemm.py --math ab --error > $R/gemm_microbench.jsonl 2>/dev/null
python bench.py --workload citation2 --force-dist --dp-exchange shard --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-stress --no-roofline > $R/bench_citation2_shard_1rank.json 2>/dev/null
PLNLP_GEMM_MATH=f32 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-stress > $R/bench_collab_f32_mfma.json 2>/dev/null
python scripts/bench_agg.py --cases collab,uniform,uniform_big,ddi --feat 256,512 --tune 0,16,32 > $R/agg_microbench.jsonl 2>/dev/null
python scripts/bench_agg.py --cases rmat25 --feat 512 --tune 0,16,32 >> $R/agg_microbench.jsonl 2>/dev/null
ls -la $R
