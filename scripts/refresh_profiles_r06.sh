# round-6 measurement set: everything DESIGN.md / profiles/ quote for the FINAL round-6 tree, in one pass on one MI355X
# (the same-box A/Bs and experiments of the round are their own calls: scripts/gpu_calls/r06_call*.sh)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r06p; mkdir -p $R
# 1. the default command (what the driver runs), timed, and the same command under rocprofv3 --kernel-trace --stats
SECONDS=0
python bench.py --steps 20 --warmup 3 > $R/bench_collab.json 2> $R/bench_collab.err; echo "default bench.py: ${SECONDS}s wall" | tee $R/bench_collab_wall.txt; tail -c 300 $R/bench_collab.json
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
for mode in grads; do
  python bench.py --force-dist --dp-exchange $mode --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $R/bench_collab_${mode}_1rank.json 2>/dev/null
done
python bench.py --gpus 2 --share-gpu --steps 10 --warmup 3 --no-strong > $R/bench_collab_2ranks_shared_gpu.json 2>/dev/null
python scripts/bench_gemm.py --math st --error > $R/gemm_microbench.jsonl 2>/dev/null
# 4. counters for the kernels this round added: the whole-block stationary GEMM at 224-column tiles, the dense aggregation
rm -rf $R/pmc
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -f csv -d $R/pmc/a -o g -- python3 scripts/bench_gemm.py --math blk --shapes cit_in_fwd_k192,cit_l2_fwd_k200 --iters 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES -f csv -d $R/pmc/b -o g -- python3 scripts/bench_gemm.py --math blk --shapes cit_in_fwd_k192,cit_l2_fwd_k200 --iters 3 > /dev/null 2>&1
python3 scripts/pmc_collect.py gemm_ $R/gemm_pmc.json "$R/pmc/**/*counter_collection.csv" > /dev/null
rm -rf $R/pmc
for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  d=$R/pmc_d/$(echo $pass | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $pass -f csv -d $d -o s -- python3 scripts/probe_ddi_dense_agg.py > /dev/null 2>&1
done
python3 scripts/pmc_collect.py agg $R/ddi_agg_pmc.json "$R/pmc_d/**/*counter_collection.csv" > /dev/null
rm -rf $R/pmc_d
# 5. the segment backward alone (one wave per segment | long segments shared by a workgroup) and its counters
python scripts/probe_segment_bwd.py > $R/segment_backward_times.jsonl 2>/dev/null
mkdir -p $R/pmc_s
for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"; do
  d=$R/pmc_s/$(echo $pass | tr ' ' '_')
  PROBE_ITERS=3 rocprofv3 --kernel-trace --pmc $pass -f csv -d $d -o s -- python3 scripts/probe_segment_bwd.py > /dev/null 2>&1
done
python3 scripts/pmc_collect.py edge_segment $R/segment_backward_pmc.json "$R/pmc_s/**/*counter_collection.csv" > /dev/null
rm -rf $R/pmc_s
ls -la $R
