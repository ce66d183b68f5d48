# round-4 measurement set: everything DESIGN.md / profiles/ quote for this round, in one pass on one MI355X
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r04p; mkdir -p $R
# 1. the default command (what the driver runs), and the same command under rocprofv3 --kernel-trace --stats
python bench.py --steps 20 --warmup 5 > $R/bench_collab.json 2> $R/bench_collab.err; tail -c 300 $R/bench_collab.json
rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o collab -- python3 bench.py --steps 20 --warmup 5 > $R/bench_collab_under_rocprof.json 2>/dev/null
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
# 3. counters of the split-bf16 GEMMs on the path's shapes: the stationary-weights kernel and the tile kernel (separate --pmc passes)
SH=ddi_pred_dgrad,ddi_pred_fwd,collab_step_fwd,collab_step_dgrad,cit_l2_fwd_k200
for arm in 1 0; do
  export PLNLP_GEMM_STATIONARY_B=$arm
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -f csv -d $R/pmc_g/a -o g -- python3 scripts/bench_gemm.py --math bf16x3 --shapes $SH --iters 3 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES -f csv -d $R/pmc_g/b -o g -- python3 scripts/bench_gemm.py --math bf16x3 --shapes $SH --iters 3 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY -f csv -d $R/pmc_g/c -o g -- python3 scripts/bench_gemm.py --math bf16x3 --shapes $SH --iters 3 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA -f csv -d $R/pmc_g/d -o g -- python3 scripts/bench_gemm.py --math bf16x3 --shapes $SH --iters 3 > /dev/null 2>&1
  python3 scripts/pmc_collect.py gemm $R/gemm_pmc_stationary$arm.json "$R/pmc_g/**/*counter_collection.csv" > /dev/null
  rm -rf $R/pmc_g
done
unset PLNLP_GEMM_STATIONARY_B
python3 - <<PY
import json
out = {}
for arm, name in ((1, "stationary_weights"), (0, "tile_128x128")):
    d = json.load(open("$R/gemm_pmc_stationary%d.json" % arm))
    for k, v in d.items():
        if v.get("launches", 0) < 3 or "SQ_WAVE_CYCLES" not in v:
            continue
        wc = v["SQ_WAVE_CYCLES"]
        v["derived"] = {"clock_GHz_per_xcd": v.get("GRBM_GUI_ACTIVE", 0) / 8 / (v["kernel_us_under_pmc"] * 1e3),
                        "wave_wait_any": v["SQ_WAIT_ANY"] / wc, "wave_issue_stall": v["SQ_WAIT_INST_ANY"] / wc,
                        "wave_lds_issue_stall": v.get("SQ_WAIT_INST_LDS", 0) / wc,
                        "valu_per_mfma": v.get("SQ_INSTS_VALU", 0) / max(v.get("SQ_INSTS_MFMA", 1), 1),
                        "lds_per_mfma": v.get("SQ_INSTS_LDS", 0) / max(v.get("SQ_INSTS_MFMA", 1), 1),
                        "mfma_busy_per_gui_cycle": v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(v.get("GRBM_GUI_ACTIVE", 1), 1)}
        out[name + " | " + k] = v
json.dump(out, open("$R/gemm_pmc.json", "w"), indent=1)
for k, v in out.items():
    print(k[:100], {a: round(b, 3) for a, b in v["derived"].items()}, round(v["kernel_us_under_pmc"], 1), "us")
PY
# 4. the other workloads and forms
for w in ddi citation2; do
  python bench.py --workload $w --steps 10 --warmup 5 --no-parity --no-stress --cpu-steps 1 > $R/bench_$w.json 2>/dev/null
done
python bench.py --workload rmat --as-rank 0/8 --steps 5 --warmup 2 > $R/bench_rmat_rank0of8.json 2>/dev/null
for mode in shard grads scores; do
  python bench.py --force-dist --dp-exchange $mode --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $R/bench_collab_${mode}_1rank.json 2>/dev/null
done
python scripts/bench_gemm.py --math st --error > $R/gemm_microbench.jsonl 2>/dev/null
python scripts/bench_gemm.py --math ab --error --shapes collab_fwd,collab_wgrad,ddi_pred_wgrad,ddi_enc_fwd,square4k,collab_wgrad_T > $R/gemm_microbench_f32_vs_bf16x3.jsonl 2>/dev/null
ls -la $R
