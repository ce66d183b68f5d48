#!/bin/bash
# round 4, call 2: round-4 tests without the oracle fixture, GEMM A/B (tile vs stationary-weights), counters for both
O=$GRAFT_REPO_ROOT/gpurun_out/r04c02; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_round4.py -q -k "stationary or random_walk_pairs or driver or full_size" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -30 $O/tests.log
timeout 600 python scripts/bench_gemm.py --math st --shapes collab_step_fwd,collab_step_dgrad,ddi_pred_fwd,ddi_pred_dgrad,cit_in_fwd_k192,cit_l2_fwd_k200,collab_fwd_plain > $O/gemm_st.jsonl 2> $O/gemm_st.err
cat $O/gemm_st.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['shape'], 'stationary' if r.get('stationary_b') else 'tile      ', r['ms'], 'ms', r['TFLOPs'], 'TF', r.get('frac_of_2500'))
"
tail -3 $O/gemm_st.err
# counters: the stationary kernel and the tile kernel on two shapes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for arm in 1 0; do
  export PLNLP_GEMM_STATIONARY_B=$arm
  rm -rf gpurun_out/pmc_g4
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -f csv -d gpurun_out/pmc_g4/a -o g -- python3 scripts/bench_gemm.py --math bf16x3 --shapes ddi_pred_dgrad,cit_in_fwd_k192,collab_step_fwd --iters 3 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES -f csv -d gpurun_out/pmc_g4/b -o g -- python3 scripts/bench_gemm.py --math bf16x3 --shapes ddi_pred_dgrad,cit_in_fwd_k192,collab_step_fwd --iters 3 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY -f csv -d gpurun_out/pmc_g4/c -o g -- python3 scripts/bench_gemm.py --math bf16x3 --shapes ddi_pred_dgrad,cit_in_fwd_k192,collab_step_fwd --iters 3 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA -f csv -d gpurun_out/pmc_g4/d -o g -- python3 scripts/bench_gemm.py --math bf16x3 --shapes ddi_pred_dgrad,cit_in_fwd_k192,collab_step_fwd --iters 3 > /dev/null 2>&1
  python3 scripts/pmc_collect.py gemm $O/pmc_gemm_stationary$arm.json "gpurun_out/pmc_g4/**/*counter_collection.csv" > /dev/null
  rm -rf gpurun_out/pmc_g4
done
unset PLNLP_GEMM_STATIONARY_B
python3 - <<PY
import json
for arm in (1, 0):
    d = json.load(open("$O/pmc_gemm_stationary%d.json" % arm))
    for k, v in d.items():
        if v.get("launches", 0) < 3 or "SQ_WAVE_CYCLES" not in v: continue
        wc = v["SQ_WAVE_CYCLES"]
        print(arm, k[:70], "us", round(v["kernel_us_under_pmc"], 1), "clk", round(v.get("clock_GHz", 0), 3),
              "wait_any", round(v["SQ_WAIT_ANY"] / wc, 2), "wait_inst", round(v["SQ_WAIT_INST_ANY"] / wc, 2),
              "lds_wait", round(v.get("SQ_WAIT_INST_LDS", 0) / wc, 2), "valu/mfma", round(v.get("SQ_INSTS_VALU", 0) / max(v.get("SQ_INSTS_MFMA", 1), 1), 2),
              "mfma_busy_cyc", v.get("SQ_VALU_MFMA_BUSY_CYCLES"), "gui", v.get("GRBM_GUI_ACTIVE"))
PY
timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress > $O/bench_collab.json 2> $O/bench_collab.err
python -c "
import json; r = json.loads(open('$O/bench_collab.json').read().strip().splitlines()[-1]); print('collab', r['ms_per_step'], 'ms', r['value'] / 1e6, 'M edges/s', r.get('roofline_mfma', {}).get('kernel_ms'))
"
