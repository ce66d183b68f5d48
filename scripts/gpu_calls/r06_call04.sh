#!/bin/bash
# round 6, call 4: gemm_x3b -- its tests, same-box A/B of the three steps (PLNLP_GEMM_BLOCK off / auto / all), clocks under both kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r06; mkdir -p $R
timeout 900 python -m pytest tests/test_hip_round6.py tests/test_hip_round4.py tests/test_hip_round5.py -x -q -m gpu -k "block or stationary or head or counters" 2>&1 | tail -8 > $R/call04_tests.txt
cat $R/call04_tests.txt
for rep in 1 2; do
for w in citation2 ddi collab; do
for mode in off auto all; do
  PLNLP_GEMM_BLOCK=$mode timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('$w $mode rep$rep', r.get('ms_per_step'))"
done; done; done > $R/call04_steps_ab.txt 2>&1
cat $R/call04_steps_ab.txt
rm -rf gpurun_out/pmc_x3b
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -f csv -d gpurun_out/pmc_x3b/a -o g -- python3 scripts/bench_gemm.py --math blk --shapes ddi_pred_dgrad,collab_fwd_plain,cit_in_fwd_k192 --iters 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES -f csv -d gpurun_out/pmc_x3b/b -o g -- python3 scripts/bench_gemm.py --math blk --shapes ddi_pred_dgrad,collab_fwd_plain,cit_in_fwd_k192 --iters 3 > /dev/null 2>&1
python3 scripts/pmc_collect.py gemm_ $R/call04_pmc_x3b.json "gpurun_out/pmc_x3b/**/*counter_collection.csv" > /dev/null
rm -rf gpurun_out/pmc_x3b
python3 - <<PY
import json
d=json.load(open("$R/call04_pmc_x3b.json"))
for k,v in d.items():
    if 'split_b' in k: continue
    print(k[:100]); print("   ", {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
PY
