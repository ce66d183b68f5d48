#!/bin/bash
# round 4, call 5: tests (GEMM, driver, full-size, config 5 at one rank's share); tile-width sweep; config 5 as rank 0 of 8
# with PMC passes for the aggregation launch on the 102 GB source
O=$GRAFT_REPO_ROOT/gpurun_out/r04c05; mkdir -p $O
timeout 1800 python -m pytest tests/test_hip_round4.py -q -k "stationary or random_walk_pairs or driver or full_size or config5" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -12 $O/tests.log
timeout 900 python scripts/bench_gemm.py --math nb --shapes collab_step_fwd,collab_step_dgrad,ddi_pred_fwd,ddi_pred_dgrad,cit_in_fwd_k192,cit_l2_fwd_k200 > $O/gemm_nb.jsonl 2> $O/gemm_nb.err
cat $O/gemm_nb.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['shape'], r.get('stationary_b'), r['ms'], 'ms', r['TFLOPs'], 'TF', r.get('frac_of_2500'))
"
tail -3 $O/gemm_nb.err
timeout 900 python bench.py --workload rmat --as-rank 0/8 --steps 5 --warmup 2 > $O/bench_rmat_rank0of8.json 2> $O/bench_rmat.err
tail -2 $O/bench_rmat.err; cat $O/bench_rmat_rank0of8.json | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({k: r[k] for k in ('value', 'ms_per_step', 'graph_build_s', 'gemm_layer1_ms', 'gemm_TFLOPs_f32_equivalent')})); print(json.dumps(r['roofline'], indent=1))
"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  d=gpurun_out/pmc_r5/$(echo $pass | tr ' ' '_')
  timeout 900 rocprofv3 --kernel-trace --pmc $pass -f csv -d $d -o a -- python3 bench.py --workload rmat --as-rank 0/8 --steps 2 --warmup 1 > /dev/null 2>&1
done
python3 scripts/pmc_collect.py csr_agg $O/pmc_rmat_rank0of8.json "gpurun_out/pmc_r5/**/*counter_collection.csv" | head -60
rm -rf gpurun_out/pmc_r5
