set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
S=collab_fwd,collab_fwd_plain,collab_dgrad,collab_wgrad,ddi_pred_fwd,square4k,collab_dgrad_T
for i in 1 2; do
PLNLP_HIP_LIB=$PWD/plnlp_amd/libplnlp_hip_oldgemm.so python scripts/bench_gemm.py --shapes $S > gpurun_out/r02/gemm_ab_old_$i.jsonl 2>/dev/null
python scripts/bench_gemm.py --shapes $S > gpurun_out/r02/gemm_ab_new_$i.jsonl 2>/dev/null
done
paste -d'\n' gpurun_out/r02/gemm_ab_old_1.jsonl gpurun_out/r02/gemm_ab_new_1.jsonl gpurun_out/r02/gemm_ab_old_2.jsonl gpurun_out/r02/gemm_ab_new_2.jsonl | cut -c1-120
python scripts/bench_agg.py --cases uniform_big,rmat23,collab --feat 256,512 --tune 0,4,8,12,halves > gpurun_out/r02/agg_tune.jsonl 2>/dev/null; cat gpurun_out/r02/agg_tune.jsonl | cut -c1-200
timeout 900 python -m pytest tests/test_hip_round2.py -q -m gpu -s -k "reference_style or hits20 or sharded or block" > gpurun_out/r02/pytest_round2b.log 2>&1; tail -8 gpurun_out/r02/pytest_round2b.log
timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -x > gpurun_out/r02/pytest_parity2.log 2>&1; tail -3 gpurun_out/r02/pytest_parity2.log
timeout 300 python bench.py --no-cpu-baseline --no-parity --no-stress > gpurun_out/r02/bench_collab_quick2.json 2>/dev/null; head -c 330 gpurun_out/r02/bench_collab_quick2.json
