set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
python scripts/bench_gemm.py --tail ab --shapes collab_fwd,collab_fwd_plain,collab_fwd_plain_7rounds,collab_dgrad,ddi_pred_fwd,cit_in_fwd_k180,cit_l2_fwd_k200,collab_dgrad_T,ddi_enc_fwd > gpurun_out/r02/gemm_tail_ab.jsonl 2>/dev/null; cut -c1-160 gpurun_out/r02/gemm_tail_ab.jsonl
bash scripts/pmc_gemm2.sh > gpurun_out/r02/pmc_gemm2.log 2>&1; tail -100 gpurun_out/r02/pmc_gemm2.log | cut -c1-200
bash scripts/pmc_agg.sh > gpurun_out/r02/pmc_agg.log 2>&1; tail -120 gpurun_out/r02/pmc_agg.log | cut -c1-200
