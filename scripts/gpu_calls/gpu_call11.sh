set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
PLNLP_HIP_LIB=$PWD/plnlp_amd/libplnlp_hip_bk16pf2.so timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "gemm or wgrad or encoder or mlp or single_step" > gpurun_out/r02/pytest_bk16.log 2>&1; tail -4 gpurun_out/r02/pytest_bk16.log
S=collab_fwd,collab_fwd_plain,collab_dgrad,collab_wgrad,ddi_pred_fwd,ddi_pred_wgrad,square4k,cit_in_fwd_k180,cit_l2_fwd_k200,collab_dgrad_T,ddi_enc_fwd
for i in 1 2; do
python scripts/bench_gemm.py --shapes $S > gpurun_out/r02/gemm_bk32_$i.jsonl 2>/dev/null
PLNLP_HIP_LIB=$PWD/plnlp_amd/libplnlp_hip_bk16pf2.so python scripts/bench_gemm.py --shapes $S > gpurun_out/r02/gemm_bk16pf2_$i.jsonl 2>/dev/null
PLNLP_HIP_LIB=$PWD/plnlp_amd/libplnlp_hip_bk16pf3.so python scripts/bench_gemm.py --shapes $S > gpurun_out/r02/gemm_bk16pf3_$i.jsonl 2>/dev/null
done
python - <<'PY'
import json
rows = {}
for v in ("bk32", "bk16pf2", "bk16pf3"):
    for i in (1, 2):
        for l in open(f"gpurun_out/r02/gemm_{v}_{i}.jsonl"):
            d = json.loads(l); rows.setdefault(d["shape"], {}).setdefault(v, []).append(d["TFLOPs"])
for s, r in rows.items():
    print(f"{s:22s}", "  ".join(f"{v}: {'/'.join(f'{x:6.1f}' for x in r[v])}" for v in r))
PY
