set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_hip_round2.py -q -m gpu -x -k "slab or row_restricted or max" > gpurun_out/r02/pytest_slab2.log 2>&1; tail -4 gpurun_out/r02/pytest_slab2.log
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "aggregate or row_split or hub or encoder" > gpurun_out/r02/pytest_agg2.log 2>&1; tail -4 gpurun_out/r02/pytest_agg2.log
python scripts/bench_agg.py --cases ddi,collab,uniform_big --feat 256,512 --tune 0,16,32 > gpurun_out/r02/agg_tune3.jsonl 2>/dev/null; cut -c1-190 gpurun_out/r02/agg_tune3.jsonl
for w in ddi collab; do timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-parity --no-stress --no-cpu-baseline --no-roofline > gpurun_out/r02/bench_${w}_q4.json 2>/dev/null; head -c 330 gpurun_out/r02/bench_${w}_q4.json; echo; done
