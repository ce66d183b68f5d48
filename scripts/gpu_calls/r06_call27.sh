#!/bin/bash
# round 6, call 27: few segments over a table beyond one XCD's L2: XCD-pinned column slabs (ddi's scorer backward): tests, the kernel
# alone with and without the slabs, the ddi step x 2, counters of the slab form
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round6.py -q -m gpu -x -k "segment_backward" 2>&1 | tail -8
PROBE_FORMS=wave,noslab,auto PROBE_SHAPES=ddi python scripts/probe_segment_bwd.py | tee $O/call27_times.txt
for rep in 1 2; do
  for form in noslab auto; do
  PLNLP_EDGE_SEGMENT=$form python bench.py --workload ddi --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('ddi $form rep$rep', round(r['ms_per_step'], 4))"
  done
done | tee $O/call27_steps.txt
mkdir -p $O/pmc27
for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$O/pmc27/$(echo $pass | tr ' ' '_')
  PROBE_ITERS=3 PROBE_FORMS=noslab,auto PROBE_SHAPES=ddi rocprofv3 --kernel-trace --pmc $pass -f csv -d $d -o s -- python3 scripts/probe_segment_bwd.py > /dev/null 2>&1
done
python3 scripts/pmc_collect.py edge_segment $O/call27_segment_pmc.json "$O/pmc27/**/*counter_collection.csv" > /dev/null
rm -rf $O/pmc27
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06/call27_segment_pmc.json'))
for k,v in d.items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
PY
