#!/bin/bash
# round 5, call 1: (a) the north-star aggregation kernel (uniform 2.9 M-node graph, F = 512) on the round-2 tree (b60826f,
# ./ab_r2), the round-3 tree (90a0749, ./ab_r3) and this tree, interleaved on ONE box, forms fixed (tune 0 = one wave per
# row, 16 = 128-column slabs: the form the tuner picks there), 3 repetitions; (b) which gloo collectives run on device tensors
# when several processes share the GPU (the world > 1 GPU tests need to know)
O=$GRAFT_REPO_ROOT/gpurun_out/r05c01; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for t in r2:ab_r2 r3:ab_r3 head:.; do
    n=${t%%:*}; d=${t##*:}
    ( cd $d && timeout 300 python scripts/bench_agg.py --cases uniform_big --feat 512 --tune 0,16 2> $O/agg_${n}_$rep.err | sed "s/^{/{\"tree\": \"$n\", \"rep\": $rep, /" >> $O/agg_ab.jsonl )
  done
done
cat $O/agg_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['tree'], r['rep'], 'tune', r['tune'], r['ms'], 'ms', r['frac_of_8TBps'])
"
timeout 300 python scripts/probe_gloo_cuda.py 2 > $O/gloo_probe_w2.txt 2>&1; tail -5 $O/gloo_probe_w2.txt
timeout 300 python scripts/probe_gloo_cuda.py 4 > $O/gloo_probe_w4.txt 2>&1; tail -7 $O/gloo_probe_w4.txt
# (c) the HIP path at world 2 and 4 on this one GPU over gloo
timeout 1500 python -m pytest tests/test_hip_multirank.py -x -q -m gpu > $O/multirank.txt 2>&1; tail -40 $O/multirank.txt
