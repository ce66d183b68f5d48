#!/bin/bash
# round 3, call 22: hub pass by source range + XCD-pinned slabs: sweeps on the collab graph
O=gpurun_out/r03c22; mkdir -p $O
python scripts/bench_agg.py --cases collab --feat 256 --tune 0,64 --hub-order none,8192:128,4096:128,16384:128,8192:64,8192:256,2048:128 > $O/agg_hub_sweep.jsonl 2> $O/agg_hub_sweep.err
echo "rc=$?" >> $O/agg_hub_sweep.err
python scripts/bench_agg.py --cases collab --feat 256 --tune 0,64 --threshold 128,256,512 --hub-order none,8192:128 > $O/agg_hub_thr.jsonl 2>> $O/agg_hub_sweep.err
python scripts/bench_agg.py --cases collab,ddi --feat 512 --tune 0,64 --hub-order none,8192:128 > $O/agg_hub_f512.jsonl 2>> $O/agg_hub_sweep.err
python scripts/bench_agg.py --cases uniform_big,citation2 --feat 256 --tune 0,64 > $O/agg_xcd_big.jsonl 2>> $O/agg_hub_sweep.err
cat $O/agg_hub_sweep.jsonl $O/agg_hub_thr.jsonl $O/agg_hub_f512.jsonl $O/agg_xcd_big.jsonl | cut -c1-330; tail -n 5 $O/agg_hub_sweep.err
