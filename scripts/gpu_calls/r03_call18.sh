#!/bin/bash
# round 3, call 18: captured step with ROCm graph packet capture off -- tests, bench with / without capture, trained parity (collab part), probes
mkdir -p gpurun_out/r03c18
python -m pytest tests/test_hip_round3.py -x -q -m gpu -k "captured" > gpurun_out/r03c18/capture_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r03c18/capture_tests.log
python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-parity --no-stress > gpurun_out/r03c18/bench_capture.json 2> gpurun_out/r03c18/bench_capture.err
echo "rc=$?" >> gpurun_out/r03c18/bench_capture.err
PLNLP_CAPTURE=0 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > gpurun_out/r03c18/bench_eager.json 2> gpurun_out/r03c18/bench_eager.err
python -m pytest tests/test_hip_round3.py -x -q -m gpu -k "trained_regime" -s > gpurun_out/r03c18/trained.log 2>&1
echo "rc=$?" >> gpurun_out/r03c18/trained.log
for c in sage_mlp_whinge_noweight sage_mlp_auc; do
  python scripts/probe_drift.py $c > gpurun_out/r03c18/probe_$c.txt 2>&1
done
DBG_NODES=3000 DBG_EDGES=20500 DBG_H=64 DBG_SYNC_AT=step DBG_ITEM=1 timeout 120 python scripts/debug_capture.py 1.0 2048 0.3 2>&1 | tail -n 2 > gpurun_out/r03c18/dbg_item_default_env.txt
DBG_NODES=3000 DBG_EDGES=20500 DBG_H=64 DBG_SYNC_AT=step DBG_ITEM=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 timeout 120 python scripts/debug_capture.py 1.0 2048 0.3 2>&1 | tail -n 3 > gpurun_out/r03c18/dbg_item_packet_capture_on.txt
tail -n 4 gpurun_out/r03c18/capture_tests.log; tail -n 3 gpurun_out/r03c18/*.err; tail -n 40 gpurun_out/r03c18/trained.log | cut -c1-220; cat gpurun_out/r03c18/dbg_item*.txt | cut -c1-200
