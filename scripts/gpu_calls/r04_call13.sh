#!/bin/bash
# the stale-memory checks (every fresh float buffer pre-filled with NaN), the row-indexed aggregation's padding rows,
# the edge-lists tests, and the aggregation tests of the earlier rounds
mkdir -p gpurun_out/r04c13
timeout 1500 python -m pytest tests/test_hip_round4.py -x -q -m gpu -k "stale or row_indexed or edge_lists" > gpurun_out/r04c13/new.log 2>&1
tail -15 gpurun_out/r04c13/new.log
timeout 900 python -m pytest tests/test_hip_round3.py tests/test_hip_round2.py tests/test_hip.py -x -q -m gpu -k "aggregat or agg or sage or gcn or compact" > gpurun_out/r04c13/agg.log 2>&1
tail -5 gpurun_out/r04c13/agg.log
